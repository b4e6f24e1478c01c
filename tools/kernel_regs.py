"""VGPRs / SGPRs / scratch / LDS / code size of the kernels of a libsqair_hip*.so (from the AMDGPU metadata notes)."""
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "sqair_amd", "libsqair_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
so = open(path, "rb").read()
for k, i in enumerate([m.start() for m in re.finditer(b"\x7fELF", so)][1:]):
    if struct.unpack_from("<H", so, i + 18)[0] != 224:
        continue
    co = "/tmp/_sq_regs%d.co" % k
    open(co, "wb").write(so[i:])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    for line in out.splitlines():
        m = re.match(r"\s+[-.]?\s*\.?(\w+):\s+(.*)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "name" and not val.endswith(".kd"):
            cur = {"name": val}
        elif key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "agpr_count"):
            cur[key] = val
        elif key == "symbol" and cur:
            name = subprocess.run(["c++filt", cur.get("name", "")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name)
            if pat in name:
                print("%-60s vgpr %4s agpr %3s sgpr %4s scratch %5s lds %6s spill %s" % (
                    name[:60], cur.get("vgpr_count"), cur.get("agpr_count", "0"), cur.get("sgpr_count"), cur.get("private_segment_fixed_size"),
                    cur.get("group_segment_fixed_size"), cur.get("vgpr_spill_count", "0")))
            cur = {}
