"""Code size in bytes of every kernel in libsqair_hip.so (the slot-loop kernels are sensitive to it: DESIGN.md section 2)."""
import re
import struct
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = open(os.path.join(ROOT, "sqair_amd", "libsqair_hip.so"), "rb").read()
rows = set()
for k, i in enumerate([m.start() for m in re.finditer(b"\x7fELF", so)][1:]):
    if struct.unpack_from("<H", so, i + 18)[0] == 224:  # EM_AMDGPU
        path = "/tmp/_sq_dev%d.co" % k
        open(path, "wb").write(so[i:])
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "--wide", path], capture_output=True, text=True).stdout
        for line in out.splitlines():
            p = line.split()
            if len(p) >= 8 and p[3] == "FUNC":
                rows.add((int(p[2]), p[7]))
names = subprocess.run(["c++filt"], input="\n".join(n for _, n in sorted(rows)), capture_output=True, text=True).stdout.splitlines()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for (sz, _), d in sorted(zip(sorted(rows), names), reverse=True):
    d = re.sub(r"\(.*", "", d)
    if pat in d:
        print("%7d  %s" % (sz, d))
