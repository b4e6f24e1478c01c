"""VGPRs / AGPRs / SGPRs / scratch / LDS / spill counts of the kernels of a libsqair_hip*.so (from the AMDGPU metadata notes).
    python tools/kernel_regs.py [substring of the demangled name] [library.so]
`kernel_table(path)` returns the same as a list of dicts (tests/test_kernel_resources.py: no kernel of a shipped library spills)."""
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count",
        "sgpr_spill_count")


def kernel_table(path):
    so = open(path, "rb").read()
    rows = []
    for k, i in enumerate([m.start() for m in re.finditer(b"\x7fELF", so)][1:]):
        if struct.unpack_from("<H", so, i + 18)[0] != 224:   # EM_AMDGPU
            continue
        co = "/tmp/_sq_regs_%d_%d.co" % (os.getpid(), k)
        open(co, "wb").write(so[i:])
        out = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
        os.remove(co)
        # one YAML list item per kernel ("  - .agpr_count: ..." opens it; its keys are in alphabetical order, so some of them
        # come BEFORE .name); nested items (.args) are indented deeper
        cur, ind = None, None
        in_kernels = False
        for line in out.splitlines():
            if re.match(r"\s*amdhsa\.kernels:", line):
                in_kernels = True
                continue
            if not in_kernels:
                continue
            m = re.match(r"(\s*)(- )?\.(\w+):\s*(.*)", line)
            if not m:
                continue
            lead, dash, key, val = len(m.group(1)), m.group(2), m.group(3), m.group(4).strip()
            if dash and (ind is None or lead == ind) and key != "address_space" and key != "offset" and key != "actual_access":
                if ind is None:
                    ind = lead
                if cur and "vgpr_count" in cur and "mangled" in cur:
                    rows.append(cur)
                cur = {}
            if cur is None or lead != ind + 2 and not (dash and lead == ind):
                continue
            if key == "name":
                cur["mangled"] = val
            elif key in KEYS:
                cur[key] = int(val)
        if cur and "vgpr_count" in cur and "mangled" in cur:
            rows.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    seen, uniq = set(), []
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", n)
        if r["mangled"] not in seen:
            seen.add(r["mangled"])
            uniq.append(r)
    return uniq


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "sqair_amd", "libsqair_hip.so")
    for r in kernel_table(path):
        if pat in r["name"]:
            print("%-64s vgpr %4d agpr %3d sgpr %4d scratch %5d lds %6d vgpr-spill %3d sgpr-spill %3d" % (
                r["name"][:64], r["vgpr_count"], r.get("agpr_count", 0), r.get("sgpr_count", 0), r.get("private_segment_fixed_size", 0),
                r.get("group_segment_fixed_size", 0), r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0)))


if __name__ == "__main__":
    main()
