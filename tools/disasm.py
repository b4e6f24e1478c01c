"""Disassembles one kernel of a built library: python tools/disasm.py <substring of the demangled name> [library.so]"""
import os, re, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "sqair_amd", "libsqair_hip.so")
so = open(lib, "rb").read()
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
for k, i in enumerate([m.start() for m in re.finditer(b"\x7fELF", so)][1:]):
    if struct.unpack_from("<H", so, i + 18)[0] != 224:
        continue
    path = "/tmp/_sq_dis%d.co" % k
    open(path, "wb").write(so[i:])
    out = subprocess.run([OBJDUMP, "-d", "--demangle", path], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", out)
    for b in blocks:
        head = b.split("\n", 1)[0]
        if pat in head:
            print(b)
            sys.exit(0)
print("not found")
