"""Which kernels wait for a LOAD that was requested AFTER a store or an atomic?  On gfx950 stores and atomics without return count in
vmcnt like loads and the counter retires in order, so waiting for such a load also waits for the older store's round trip -- the
weight-gradient kernel's atomics loop lost 16 round trips that way.  Loads requested BEFORE the stores
are harmless (the wait leaves the younger stores outstanding).  Static, linear scan of the disassembly: branches are not followed
(a store in one arm and a load in another are reported too), loops are not unrolled.

    python tools/store_then_wait.py [library.so]"""
import os, re, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sqair_amd", "libsqair_hip.so")
so = open(lib, "rb").read()
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
for k, i in enumerate([m.start() for m in re.finditer(b"\x7fELF", so)][1:]):
    if struct.unpack_from("<H", so, i + 18)[0] != 224:
        continue
    path = "/tmp/_sq_stw%d.co" % k
    open(path, "wb").write(so[i:])
    out = subprocess.run([OBJDUMP, "-d", "--demangle", path], capture_output=True, text=True).stdout
    for b in re.split(r"\n(?=[0-9a-f]{16} <)", out):
        head = b.split("\n", 1)[0]
        m = re.match(r"[0-9a-f]{16} <(.*)>:", head)
        if not m:
            continue
        name = re.sub(r"\(.*", "", m.group(1))
        lines = [l.strip() for l in b.split("\n")[1:]]
        stores = 0
        late_loads = 0   # loads requested after a store
        hits = []
        for n, l in enumerate(lines):
            op = l.split()[0] if l else ""
            if op.startswith(("global_store", "global_atomic", "buffer_store", "buffer_atomic", "flat_store", "flat_atomic")):
                stores += 1
            elif op.startswith(("global_load", "buffer_load", "flat_load")) and stores:
                late_loads += 1
            elif op == "s_waitcnt" and "vmcnt" in l and late_loads:
                hits.append((n, stores, late_loads))
                late_loads = 0
        if hits:
            print("%-60s %2d waits for loads requested after a store / atomic (instructions %s of %d)" % (
                name[:60], len(hits), ", ".join(str(h[0]) for h in hits[:6]), len(lines)))
