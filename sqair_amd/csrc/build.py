"""Builds libsqair_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

Variants of the SAME sources (python sqair_amd/csrc/build.py [--force] [--timeline] [--knobs] [--wide]):
  libsqair_hip.so            the product: no environment knobs, no timing code in any kernel
  libsqair_hip_timeline.so   -DSQAIR_TIMELINE: every wave stamps its start / end on the device wall clock (sqair_common.h);
                             what bench.py's roofline and tools/timeline.py measure the per-dispatch timeline with
  tools/bin/libsqair_hip_knobs.so  -DSQAIR_KNOBS: the measurement knobs of tools/ (tile shapes, fusion switches, dumps)
  libsqair_hip_wide.so       -DSQAIR_WIDE: the same C-ABI for the rest of the reference's flag range (n_what up to 128, up to 16
                             object slots, n_units up to 16): a larger slot record and plain-loop variants of the per-row kernels;
                             sqair_amd picks it from the flags when the product library's limits are exceeded

Every binary carries the hash of the sources it was compiled from (`-DSQAIR_BUILD_ID`, exported as `sqair_build_id()`, also
greppable in the file as `SQAIR_BUILD_ID=<16 hex>;`) and its variant (`sqair_build_flags()`).  A binary is rebuilt when that id
differs from the hash of the sources on disk -- not when modification times say so: a `.so` that travelled to another box next
to newer sources is caught, and `sqair_amd._capi.lib()` refuses a binary whose id is not the sources' id.
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["sqair_api.hip", "sqair_linear.hip", "sqair_glue.hip", "sqair_bwd.hip", "sqair_train.hip", "sqair_linear_dx.hip", "sqair_chain.hip"]
OUT = os.path.join(os.path.dirname(HERE), "libsqair_hip.so")
OUT_TIMELINE = os.path.join(os.path.dirname(HERE), "libsqair_hip_timeline.so")
OUT_KNOBS = os.path.join(ROOT, "tools", "bin", "libsqair_hip_knobs.so")
OUT_WIDE = os.path.join(os.path.dirname(HERE), "libsqair_hip_wide.so")
VARIANTS = {"product": (OUT, []), "timeline": (OUT_TIMELINE, ["-DSQAIR_TIMELINE"]), "knobs": (OUT_KNOBS, ["-DSQAIR_KNOBS"]),
            "wide": (OUT_WIDE, ["-DSQAIR_WIDE"])}
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
               "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
_ID_MARK = re.compile(rb"SQAIR_BUILD_ID=([0-9a-f]{16});")


def source_id():
    """sha256 (16 hex digits) over the kernel sources (csrc/*.hip, *.h, *.inc), the C-ABI header and the compiler flags: the
    identity of what a libsqair_hip*.so was, or would be, compiled from."""
    hsh = hashlib.sha256()
    for f in sorted(f for f in os.listdir(HERE) if f.endswith((".hip", ".h", ".inc"))):
        hsh.update(f.encode())
        hsh.update(open(os.path.join(HERE, f), "rb").read())
    hsh.update(open(os.path.join(ROOT, "include", "sqair_hip.h"), "rb").read())
    hsh.update(" ".join(HIPCC_FLAGS).encode())
    return hsh.hexdigest()[:16]


def binary_id(path):
    """The build id embedded in a shared object (read from the file, nothing is loaded); None if absent."""
    try:
        m = _ID_MARK.search(open(path, "rb").read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force=False, verbose=False, variant="product"):
    out, defs = VARIANTS[variant]
    sid = source_id()
    if not force and binary_id(out) == sid:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [f for f in HIPCC_FLAGS if f != "-shared"] + defs + \
        ['-DSQAIR_BUILD_ID="{}"'.format(sid), '-DSQAIR_BUILD_VARIANT="{}"'.format(variant)]
    # one hipcc per translation unit, all at once (the six files take ~25 s each), then the link
    objdir = os.path.join(HERE, "_obj", variant)
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    cmds = [[hipcc] + common + ["-c", os.path.join(HERE, s), "-o", o] for s, o in zip(SOURCES, objs)]
    if verbose:
        print(" ".join(cmds[0][:-3]) + " -c {" + ",".join(SOURCES) + "}")
    procs = [subprocess.Popen(c) for c in cmds]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), cmds[[i for i, r in enumerate(rcs) if r][0]])
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    assert binary_id(out) == sid, "the build id did not make it into {}".format(out)
    return out


if __name__ == "__main__":
    which = [v for v in VARIANTS if "--" + v in sys.argv] or ["product"]
    for v in which:
        print(build(force="--force" in sys.argv, verbose=True, variant=v))
