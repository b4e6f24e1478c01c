"""Builds libsqair_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

Variants of the SAME sources (python sqair_amd/csrc/build.py [--force] [--timeline] [--knobs]):
  libsqair_hip.so            the product: no environment knobs, no timing code in any kernel
  libsqair_hip_timeline.so   -DSQAIR_TIMELINE: every wave stamps its start / end on the device wall clock (sqair_common.h);
                             what bench.py's roofline and tools/timeline.py measure the per-dispatch timeline with
  tools/bin/libsqair_hip_knobs.so  -DSQAIR_KNOBS: the measurement knobs of tools/ (tile shapes, fusion switches, dumps)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["sqair_api.hip", "sqair_linear.hip", "sqair_glue.hip", "sqair_bwd.hip", "sqair_train.hip", "sqair_linear_dx.hip"]
OUT = os.path.join(os.path.dirname(HERE), "libsqair_hip.so")
OUT_TIMELINE = os.path.join(os.path.dirname(HERE), "libsqair_hip_timeline.so")
OUT_KNOBS = os.path.join(ROOT, "tools", "bin", "libsqair_hip_knobs.so")
VARIANTS = {"product": (OUT, []), "timeline": (OUT_TIMELINE, ["-DSQAIR_TIMELINE"]), "knobs": (OUT_KNOBS, ["-DSQAIR_KNOBS"])}


def build(force=False, verbose=False, variant="product"):
    out, defs = VARIANTS[variant]
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".inc"))] + \
        [os.path.join(ROOT, "include", "sqair_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + defs + \
          ["-o", out] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    which = [v for v in VARIANTS if "--" + v in sys.argv] or ["product"]
    for v in which:
        print(build(force="--force" in sys.argv, verbose=True, variant=v))
