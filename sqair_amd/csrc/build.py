"""Builds libsqair_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["sqair_api.hip", "sqair_linear.hip", "sqair_glue.hip", "sqair_bwd.hip", "sqair_train.hip", "sqair_linear_dx.hip"]
OUT = os.path.join(os.path.dirname(HERE), "libsqair_hip.so")


def build(force=False, verbose=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, f) for f in ("sqair_common.h", "sqair_glue.h", "sqair_internal.h", "sqair_dx.h", "sqair_rowops.h", "sqair_lin_device.h", "sqair_bwd.h", "sqair_linear_kernel.inc", "sqair_wgrad_kernel.inc")] + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "sqair_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           "-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
