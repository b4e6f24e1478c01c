"""The C-ABI is usable without Python or torch: tests/native/abi_smoke.c (C99, gcc) links libsqair_hip.so and drives the
forward pass, the objective, a gradient evaluation and an optimiser step through include/sqair_hip.h alone."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "native", "abi_smoke")


def _build():
    from sqair_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        pytest.skip("libsqair_hip.so not built")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    cmd = [gcc, "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), SRC,
           "-L" + os.path.dirname(_capi.LIB_PATH), "-lsqair_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../../sqair_amd", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_c99_host_compiles_and_links_against_the_abi():
    exe = _build()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c99_host_runs_forward_backward_and_update():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke OK" in out.stdout
