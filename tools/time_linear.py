"""Average time of one dense-layer launch for a list of (M, K, N, act) shapes, launches issued back to back (throughput
view, no dependent-boundary cost): python tools/time_linear.py [M,K,N,act ...].  SQAIR_MT="MT,NT" forces a tile shape of the
throughput variants."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sqair_amd import _capi  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import make_config  # noqa: E402

SHAPES = [(640, 362, 1152, 0), (640, 312, 768, 0), (640, 400, 256, 1), (6400, 56, 256, 1), (6400, 256, 256, 1), (6400, 256, 400, 0),
          (1280, 256, 256, 1), (5120, 362, 1152, 0), (51200, 256, 256, 1)]


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or SHAPES
    lib = _capi.lib(__import__('os').environ.get('SQAIR_TOOL_LIB'), allow_stale=bool(__import__('os').environ.get('SQAIR_TOOL_LIB')))  # tools may point at the -DSQAIR_KNOBS build (tools/bin/libsqair_hip_knobs.so)
    h = C.c_void_p()
    cfg = make_config(make_flags(), (50, 50))
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, K, N, act in shapes:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(K, N, device="cuda") / np.sqrt(K)
        b = torch.randn(N, device="cuda")
        y = torch.zeros(M, N, device="cuda")
        nt, kc = (N + 15) // 16, (K + 15) // 16
        scratch = torch.zeros(2 * nt * kc * 256 + 2 * nt * 16 + 256 + M * ((K + 3) // 4 * 4) + 64, device="cuda")
        us = C.c_float()
        rc = lib.sqair_debug_linear_time(h, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, act, scratch.data_ptr(),
                                         scratch.numel() * 4, 50, C.byref(us), s)
        assert rc == 0, lib.sqair_last_error(h)
        ref = x.double() @ w.double() + b.double()
        if act == 1:
            ref = torch.where(ref > 0, ref, torch.expm1(ref))
        err = float((y.double() - ref).abs().max())
        fl = 2.0 * M * K * N
        print("M=%-6d K=%-5d N=%-5d act=%d : %8.2f us  %6.1f TFLOP/s (%.2f of 157.3)  max err %.1e" % (
            M, K, N, act, us.value, fl / us.value / 1e6, fl / us.value / 1e6 / 157.3, err))
        if hasattr(lib, "sqair_debug_big_phases"):   # knob builds: phase stamps of one workgroup of k_linear_big (10 ns ticks)
            ph = (C.c_uint64 * 8)()
            torch.cuda.synchronize()
            if lib.sqair_debug_big_phases(ph) == 0 and ph[4] > ph[0]:
                t = [ph[i] - ph[0] for i in range(5)]
                print("   k_linear_big phases of one workgroup [us]: setup %.2f | first loads %.2f | K loop %.2f | epilogue %.2f (park tile 0 at +%.2f, tile 1 at +%.2f)" % (
                    t[1] * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (ph[5] - ph[3]) * 0.01, (ph[6] - ph[3]) * 0.01))
                kc = (K + 15) // 16
                print("   K loop: %d shader cycles = %.0f per chunk (%d chunks, first-load wait included); clock %.2f GHz" % (
                    ph[7], ph[7] / kc, kc, ph[7] / max(1, t[3] - t[2]) * 0.1))
    lib.sqair_destroy(h)


if __name__ == "__main__":
    main()
