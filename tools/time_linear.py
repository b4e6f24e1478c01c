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
    lib = _capi.lib()
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
    lib.sqair_destroy(h)


if __name__ == "__main__":
    main()
