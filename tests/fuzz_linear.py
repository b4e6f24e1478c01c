"""Randomised shapes through the dense-layer dispatch (sqair_linear_test -> sq_launch_linear: split-K, deep-K 32 x 32, row-slab,
macro-tile and the four LDS-tiled tile shapes) against an fp64 product, on the GPU box:  python tests/fuzz_linear.py [n] [seed]
Shapes are drawn around the dispatch boundaries (rows 1792 / 6000, 4 / 48 column tiles, K <= 64, K >= 640) as well as at random;
the bar is test_linear_mfma_matches_fp64's (2e-5 * max(1, sqrt(K / 256)) absolute on O(1) outputs)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sqair_amd import _capi  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import make_config  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _capi.lib()
    h = C.c_void_p()
    cfg = make_config(make_flags(), (50, 50))
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = 0
    for i in range(n):
        M = int(rng.choice([rng.integers(1, 400), rng.integers(400, 2200), rng.integers(1700, 2100), rng.integers(5900, 6500),
                            rng.integers(6000, 30000)]))
        K = int(rng.choice([rng.integers(1, 70), rng.integers(60, 130), rng.integers(100, 700), rng.integers(600, 1400)]))
        N = int(rng.choice([rng.integers(1, 70), rng.integers(50, 110), rng.integers(100, 800), rng.integers(740, 1300)]))
        if M * (K + N) > 3e7:
            M = int(3e7 // (K + N))
        act = int(rng.integers(0, 5))
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        dx, dw, db = torch.tensor(x).cuda(), torch.tensor(w).cuda(), torch.tensor(b).cuda()
        y = torch.full((M, N), float("nan"), device="cuda")
        scratch = torch.empty(4 * ((K + 15) // 16) * ((N + 15) // 16) * 256 + 8192 + M * (K + 4), dtype=torch.float32, device="cuda")
        rc = lib.sqair_linear_test(h, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), M, K, N, act, scratch.data_ptr(),
                                   scratch.numel() * 4, s)
        torch.cuda.synchronize()
        ref = torch.tensor(x, dtype=torch.float64) @ torch.tensor(w, dtype=torch.float64) + torch.tensor(b, dtype=torch.float64)
        ref = [lambda v: v, lambda v: torch.where(v > 0, v, torch.expm1(v)), torch.tanh, torch.sigmoid,
               lambda v: torch.nn.functional.softplus(v) + 1e-2][act](ref)
        err = float(np.abs(y.cpu().numpy().astype(np.float64) - ref.numpy()).max()) if rc == 0 else float("nan")
        ok = rc == 0 and err < 2e-5 * max(1.0, np.sqrt(K / 256.0))
        bad += not ok
        print("%s M=%-6d K=%-5d N=%-5d act=%d  max err %.2e" % ("ok  " if ok else "FAIL", M, K, N, act, err), flush=True)
    lib.sqair_destroy(h)
    print("{} shapes, {} failures".format(n, bad))
    sys.exit(bad)


if __name__ == "__main__":
    main()
