"""Randomised shapes through the dense-layer dispatch (sqair_linear_test -> sq_launch_linear: split-K, deep-K 32 x 32, row-slab,
macro-tile and the four LDS-tiled tile shapes) against an fp64 product, on the GPU box:  python tests/fuzz_linear.py [n] [seed]
Shapes are drawn around the dispatch boundaries (rows 1792 / 6000, 4 / 48 column tiles, K <= 64, K >= 640) as well as at random;
the bar is test_linear_mfma_matches_fp64's (2e-5 * max(1, sqrt(K / 256)) absolute on O(1) outputs).  With --gru: snt.GRU steps
(sqair_gru_test: the gate layer with the GRU1 epilogue, the candidate layer with GRU2) at random row counts and input widths
against the oracle's gru() in fp64, test_gru_step's bar (2e-5)."""
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


def gru_sweep(n, rng, lib, h, s):
    from oracle import sqair_oracle as O
    nh, bad = 256, 0
    for i in range(n):
        M = int(rng.choice([rng.integers(1, 700), rng.integers(1700, 2100), rng.integers(5900, 6500), rng.integers(6000, 16000)]))
        Kx = int(rng.choice([rng.integers(1, 70), rng.integers(60, 400), rng.integers(300, 700)]))
        P, flat = {}, []
        for g in "zrh":
            P["g.w" + g] = (rng.standard_normal((Kx, nh)) / np.sqrt(Kx)).astype(np.float32)
            P["g.u" + g] = (rng.standard_normal((nh, nh)) / np.sqrt(nh)).astype(np.float32)
            P["g.b" + g] = rng.standard_normal(nh).astype(np.float32) * 0.1
            flat += [P["g.w" + g].ravel(), P["g.u" + g].ravel(), P["g.b" + g].ravel()]
        x = rng.standard_normal((M, Kx)).astype(np.float32)
        hs = rng.standard_normal((M, nh)).astype(np.float32)
        out = torch.full((M, nh), float("nan"), device="cuda")
        scratch = torch.empty(max(1 << 22, M * (8 * nh + Kx + 64)), dtype=torch.float32, device="cuda")
        dx, dh, df = torch.tensor(x).cuda(), torch.tensor(hs).cuda(), torch.tensor(np.concatenate(flat)).cuda()
        rc = lib.sqair_gru_test(h, dx.data_ptr(), dh.data_ptr(), df.data_ptr(), out.data_ptr(), M, Kx, scratch.data_ptr(),
                                scratch.numel() * 4, s)
        torch.cuda.synchronize()
        P64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in P.items()}
        ref = O.gru(P64, "g", torch.tensor(x, dtype=torch.float64), torch.tensor(hs, dtype=torch.float64))
        err = float(np.abs(out.cpu().numpy() - ref.numpy()).max()) if rc == 0 else float("nan")
        ok = rc == 0 and err < 2e-5
        bad += not ok
        print("%s GRU M=%-6d Kx=%-4d  max err %.2e%s" % ("ok  " if ok else "FAIL", M, Kx, err, "" if rc == 0 else "  rc %d" % rc), flush=True)
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 200
    rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 0)
    lib = _capi.lib()
    h = C.c_void_p()
    cfg = make_config(make_flags(), (50, 50))
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if "--gru" in sys.argv:
        bad = gru_sweep(n, rng, lib, h, s)
        lib.sqair_destroy(h)
        print("{} GRU steps, {} failures".format(n, bad))
        sys.exit(bad)
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
