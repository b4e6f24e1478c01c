#!/usr/bin/env python
"""Back-to-back time of the decoder canvas kernels (k_insert_loglik* and the adjoint) at the grid of one pass -- R x T workgroups
as ONE launch of B = sequences x T "sequences" -- for one or more builds of the library on one box, alternating rounds:

    python tools/time_insert.py [--hw 128 128] [--rows 1600] [--slots 4] [--rounds 5] [--reps 20] [A.so B.so ...]

Inputs mimic the random-initialised model the bench runs (where logits ~ 0.3 N(0, 1): boxes of about half the frame per axis,
half of the slots present).  Libraries other than the tree's are loaded with the stale-binary check waived (A / B only)."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--hw", type=int, nargs=2, default=[128, 128])
    ap.add_argument("--rows", type=int, default=1600)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--where-std", type=float, default=0.3)
    args = ap.parse_args()
    H, W = args.hw
    K, N, G = args.k, args.slots, 20
    B = args.rows // K
    R = B * K
    F = make_flags(k_particles=K, n_steps_per_image=N)
    cfg = make_config(F, (H, W))
    rng = np.random.default_rng(0)
    dev = lambda a: torch.tensor(a, device="cuda")
    gl = dev((rng.standard_normal((R, N, G * G)) * 0.3).astype(np.float32))
    where = dev((rng.standard_normal((R, N, 4)) * args.where_std).astype(np.float32))
    pres = dev((rng.uniform(size=(R, N)) > 0.5).astype(np.float32))
    img = dev(rng.uniform(size=(B, H, W)).astype(np.float32))
    mean_img = dev(rng.uniform(size=(H, W)).astype(np.float32))
    g_ll = dev(rng.standard_normal(R).astype(np.float32))
    dll = torch.zeros(R, device="cuda")
    d_gl = torch.zeros(R, N, G * G, device="cuda")
    d_wh = torch.zeros(R, N, 4, device="cuda")
    d_mean = torch.zeros(H, W, device="cuda")
    scratch = torch.empty(R * H * W, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    paths = args.libs or [_capi.LIB_PATH]
    libs = []
    for p in paths:
        l = _capi.lib(os.path.abspath(p), allow_stale=True)
        h = C.c_void_p()
        assert l.sqair_create(C.byref(cfg), C.byref(h)) == 0
        libs.append((p, l, h))

    def fwd(l, h):
        assert l.sqair_st_insert_loglik(h, gl.data_ptr(), where.data_ptr(), pres.data_ptr(), img.data_ptr(), mean_img.data_ptr(),
                                        None, dll.data_ptr(), B, s) == 0

    needs_reduce = set()   # builds older than the NULL-d_mean_img convention: their time includes the (slow, test-only) row reduction

    def bwd(l, h):
        dm = d_mean.data_ptr() if id(l) in needs_reduce else None
        rc = l.sqair_st_insert_loglik_bwd(h, gl.data_ptr(), where.data_ptr(), pres.data_ptr(), img.data_ptr(), mean_img.data_ptr(),
                                          g_ll.data_ptr(), d_gl.data_ptr(), d_wh.data_ptr(), dm, scratch.data_ptr(),
                                          scratch.numel() * 4, B, s)
        if rc == -1 and dm is None:
            needs_reduce.add(id(l))
            return bwd(l, h)
        assert rc == 0

    def timed(fn, l, h):
        fn(l, h)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn(l, h)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps

    res = {p: {"fwd": [], "bwd": []} for p, _, _ in libs}
    ref = None
    for _ in range(args.rounds):
        for p, l, h in libs:
            res[p]["fwd"].append(timed(fwd, l, h))
            res[p]["bwd"].append(timed(bwd, l, h))
    for p, l, h in libs:
        fwd(l, h)
        bwd(l, h)
        torch.cuda.synchronize()
        out = (dll.clone(), d_gl.clone(), d_wh.clone(), scratch.clone())
        if ref is None:
            ref = out
        dev_ = [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(out, ref)]
        print("{:40s} {}x{} R={} N={}: fwd {:7.1f} us (min {:7.1f})   bwd {:7.1f} us (min {:7.1f})   vs first: {}".format(
            os.path.basename(p), H, W, R, N, float(np.median(res[p]["fwd"])), min(res[p]["fwd"]), float(np.median(res[p]["bwd"])),
            min(res[p]["bwd"]), " ".join("%.1e" % v for v in dev_)))
        l.sqair_destroy(h)


if __name__ == "__main__":
    main()
