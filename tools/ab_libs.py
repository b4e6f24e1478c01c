#!/usr/bin/env python
"""A / B of two builds of the library on ONE box: alternating rounds of the forward step and the training step (what bench.py
times), so that box-to-box and run-to-run variation (about 1 %) cancels.  Kernel-level changes are judged by this, on the product
build (the stamped build schedules differently).

    python tools/ab_libs.py A.so B.so [--cfg 2] [--rounds 6] [--steps 30]

A / B may be builds of OTHER sources than the tree's (the stale-binary check is waived here, and only here)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs=2)
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import numpy as np
    import torch
    from sqair_amd import _capi
    from sqair_amd import timeline as TL
    from sqair_amd.data import config_inputs
    from sqair_amd.flags import make_flags
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.params import init_params
    from sqair_amd.train import Trainer
    ov, obs, nums, _ = config_inputs(args.cfg, B=args.batch or None)
    F = make_flags(**ov)
    Ftr = make_flags(**dict(ov, learning_rate=1e-5, train_itr=1000000))
    hw = tuple(int(v) for v in obs.shape[2:])
    B, K = int(obs.shape[1]), int(F.k_particles)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    legs = []
    for path in args.libs:
        path = os.path.abspath(path)
        _capi.lib(path, allow_stale=True)
        core = SqairCore(F, hw, lib_path=path)
        with core.on_stream():
            core.set_params(P)
            m = Model(obs, None, core, K, presence=nums, outputs="minimal")
        n = [0]

        def fwd(core=core, n=n):
            core.draw_noise(seed=1000, step=n[0], global_batch=B, b0=0)
            n[0] += 1
            core.forward(use_graph=True)
        tr = Trainer(m, Ftr, use_graph=True)
        legs.append((path, core, fwd, lambda tr=tr: tr.step(seed=2000, global_batch=B, b0=0)))
    res = {p: ([], []) for p, _, _, _ in legs}
    for r in range(args.rounds):
        for path, core, fwd, trn in legs:
            res[path][0].append(TL.time_steps(core, fwd, steps=args.steps, warm=3))
            res[path][1].append(TL.time_steps(core, trn, steps=max(5, args.steps // 2), warm=2))
    for path, _, _, _ in legs:
        f, t = res[path]
        print("{:60s} build {}  forward {:.4f} ms (min {:.4f})  training {:.4f} ms (min {:.4f})".format(
            os.path.relpath(path, ROOT), _capi.lib(path, allow_stale=True).sqair_build_id().decode(), float(np.median(f)), min(f),
            float(np.median(t)), min(t)))


if __name__ == "__main__":
    main()
