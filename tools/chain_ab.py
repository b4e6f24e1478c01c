#!/usr/bin/env python
"""A / B of the in-launch slot chain across builds of the library on ONE box: the forward step of bench.py with the option
slot_chain on, in alternating rounds (and the launch path of the first build as the yardstick).

    python tools/chain_ab.py A.so [B.so ...] [--cfg 2] [--batch n] [--rounds 5] [--steps 30] [--tile-rows n]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--tile-rows", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    from sqair_amd import _capi
    from sqair_amd import timeline as TL
    from sqair_amd.data import config_inputs
    from sqair_amd.flags import make_flags
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.params import init_params
    ov, obs, nums, _ = config_inputs(args.cfg, B=args.batch or None)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    B, K = int(obs.shape[1]), int(F.k_particles)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    legs = []
    for i, path in enumerate(args.libs):
        path = os.path.abspath(path)
        _capi.lib(path, allow_stale=True)
        for chain in ((0, 1) if i == 0 else (1,)):
            core = SqairCore(F, hw, lib_path=path, options=(dict({"slot_chain": chain}, **({"slot_chain_tile_rows": args.tile_rows} if args.tile_rows else {})) if chain else None))
            with core.on_stream():
                core.set_params(P)
                m = Model(obs, None, core, K, presence=nums, outputs="minimal")
            n = [0]

            def fwd(core=core, n=n):
                core.draw_noise(seed=1000, step=n[0], global_batch=B, b0=0)
                n[0] += 1
                core.forward(use_graph=True)
            legs.append(("{} {}".format(os.path.basename(path), "chain" if chain else "launches"), core, fwd, m))
    res = {name: [] for name, _, _, _ in legs}
    for r in range(args.rounds):
        for name, core, fwd, m in legs:
            res[name].append(TL.time_steps(core, fwd, steps=args.steps, warm=3))
    for name, core, _, m in legs:
        core.check_chain()
        print("{:48s} forward {:.4f} ms (min {:.4f}, max {:.4f})  elbo {:.4f}".format(name, float(np.median(res[name])), min(res[name]), max(res[name]), 0.0))


if __name__ == "__main__":
    main()
