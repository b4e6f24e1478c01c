#!/usr/bin/env python
"""What the two eager <-> graph transitions of a forward step cost (timing probe, not a product path): the step as bench.py issues
it -- k_fill_noise (eager), the forward graph, k_elbo (eager) -- against the SAME three pieces captured into ONE graph through the
generic capture slots (the noise key is then frozen at capture time: every replay draws the same noise, which is why this is a
probe and not the step).      python tools/step_graph_probe.py [--cfg 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=6)
    args = ap.parse_args()
    import numpy as np
    from sqair_amd import timeline as TL
    from sqair_amd.data import config_inputs
    from sqair_amd.flags import make_flags
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.params import init_params
    ov, obs, nums, _ = config_inputs(args.cfg)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    B, K = int(obs.shape[1]), int(F.k_particles)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    core = SqairCore(F, hw)
    with core.on_stream():
        core.set_params(P)
        Model(obs, None, core, K, presence=nums, outputs="minimal")
        n = [0]

        def plain():
            core.draw_noise(seed=1000, step=n[0], global_batch=B, b0=0)
            n[0] += 1
            core.forward(use_graph=True)
        plain()
        core.stream.synchronize()
        lib, h, s = core.lib, core.handle, core._stream()
        # the same calls, eager launch sequence of the forward pass instead of its graph, inside one capture
        core.check(lib.sqair_capture_begin(h, s), "sqair_capture_begin")
        core.check(lib.sqair_fill_noise(h, core.noise.data_ptr(), core.T, core.B, B, 0, 1000, 7, s), "sqair_fill_noise")
        core.check(lib.sqair_forward(*core._args(0)), "sqair_forward")
        dlp = core.out["discrete_log_prob"].data_ptr() if "discrete_log_prob" in core.out else None
        core.check(lib.sqair_elbo(h, core.out["log_weights_per_timestep"].data_ptr(), dlp, core.T, core.B, core.log_weights.data_ptr(),
                                  core.elbo_iwae_per_example.data_ptr(), core.importance_weights.data_ptr(), core.vimco_signal.data_ptr(),
                                  core.scalars.data_ptr(), core.c_means, len(core.mean_names), core.iw_means.data_ptr(), s), "sqair_elbo")
        nodes = lib.sqair_capture_end(h, s, 2)
        assert nodes > 0, nodes

        def one():
            core.check(lib.sqair_capture_launch(h, 2, s), "sqair_capture_launch")
        res = {"three pieces (bench.py's step)": [], "one graph ({} nodes)".format(nodes): []}
        for _ in range(args.rounds):
            for name, fn in zip(res, (plain, one)):
                res[name].append(TL.time_steps(core, fn, steps=40, warm=3))
    for name, v in res.items():
        print("{:36s} {:.4f} ms per step (min {:.4f}, max {:.4f})".format(name, float(np.median(v)), min(v), max(v)))


if __name__ == "__main__":
    main()
