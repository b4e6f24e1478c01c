"""In-launch slot chain against the launch-per-op path: every output bit for bit, then graph-replay time of both.
    python tools/chain_check.py [B] [K] [N] [T] [H] [W]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sqair_amd.data import make_sequences, to_float  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import Model, SqairCore  # noqa: E402
from sqair_amd.params import init_params  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    B, K, N, T, H, W = (a + [32, 5, 4, 10, 50, 50][len(a):])[:6]
    F = make_flags(k_particles=K, n_steps_per_image=N)
    hw = (H, W)
    d = make_sequences(B, T=T, canvas=hw, seed=3)
    obs = to_float(d["imgs"])
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
    rng = np.random.default_rng(0)
    nzw = 4 + int(F.n_what) + 1
    noise = rng.standard_normal((T, B * K, 2, N, nzw)).astype(np.float32)
    noise[..., -1] = rng.uniform(size=noise.shape[:-1])
    res = {}
    for name, opts in (("launches", {}), ("chain", {"slot_chain": 1})):
        core = SqairCore(F, hw, options=opts, lib_path=(os.path.join(ROOT, "tools", "bin", os.environ["KNOBS"] if os.environ.get("KNOBS", "1") != "1" else "libsqair_hip_knobs.so") if os.environ.get("KNOBS") else None))
        core.set_params(P)
        m = Model(obs, None, core, K, presence=d["nums"])
        for use_graph in (False, True):
            m.run(noise=noise, use_graph=use_graph)
            torch.cuda.synchronize()
            res[(name, use_graph)] = {k: v.detach().cpu().numpy().copy() for k, v in core.out.items()}
            res[(name, use_graph)]["elbo"] = np.array(float(m.elbo_iwae))
        with core.on_stream():
            for _ in range(5):
                core.forward(use_graph=True)
            core.stream.synchronize()
            t0 = time.perf_counter()
            reps = 50
            for _ in range(reps):
                core.forward(use_graph=True)
            core.stream.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / reps
        print("{:9s}: {:.3f} ms per forward pass (graph replay, {} nodes), elbo {:.4f}".format(
            name, ms, core.lib.sqair_graph_nodes(core.handle), float(res[(name, True)]["elbo"])), flush=True)
    ref = res[("launches", False)]
    bad = 0
    for key, r in res.items():
        if key == ("launches", False):
            continue
        worst = ("", 0.0)
        nan = 0
        for k, v in ref.items():
            if not np.array_equal(v, r[k], equal_nan=True):
                e = float(np.nanmax(np.abs(v.astype(np.float64) - r[k].astype(np.float64))))
                nan += int(np.isnan(r[k]).sum())
                if e >= worst[1] or np.isnan(e):
                    worst = (k, e)
        print("{} vs launches/eager: {}".format(key, "bit-identical" if worst[0] == "" else "DIFFERS, worst {} {:.3g}, NaNs {}".format(worst[0], worst[1], nan)))
        if worst[0] != "" and key == ("chain", False):
            for k, v in ref.items():
                if not np.array_equal(v, r[k], equal_nan=True):
                    dd = np.abs(v.astype(np.float64) - r[k].astype(np.float64))
                    first_t = int(np.argmax(dd.reshape(dd.shape[0], -1).max(1) > 0)) if dd.ndim > 1 else -1
                    print("    {:32s} max abs diff {:.3g} (values ~{:.3g}), {} of {} differ, first frame {}".format(
                        k, float(dd.max()), float(np.abs(v).max()), int((dd > 0).sum()), dd.size, first_t))
        bad += worst[0] != ""
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
