"""Generates the golden fixtures of tests/golden/*.npz with the fp64 CPU oracle.

The reference cannot be executed here (Python 2 / TF1 / Sonnet), so these vectors pin
HIP-vs-oracle parity, not oracle-vs-TF.  Inputs are stored in float32 (what the GPU consumes); the
oracle is evaluated in float64 on exactly those float32 values.  Parameters are regenerated from
(seed, jitter) by sqair_amd.params.init_params and guarded by a checksum.  Noise draws whose smallest
|u - p| over the live presence Bernoullis is below MARGIN are rejected: a flipped Bernoulli changes the
ELBO by O(1..100) nats and no tolerance survives that (hard parts).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import sqair_oracle as O  # noqa: E402
from sqair_amd.data import make_sequences, to_float  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.params import flatten_params, init_params, param_spec  # noqa: E402

MARGIN = 2e-3
HERE = os.path.dirname(os.path.abspath(__file__))


def params32(F, hw, seed, jitter, mean_img=None):
    P = init_params(F, hw, seed=seed, mean_img=mean_img, jitter=jitter)
    return {k: np.asarray(v, dtype=np.float32) for k, v in P.items()}


def checksum(P, spec):
    return hashlib.sha256(flatten_params(P, spec).tobytes()).hexdigest()


def presence_margin(o, noise, K_rows):
    """min |u - p| over Bernoullis whose previous presence was 1 (others are deterministic)."""
    u_prop = noise[:, :, 0, :, -1]
    u_disc = noise[:, :, 1, :, -1]
    p_prop = o["_prop_presence_prob"].numpy()
    p_disc = o["_disc_presence_prob"].numpy()
    live_prop = o["_prop_prev_presence"].numpy() > 0.5
    dp = o["disc_pres"].numpy()
    live_disc = np.concatenate([np.ones_like(dp[..., :1]), dp[..., :-1]], -1) > 0.5
    m = 1.0
    if live_prop.any():
        m = min(m, float(np.abs(u_prop - p_prop)[live_prop].min()))
    if live_disc.any():
        m = min(m, float(np.abs(u_disc - p_disc)[live_disc].min()))
    return m


def draw_noise(rng, T, R, N, nzw):
    nz = rng.standard_normal((T, R, 2, N, nzw)).astype(np.float32)
    nz[..., -1] = rng.uniform(size=nz.shape[:-1]).astype(np.float32)
    return nz


def make_fixture(name, F, hw, T, B, data_seed, param_seed, jitter, n_objects=(0, 2), obj_size=28, full=True):
    cfg = O.make_cfg(F, hw)
    spec = param_spec(F, hw)
    d = make_sequences(B, T=T, canvas=hw, n_objects=n_objects, obj_size=obj_size, seed=data_seed)
    obs = to_float(d["imgs"])
    mean_img = obs.mean((0, 1)).astype(np.float32)
    P = params32(F, hw, param_seed, jitter, mean_img)
    orc = O.SqairOracle(P, cfg, torch.float64)
    R = B * cfg.K
    for attempt in range(200):
        rng = np.random.default_rng(1000 * data_seed + attempt)
        noise = draw_noise(rng, T, R, cfg.N, O.noise_width(cfg))
        ru = rng.uniform(size=B).astype(np.float32)
        with torch.no_grad():
            m = orc.model(obs, noise, num=d["nums"], resample_u=ru)
        mg = presence_margin(m.outputs, noise, R)
        if mg >= MARGIN:
            break
    else:
        raise RuntimeError("no decision-stable noise draw found")
    with torch.no_grad():
        target = orc.make_target(m)
    out = dict(obs=obs, noise=noise, resample_u=ru, nums=d["nums"], mean_img=mean_img,
               meta=np.array([T, B, cfg.K, cfg.N, hw[0], hw[1], param_seed, data_seed], dtype=np.int64),
               jitter=np.float64(jitter), margin=np.float64(mg), params_sha256=np.array(checksum(P, spec)))
    names = [k for k in m.outputs if not k.startswith("_")] if full else \
        ["presence", "obj_id", "log_weights_per_timestep", "discrete_log_prob", "data_ll_per_sample", "kl_per_sample",
         "num_steps_per_sample", "where", "prop_pres", "disc_pres"]
    for k in names:
        out["out_" + k] = m.outputs[k].numpy().astype(np.float64)
    out["out_final_temporal_state"] = m.outputs["_final_temporal_state"].numpy()
    out["out_final_prior_state"] = m.outputs["_final_prior_state"].numpy()
    out["out_final_last_used_id"] = m.outputs["_final_last_used_id"].numpy().reshape(-1)
    for k in ("log_weights", "elbo_vae", "elbo_iwae_per_example", "elbo_iwae", "importance_weights", "ess", "data_ll",
              "log_p_z", "log_q_z_given_x", "kl", "mse", "raw_mse", "num_steps", "num_disc_steps", "num_prop_steps",
              "num_step_accuracy", "raw_num_step_accuracy", "iw_resampling_idx"):
        out["model_" + k] = np.asarray(getattr(m, k).numpy(), dtype=np.float64)
    out["model_vimco_target"] = np.float64(target.item())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("{}: margin {:.4f} attempt {} elbo_iwae {:.4f} elbo_vae {:.4f} target {:.4f} presence-sum {} size {:.0f} KB".format(
        name, mg, attempt, float(m.elbo_iwae), float(m.elbo_vae), float(target), float(m.presence.sum()),
        os.path.getsize(path) / 1024.0))


def grad_digest(g):
    """What the gradient fixture keeps of one parameter's gradient (fp64): every element of small tensors, 256 evenly strided
    elements of large ones, plus sum / sum of absolute values / L2 norm / max of absolute values over ALL elements."""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    idx = np.arange(g.size) if g.size <= 4096 else np.linspace(0, g.size - 1, 256).astype(np.int64)
    return idx, g[idx], np.array([g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum()), np.abs(g).max()])


def make_grad_fixture(name):
    """Gradients of the VIMCO target (model.py:150-168) w.r.t. every trainable variable for an EXISTING forward fixture (same
    frames, parameters, noise), by autograd through the fp64 oracle -> <name>_grads.npz (digest form, see grad_digest)."""
    z = np.load(os.path.join(HERE, name + ".npz"))
    T, B, K, N, H, W, pseed, _ = [int(v) for v in z["meta"]]
    F = make_flags(k_particles=K, n_steps_per_image=N)
    spec = param_spec(F, (H, W))
    P = params32(F, (H, W), pseed, float(z["jitter"]), z["mean_img"])
    assert checksum(P, spec) == str(z["params_sha256"]), "parameters regenerated from (seed, jitter) differ from the fixture's"
    orc = O.SqairOracle(P, O.make_cfg(F, (H, W)), torch.float64, requires_grad=True)
    m = orc.model(z["obs"], z["noise"], num=z["nums"], resample_u=z["resample_u"])
    assert np.array_equal(m.presence.detach().numpy(), z["out_presence"])
    target = orc.make_target(m)
    target.backward()
    out = dict(params_sha256=z["params_sha256"], vimco_target=np.float64(target.item()))
    for pname in (e[0] for e in spec):
        g = orc.P[pname].grad
        g = np.zeros(orc.P[pname].shape) if g is None else g.numpy()
        idx, vals, stats = grad_digest(g)
        out["idx/" + pname], out["val/" + pname], out["stat/" + pname] = idx, vals, stats
    path = os.path.join(HERE, name + "_grads.npz")
    np.savez_compressed(path, **out)
    print("{}_grads: target {:.6f}, {} parameters, {:.0f} KB".format(name, float(target), len(spec), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    if "--grads-only" in sys.argv:   # the forward fixtures stay as committed; only the gradient digest is (re)generated
        make_grad_fixture("k5_iwae_vimco")
        sys.exit(0)
    # cfg-1 (BASELINE.json configs[0]): T=3, B=4, K=1, N=3, 50x50 — full 38-output dump
    make_fixture("cfg1_plumbing", make_flags(k_particles=1, n_steps_per_image=3), (50, 50), T=3, B=4, data_seed=1235,
                 param_seed=0, jitter=0.05)
    # K=5 IWAE / VIMCO fixture, N=4, longer sequence so that propagation, deaths and re-discovery all occur
    make_fixture("k5_iwae_vimco", make_flags(k_particles=5, n_steps_per_image=4), (50, 50), T=5, B=3, data_seed=77,
                 param_seed=1, jitter=0.05, full=False)
    # 128x128 frames (cfg-5 shape family): stresses the LDS-staged crop / insert path
    make_fixture("hw128_small", make_flags(k_particles=2, n_steps_per_image=4), (128, 128), T=2, B=2, data_seed=5,
                 param_seed=2, jitter=0.05, n_objects=(1, 2), obj_size=72, full=False)
    make_grad_fixture("k5_iwae_vimco")
