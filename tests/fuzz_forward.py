"""Randomised configuration sweep of the forward pass against the live oracle (a checker script, not collected by pytest: it
lives under tests/ because it calls the oracle).  On the GPU box:

    python tests/fuzz_forward.py [n_cases] [seed] [--grad | --chain | --gen]

Every case draws sizes (particles, slots, frames, sequences, frame shape, n_what, n_units), cells, priors and the boolean model
flags at random inside the library's limits, a decision-stable noise draw on the ORACLE's margin (tests/hip_util.stable_noise),
and compares presence / ids exactly, every output at 5e-4 scaled, the bounds at 1e-4 relative; with --grad also every
parameter's gradient against autograd through the fp64 oracle (tests/test_hip_backward._full_backward_case's bar).  With --chain
the sizes are drawn inside what the in-launch slot chain takes (VanillaRNN slot cell, GRU temporal cell, n_units 8) and the
pass with `slot_chain` on must reproduce the launches bit for bit, eager and as a graph replay.  With --gen the generation
modes (`sample_from_prior`, frames after a random `generate_after` drawn from the priors: seq.py:198-200,
sqair_modules.py:157-170, 294-302) with a second noise tensor, the draw chosen on the oracle's posterior AND prior margins.
Round 5: 100 forward cases
(seeds 1, 3) without a failure; 30 gradient cases (seed 2) with one beyond the tight bar, on parameters whose gradient is 3e-4 of
the largest one, by exactly what fp32 autograd through the oracle misses the fp64 one (run_case's second bar).  Prints one
line per case and the failures with their configuration; exit code = number of failures."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sqair_amd.data import make_sequences, to_float  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from tests.hip_util import params32, run_hip, stable_noise  # noqa: E402
from tests.test_hip_forward import _check_against  # noqa: E402


def draw_case(rng):
    cells = ["VanillaRNN", "GRU", "LSTM"]
    K = int(rng.choice([1, 2, 3, 5, 7, 16]))
    N = int(rng.integers(1, 9))
    T = int(rng.integers(1, 5))
    B = int(rng.integers(1, 8))
    hw = (int(rng.integers(8, 90)), int(rng.integers(8, 90)))
    flags = dict(k_particles=K, n_steps_per_image=N,
                 n_what=int(rng.choice([50, 50, 10, 33, 64])), n_units=int(rng.choice([8, 8, 4, 2, 6, 12])),
                 transition=str(rng.choice(cells, p=[0.6, 0.2, 0.2])), time_transition=str(rng.choice(cells, p=[0.2, 0.6, 0.2])),
                 prior_transition=str(rng.choice(cells, p=[0.2, 0.6, 0.2])),
                 prop_prior_type=str(rng.choice(["rnn", "rw", "guided"])), disc_prior_type=str(rng.choice(["cat", "geom"])),
                 rec_where_prior=bool(rng.integers(2)), masked_glimpse=bool(rng.integers(2)),
                 glimpse_size=int(rng.choice([20, 20, 12, 8])))
    return flags, hw, T, B


def run_chain_case(flags, hw, T, B, seed):
    from tests.test_slot_chain import _inputs, _run
    K, N = flags["k_particles"], flags["n_steps_per_image"]
    extra = {k: v for k, v in flags.items() if k not in ("k_particles", "n_steps_per_image")}
    F, d, obs, P, noise = _inputs(B, K, N, T, hw, seed=seed, obj_size=max(2, min(20, min(hw) // 2)), **extra)
    _, _, ref = _run(F, hw, d, obs, P, noise, K, chain=False, use_graph=False)
    for use_graph in (False, True):
        _, _, got = _run(F, hw, d, obs, P, noise, K, chain=True, use_graph=use_graph)   # (debug on: every chain launch's status is checked)
        for k, v in ref.items():
            assert np.array_equal(v, got[k], equal_nan=True), (k, use_graph)


def run_gen_case(flags, hw, T, B, seed, rng):
    import torch
    from oracle import sqair_oracle as O
    from sqair_amd.model import Model, SqairCore
    from tests.hip_util import MARGIN, MAX_DRAWS, draw_noise, presence_margins, prior_presence_margins
    T = max(T, 2)
    flags = dict(flags, sample_from_prior=True, generate_after=int(rng.integers(-1, T)))
    if flags["prop_prior_type"] == "rw":
        flags["rec_where_prior"] = False
    F = make_flags(**flags)
    K, N, nzw = int(F.k_particles), int(F.n_steps_per_image), 4 + int(F.n_what) + 1
    d = make_sequences(B, T=T, canvas=hw, n_objects=(1, 2), obj_size=max(2, min(20, min(hw) // 2)), seed=seed)
    obs = to_float(d["imgs"])
    P = params32(F, hw, seed, 0.05, obs.mean((0, 1)))
    orc = O.SqairOracle(P, O.make_cfg(F, hw), torch.float64)
    for attempt in range(MAX_DRAWS):
        r2 = np.random.default_rng(seed * 7 + attempt)
        noise, gen_noise = draw_noise(r2, T, B * K, N, nzw), draw_noise(r2, T, B * K, N, nzw)
        with torch.no_grad():
            ref = orc.model(obs, noise, num=d["nums"], gen_noise=gen_noise)
        mg = min(float(presence_margins(ref.outputs, noise).min()), float(prior_presence_margins(ref.outputs, gen_noise).min()))
        if mg >= MARGIN:
            break
    else:
        raise AssertionError("no decision-stable noise draw")
    core = SqairCore(F, hw)
    core.set_params(P)
    m = Model(obs, None, core, K, presence=d["nums"])
    m.run(noise=noise, gen_noise=gen_noise)
    for k in ("prop_pres", "disc_pres", "presence", "obj_id"):
        assert np.array_equal(getattr(m, k).cpu().numpy(), getattr(ref, k).numpy().astype(np.float32)), (k, flags["generate_after"])
    for k, v in core.out.items():
        name = "_" + k if k.startswith("final_") else k
        if name not in ref.outputs:
            continue
        want = ref.outputs[name].numpy()
        err = np.abs(v.cpu().numpy().reshape(want.shape) - want).max() / max(np.abs(want).max(), 1.0)
        assert err <= 5e-4, (k, err, flags["generate_after"])
    assert abs(float(m.elbo_iwae) - float(ref.elbo_iwae)) <= 1e-4 * max(1.0, abs(float(ref.elbo_iwae)))


def run_case(flags, hw, T, B, seed, grad):
    F = make_flags(**flags)
    K, N = int(F.k_particles), int(F.n_steps_per_image)
    if grad:
        from tests.test_hip_backward import _check_report, _full_backward_case
        extra = {k: v for k, v in flags.items() if k not in ("k_particles", "n_steps_per_image")}
        report, _, _ = _full_backward_case(max(K, 2), N, T, B, hw, seed=seed, flags=extra)
        try:
            _check_report(report, ill_scale=True)
        except AssertionError:
            # Second bar for parameters whose gradient is tiny against the pass's largest one (sums of cancelling terms: the fp32
            # ORACLE misses the fp64 one by the same amount there -- seed 2 case 11: prop.where_bias.l1.b, |grad| 1.7e-2 of 54,
            # error 7.2e-5 on the HIP path and 7.2e-5 in fp32 autograd): error against 1e-3 of the largest gradient.
            gmax = max(s for _, _, s in report)
            bad = [(n, e, s) for n, e, s in report if not np.isfinite(e) or e > 5e-4 * max(s, 1e-3 * gmax)]
            if bad:
                raise
            print("     (small gradients beyond the tight bar, within 5e-7 of the largest gradient: fp32 cancellation)")
        return
    d = make_sequences(B, T=T, canvas=hw, n_objects=(0, 2), obj_size=max(2, min(20, min(hw) // 2)), seed=seed)
    obs = to_float(d["imgs"])
    P = params32(F, hw, seed, 0.05, obs.mean((0, 1)))
    noise, ref, _, _ = stable_noise(F, hw, P, obs, T, B * K, N, seed0=seed, nums=d["nums"], nzw=4 + int(F.n_what) + 1)
    m = run_hip(F, hw, P, obs, noise, nums=d["nums"])
    ref_out = {k: v.numpy() for k, v in ref.outputs.items() if not k.startswith("_")}
    ref_model = {k: getattr(ref, k).numpy() for k in ("log_weights", "elbo_iwae_per_example", "elbo_vae", "elbo_iwae",
                                                      "data_ll", "kl", "log_p_z", "log_q_z_given_x")}
    _check_against(m, ref_out, ref_model, list(ref_out), T)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    grad, chain, gen = "--grad" in sys.argv, "--chain" in sys.argv, "--gen" in sys.argv
    n, seed = (int(args[0]) if args else 30), (int(args[1]) if len(args) > 1 else 0)
    rng = np.random.default_rng(seed)
    failures = []
    for i in range(n):
        flags, hw, T, B = draw_case(rng)
        if chain:
            flags.update(transition="VanillaRNN", time_transition="GRU", n_units=8, n_what=min(flags["n_what"], 50))
            B = int(rng.integers(1, 65))
            while B * flags["k_particles"] > 320:
                B = max(1, B // 2)
        tag = "case {:3d}: hw {} T {} B {} {}".format(i, hw, T, B, flags)
        try:
            if chain:
                run_chain_case(flags, hw, T, B, seed * 1000 + i)
            elif gen:
                run_gen_case(flags, hw, T, B, seed * 1000 + i, rng)
            else:
                run_case(flags, hw, T, B, seed * 1000 + i, grad)
            print("ok   " + tag, flush=True)
        except AssertionError as e:
            if "no decision-stable noise draw" in str(e):
                print("skip " + tag + " (no decision-stable draw)", flush=True)
                continue
            failures.append((tag, traceback.format_exc(limit=3)))
            print("FAIL " + tag, flush=True)
        except Exception:
            failures.append((tag, traceback.format_exc(limit=4)))
            print("FAIL " + tag, flush=True)
    for tag, tb in failures:
        print("\n" + tag + "\n" + tb)
    print("{} cases, {} failures".format(n, len(failures)))
    sys.exit(len(failures))


if __name__ == "__main__":
    main()
