#!/usr/bin/env python
"""Turnkey evaluation of a TF checkpoint dump: the oracle-vs-TF pin route (and HIP-vs-oracle on the same numbers).

    python tools/eval_checkpoint.py <tf_vars.npz> <seq_mnist_validation.pickle> [--batches 2] [--batch-size 32]
                                    [--flags flags.json] [--seed 0] [--oracle fp64|fp32|none] [--no-hip] [--json out.json]

What it replaces: the reference's checkpoint evaluator, sqair/scripts/eval.py:191-224 (restore -> `sess.run` of
{elbo_iwae, elbo_vae, num_step_accuracy, data_ll, kl} over n_batches -> mean), and the notebook cell that printed the released
model's validation record (notebooks/play.ipynb:480: 2 batches x 32 sequences of the validation pickle, K = 5):
elbo_iwae 6095.4565, elbo_vae 5941.5574, data_ll 640.4481, kl 30.7866, log_p_z 16.1933, log_q_z_given_x 46.9800,
num_steps/t 1.0953 (disc 0.1678, prop 0.9275), num_steps_acc 0.9453.

Inputs
  tf_vars.npz   every variable of a `tf.train.Saver` checkpoint by its TF name — produced in the reference's own environment
                by the three-line dump of INTEGRATION.md section 4 (`tf.train.NewCheckpointReader`); optimiser slots are ignored.
  pickle        a dataset in the reference's layout (sqair/data/data.py:189-201), e.g. seq_mnist_validation.pickle written by
                sqair/data/create_seq_mnist.py.  Evaluation batches are the first `batches` consecutive full batches, the
                order the reference's non-shuffled feed visits them (sqair/data/data.py:203-245).
  --flags       the run's flags.json (release_models/mnist_mlp/1/flags.json); default: the released run's model flags.

What it prints: per metric the HIP path, the oracle (fp64 by default) on the SAME batches and noise, their relative
difference, and the notebook's recorded value with its normalisation (per sequence: elbo_*; per frame: the rest —
tests/golden/tf_variables.json).  The recorded numbers are Monte-Carlo estimates under TF's own random draws: agreement
with them is statistical (a fraction of a nat per frame over 64 sequences), agreement between HIP and oracle is exact to 1e-4.
Exit status 1 if HIP and oracle disagree beyond 1e-4 relative on a batch where their presence decisions are identical.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRICS = ["elbo_iwae", "elbo_vae", "data_ll", "kl", "log_p_z", "log_q_z_given_x", "num_steps", "num_disc_steps", "num_prop_steps",
           "num_step_accuracy"]
RECORD_KEY = {"num_steps": "num_steps/t", "num_disc_steps": "num_disc_steps/t", "num_prop_steps": "num_prop_steps/t",
              "num_step_accuracy": "num_steps_acc"}


def evaluate(tf_vars_path, data_path, batches=2, batch_size=32, flags_json=None, seed=0, oracle="fp64", hip=True, device="cuda:0",
             seq_len=None):
    import torch
    from sqair_amd import checkpoint as ck
    from sqair_amd.dataio import MinibatchFeed, load_dataset, process_data
    from sqair_amd.flags import load_flags_json, make_flags
    F = load_flags_json(flags_json) if flags_json else make_flags(n_steps_per_image=3, k_particles=5)
    data = process_data(load_dataset(data_path), n_timesteps=seq_len)
    hw = tuple(int(v) for v in data["imgs"].shape[2:4])
    feed = MinibatchFeed(data, batch_size, shuffle=False)
    with np.load(tf_vars_path) as z:
        tf_vars = {k: z[k] for k in z.files}
    P = {k: np.asarray(v, dtype=np.float32) for k, v in ck.from_tf_dict(tf_vars, F, hw, strict=True).items()}
    K, N = int(F.k_particles), int(F.n_steps_per_image)
    nzw = 4 + int(F.n_what) + 1
    model = orc = None
    if hip:
        from sqair_amd.model import Model, SqairCore
        core = SqairCore(F, hw, device=device)
        core.set_params(P)
    if oracle != "none":
        from oracle import sqair_oracle as O   # the checker: this tool is test infrastructure, like tests/ and bench.py's cpu leg
        orc = O.SqairOracle(P, O.make_cfg(F, hw), torch.float64 if oracle == "fp64" else torch.float32)
    acc = {"hip": {m: [] for m in METRICS}, "oracle": {m: [] for m in METRICS}}
    worst_rel, same_all = 0.0, True
    for b in range(batches):
        batch = feed.next()
        obs, nums = batch["imgs"], batch["nums"]
        T = obs.shape[0]
        rng = np.random.default_rng(seed + b)
        noise = rng.standard_normal((T, batch_size * K, 2, N, nzw)).astype(np.float32)
        noise[..., -1] = rng.uniform(size=noise.shape[:-1]).astype(np.float32)
        ru = rng.uniform(size=batch_size).astype(np.float32)
        if hip:
            if model is None:
                model = Model(obs, None, core, K, presence=nums, debug=True)
            else:
                model.rebind(obs, presence=nums)
            model.run(noise=noise, resample_u=ru)
            for m in METRICS:
                acc["hip"][m].append(float(getattr(model, m)))
        if orc is not None:
            with torch.no_grad():
                ref = orc.model(obs, noise, num=nums, resample_u=ru)
            for m in METRICS:
                acc["oracle"][m].append(float(getattr(ref, m)))
            if hip:
                same = np.array_equal(model.presence.cpu().numpy(), ref.presence.numpy().astype(np.float32))
                same_all = same_all and same
                if same:
                    a, r = model.log_weights.cpu().numpy().astype(np.float64), ref.log_weights.numpy().astype(np.float64)
                    worst_rel = max(worst_rel, float(np.abs(a - r).max() / np.abs(r).max()))
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "tf_variables.json")))
    record, norm = rec["validation_record"], rec["record_normalisation"]
    out = dict(tf_vars=os.path.basename(tf_vars_path), dataset=os.path.basename(data_path), batches=batches, batch_size=batch_size,
               k_particles=K, n_steps_per_image=N, seq_len=int(data["imgs"].shape[0]), img_hw=list(hw), seed=seed,
               hip_vs_oracle=dict(identical_presence=bool(same_all), log_weights_max_rel_err=worst_rel) if (hip and orc is not None) else None,
               metrics={})
    for m in METRICS:
        key = RECORD_KEY.get(m, m)
        out["metrics"][m] = dict(
            hip=float(np.mean(acc["hip"][m])) if acc["hip"][m] else None,
            oracle=float(np.mean(acc["oracle"][m])) if acc["oracle"][m] else None,
            recorded=record.get(key),
            normalisation="per sequence" if key in norm["per_sequence"] else ("fraction" if key in norm["fraction"] else "per frame"))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("tf_vars")
    ap.add_argument("dataset")
    ap.add_argument("--batches", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--flags", default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--seq-len", type=int, default=None, help="truncate the sequences to their first frames (mnist_tools.py:40-58)")
    ap.add_argument("--oracle", default="fp64", choices=["fp64", "fp32", "none"])
    ap.add_argument("--no-hip", action="store_true", help="oracle only (a machine without an MI355X)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    out = evaluate(a.tf_vars, a.dataset, a.batches, a.batch_size, a.flags, a.seed, a.oracle, not a.no_hip, a.device, a.seq_len)
    print("{} on {}: {} batches x {} sequences, T = {}, K = {}, N = {}".format(
        out["tf_vars"], out["dataset"], out["batches"], out["batch_size"], out["seq_len"], out["k_particles"], out["n_steps_per_image"]))
    print("{:<20s} {:>14s} {:>14s} {:>10s} {:>12s}  {}".format("metric", "HIP", "oracle", "rel.diff", "notebook", "normalisation"))
    for m, v in out["metrics"].items():
        rd = "" if v["hip"] is None or v["oracle"] is None else "{:.1e}".format(abs(v["hip"] - v["oracle"]) / max(abs(v["oracle"]), 1e-30))
        fmt = lambda x: "-" if x is None else "{:.4f}".format(x)  # noqa: E731
        print("{:<20s} {:>14s} {:>14s} {:>10s} {:>12s}  {}".format(m, fmt(v["hip"]), fmt(v["oracle"]), rd, fmt(v["recorded"]), v["normalisation"]))
    hv = out["hip_vs_oracle"]
    rc = 0
    if hv is not None:
        print("HIP vs oracle: identical presence decisions on every batch: {}; sequence log-weights max rel err {:.2e} (bar 1e-4)".format(
            hv["identical_presence"], hv["log_weights_max_rel_err"]))
        rc = 1 if hv["log_weights_max_rel_err"] > 1e-4 else 0
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
    sys.exit(rc)


if __name__ == "__main__":
    main()
