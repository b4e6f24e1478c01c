"""tools/eval_checkpoint.py — the turnkey TF-checkpoint evaluator (oracle-vs-TF pin route): fed a
synthetic `.npz` keyed by all 105 TF variable names of the reference's listing (tests/golden/tf_variables.json) and a
synthetic dataset pickle in the reference's layout.  CPU: oracle leg only; GPU: HIP and oracle on the same batches."""
import importlib.util
import json
import os

import numpy as np
import pytest

from sqair_amd.data import make_sequences
from sqair_amd.dataio import save_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "tf_variables.json")))


def _tool():
    spec = importlib.util.spec_from_file_location("eval_checkpoint", os.path.join(ROOT, "tools", "eval_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _synthetic_inputs(tmp_path, n_seq=8, T=3):
    rng = np.random.default_rng(0)
    tf_vars = {}
    for v in GOLD["variables"]:
        shape = tuple(v["shape"])
        fan_in = shape[0] if len(shape) == 2 else 1
        tf_vars[v["name"]] = (rng.standard_normal(shape) * (0.5 / np.sqrt(max(fan_in, 1)))).astype(np.float32)
    assert len(tf_vars) == 105 and sum(int(a.size) for a in tf_vars.values()) == 2951522
    tf_vars["global_step"] = np.asarray(1000000)                       # extra entries of a real dump are ignored
    tf_vars[GOLD["variables"][0]["name"] + "/RMSProp"] = np.zeros(3)
    npz = str(tmp_path / "sqair_tf_vars.npz")
    np.savez(npz, **tf_vars)
    d = make_sequences(n_seq, T=T, canvas=(50, 50), n_objects=(0, 2), seed=5)
    pkl = str(tmp_path / "seq_mnist_validation.pickle")
    save_dataset(pkl, d)
    return npz, pkl


def test_evaluator_oracle_leg_on_a_synthetic_tf_dump(tmp_path):
    npz, pkl = _synthetic_inputs(tmp_path)
    out = _tool().evaluate(npz, pkl, batches=2, batch_size=4, oracle="fp32", hip=False)
    assert out["k_particles"] == 5 and out["n_steps_per_image"] == 3 and out["seq_len"] == 3
    for m, v in out["metrics"].items():
        assert v["hip"] is None and np.isfinite(v["oracle"]), m
    rec = out["metrics"]
    assert rec["elbo_iwae"]["recorded"] == 6095.4565 and rec["elbo_iwae"]["normalisation"] == "per sequence"
    assert rec["data_ll"]["recorded"] == 640.4481 and rec["data_ll"]["normalisation"] == "per frame"
    assert rec["num_steps"]["recorded"] == 1.0953 and rec["num_step_accuracy"]["normalisation"] == "fraction"
    # a dump with a missing variable is refused (strict), not silently initialised
    with np.load(npz) as z:
        broken = {k: z[k] for k in z.files if not k.endswith("air_decoder/Variable")}
    np.savez(str(tmp_path / "broken.npz"), **broken)
    with pytest.raises(KeyError):
        _tool().evaluate(str(tmp_path / "broken.npz"), pkl, batches=1, batch_size=4, oracle="none", hip=False)


@pytest.mark.gpu
def test_evaluator_hip_and_oracle_agree_on_a_synthetic_tf_dump(tmp_path):
    npz, pkl = _synthetic_inputs(tmp_path)
    out = _tool().evaluate(npz, pkl, batches=2, batch_size=4, oracle="fp64", hip=True)
    hv = out["hip_vs_oracle"]
    if hv["identical_presence"]:
        assert hv["log_weights_max_rel_err"] <= 1e-4
        for m, v in out["metrics"].items():
            assert abs(v["hip"] - v["oracle"]) <= 1e-4 * max(abs(v["oracle"]), 1.0), m
    assert all(np.isfinite(v["hip"]) for v in out["metrics"].values())
