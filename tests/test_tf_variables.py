"""The reference's only recorded ground truth on this path — the variable listing of the shipped MLP-SQAIR config and the
validation record of the released checkpoint (notebooks/play.ipynb:239-362, :480; extracted into
tests/golden/tf_variables.json by tests/golden/make_tf_variables.py) — against the parameter inventory and the checkpoint
interchange.  No GPU."""
import json
import os

import numpy as np
import pytest

from sqair_amd import checkpoint as ck
from sqair_amd.flags import MODEL_FLAGS, make_flags
from sqair_amd.params import flatten_params, init_params, param_spec

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tf_variables.json")))
HW = (50, 50)


def _flags_from_listing():
    kw = {}
    for k, v in GOLD["flags"].items():
        if k in MODEL_FLAGS:
            d = MODEL_FLAGS[k]
            kw[k] = (v == "True") if isinstance(d, bool) else type(d)(v)
    return make_flags(**kw)


def test_every_tf_variable_name_and_shape_matches_the_reference_listing():
    F = _flags_from_listing()
    assert int(F.n_steps_per_image) == 3 and int(F.k_particles) == 5
    want = {v["name"]: tuple(v["shape"]) for v in GOLD["variables"]}
    assert len(want) == 105
    got = {tf: ck.tf_shape(name, shape) for name, shape, init, tf in param_spec(F, HW)}
    assert len(got) == 105
    assert sorted(got) == sorted(want), (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
    for n in want:
        assert got[n] == want[n], (n, got[n], want[n])
    total = sum(int(np.prod(s)) if len(s) else 1 for s in got.values())
    assert total == GOLD["total"] == 2951522
    # per-scope totals as the reference prints them (its listing prints `sequence` twice: 14 848 before the last variable)
    by_scope = {}
    for n, s in got.items():
        by_scope[n.split("/")[0]] = by_scope.get(n.split("/")[0], 0) + (int(np.prod(s)) if len(s) else 1)
    for scope in ("decoder", "discovery", "model", "propagation"):
        assert by_scope[scope] == GOLD["scope_totals"][scope], scope
    assert by_scope["sequence"] == GOLD["scope_totals"]["sequence"] + 256 * 256


def test_reference_flag_defaults_match_the_listing():
    # every model flag printed by the notebook equals this package's default, except the ones the notebook's run changed
    # (flags.json of the released run: batch_size etc. are driver flags; seq_len / stage_itr come from the data config)
    F = make_flags()
    for k, v in GOLD["flags"].items():
        if k in MODEL_FLAGS:
            d = getattr(F, k)
            if isinstance(d, bool):
                assert d == (v == "True"), k
            elif isinstance(d, (int, float)):
                assert float(d) == float(v), k
            else:
                assert str(d) == v or float(d) == float(v), k


def test_a_dict_keyed_by_the_reference_names_round_trips():
    F = _flags_from_listing()
    rng = np.random.default_rng(0)
    tfd = {v["name"]: rng.standard_normal(v["shape"]).astype(np.float32) for v in GOLD["variables"]}
    P = ck.from_tf_dict(tfd, F, HW)            # what INTEGRATION.md's dump script would hand over
    spec = param_spec(F, HW)
    assert flatten_params(P, spec).shape == (GOLD["total"],)
    back = ck.to_tf_dict(P, F, HW)
    assert sorted(back) == sorted(tfd)
    for n in tfd:
        assert back[n].shape == tfd[n].shape and np.array_equal(back[n], tfd[n]), n
    # exact shapes are required: a transposed matrix with the right element count is refused
    bad = dict(tfd)
    n = "propagation/gru/wz"
    bad[n] = np.ascontiguousarray(tfd[n].T)
    with pytest.raises(ValueError):
        ck.from_tf_dict(bad, F, HW)
    # the mean image alone may come with or without TF's trailing channel axis
    ok = dict(tfd)
    ok["decoder/air_decoder/Variable"] = tfd["decoder/air_decoder/Variable"][..., 0]
    assert np.array_equal(ck.from_tf_dict(ok, F, HW)["dec.mean_img"], P["dec.mean_img"])
    del bad[n]
    with pytest.raises(KeyError):
        ck.from_tf_dict(bad, F, HW)


def test_validation_record_normalisation():
    """Pins WHICH of the record's numbers are per frame and which per sequence (model.py:88-135, :202-205): with
    T = 10, data_ll - kl (importance-weighted per-frame means) must sit at elbo_iwae / T, kl = log_q - log_p, the
    step counts add up, and everything stays below the perfect-reconstruction bound 2500 (-ln .3 - .5 ln 2 pi)."""
    r, T = GOLD["validation_record"], GOLD["record_normalisation"]["seq_len"]
    assert abs(r["kl"] - (r["log_q_z_given_x"] - r["log_p_z"])) < 2e-3
    assert abs(r["num_steps/t"] - (r["num_disc_steps/t"] + r["num_prop_steps/t"])) < 2e-4
    assert abs((r["data_ll"] - r["kl"]) - r["elbo_iwae"] / T) < 0.5
    assert r["elbo_vae"] <= r["elbo_iwae"]
    bound = 2500 * (-np.log(0.3) - 0.5 * np.log(2 * np.pi))
    assert r["elbo_iwae"] / T < r["data_ll"] < bound
