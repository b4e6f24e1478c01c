"""Model / data / driver flags of the SQAIR hot path, names and defaults kept intact.

Mirrors the flag surface the reference defines at import time of its config modules
(reference: sqair/common_model_flags.py:32-56, sqair/configs/mlp_mnist_model.py:42-52,
sqair/configs/seq_mnist_data.py:28-29, sqair/scripts/experiment.py:41-69) so that a run
directory's ``flags.json`` (reference: release_models/mnist_mlp/1/flags.json) can be fed to
this package unchanged.  Only the *model* flags influence the hot path; data / driver flags
are carried so that ``flags.json`` round-trips without "unknown flag" errors.

Derived parameters follow ``get_params()`` (reference: sqair/common_model_flags.py:59-71):
``n_hidden = 32 * n_units``, two hidden layers, ``steps_pred_hidden = n_hidden // 2``
(the reference relies on Python-2 integer division there).
"""
from __future__ import annotations

import copy
import json
from types import SimpleNamespace

MODEL_FLAGS = dict(
    # common_model_flags.py:32-56
    transform_var_bias=-3.0,
    output_scale=0.25,
    scale_prior="-2",
    glimpse_size=20,
    prop_prior_step_bias=10.0,
    prop_prior_type="rnn",
    masked_glimpse=True,
    k_particles=5,
    n_steps_per_image=3,
    transition="VanillaRNN",
    time_transition="GRU",
    prior_transition="GRU",
    output_std=0.3,
    n_units=8,
    n_what=50,
    # configs/mlp_mnist_model.py:42-52
    disc_prior_type="cat",
    step_success_prob=0.75,
    disc_step_bias=1.0,
    prop_step_bias=5.0,
    sample_from_prior=False,
    rec_where_prior=True,
    # not a reference flag: the `generate_after` constructor argument of SequentialAIR (seq.py:46, :198-200), which the
    # shipped config never sets; frames t > generate_after are generated from the priors when sample_from_prior is on
    generate_after=-1,
)

DATA_FLAGS = dict(train_path="seq_mnist_train.pickle", valid_path="seq_mnist_validation.pickle",
                  seq_len=0, stage_itr=0)

DRIVER_FLAGS = dict(
    data_config="configs/orig_seq_mnist.py", model_config="configs/mlp_mnist_model.py",
    results_dir="../checkpoints", run_name="test_run", batch_size=32,
    log_itr=int(1e4), report_loss_every=int(1e3), save_itr=int(1e5), fig_itr=int(1e4),
    train_itr=int(2e6), resume=False, log_at_start=False, eval_on_train=True,
    eval_size_fraction=1.0, opt="rmsprop", learning_rate=1e-5, l2=0.0, schedule="4,6,10",
    test_run=False, gpu="0", debug=False,
)

_ALL = {}
_ALL.update(MODEL_FLAGS)
_ALL.update(DATA_FLAGS)
_ALL.update(DRIVER_FLAGS)


class Flags(SimpleNamespace):
    """Attribute bag of flags; unknown names are rejected like the reference's parser does
    (reference: sqair/experiment_tools.py:230-233 asserts every CLI flag is consumed)."""

    def update(self, **kw):
        for k, v in kw.items():
            if k not in _ALL:
                raise ValueError("unknown flag '{}'".format(k))
            setattr(self, k, type(_ALL[k])(v) if not isinstance(_ALL[k], bool) else bool(v))
        return self

    def to_json(self):
        return json.dumps(self.__dict__, sort_keys=True)


FLAGS = Flags(**copy.deepcopy(_ALL))


def make_flags(**overrides):
    f = Flags(**copy.deepcopy(_ALL))
    f.update(**overrides)
    return f


def load_flags_json(path, base=None):
    """Reads a reference run's flags.json, ignoring the stale keys it is known to carry
    (reference: release_models/mnist_mlp/1/flags.json has ``constant_prop_prior``,
    ``input_type``, ``per_timestep_vimco`` with no definition in code)."""
    with open(path) as fh:
        d = json.load(fh)
    f = base if base is not None else make_flags()
    for k, v in d.items():
        if k in _ALL:
            f.update(**{k: v})
    return f


def parse_string_flag(flag, num_elements=-1):
    """reference: sqair/configs/mlp_mnist_model.py:55-71."""
    values = [float(s.strip()) for s in str(flag).split(",")]
    if len(values) == 1 and num_elements > 1:
        values = values * num_elements
    elif num_elements != -1 and len(values) != num_elements:
        raise ValueError('Incorrect number of elements in flag "{}"'.format(flag))
    return values


def get_params(F=None):
    """reference: sqair/common_model_flags.py:59-71."""
    F = F if F is not None else FLAGS
    n_hidden = 32 * int(F.n_units)
    return SimpleNamespace(
        glimpse_size=[int(F.glimpse_size)] * 2,
        n_hidden=n_hidden,
        n_layers=2,
        n_hiddens=[n_hidden] * 2,
        steps_pred_hidden=[n_hidden // 2],
    )
