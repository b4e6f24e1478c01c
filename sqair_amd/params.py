"""Trainable-parameter inventory of MLP-SQAIR and its flat fp32 buffer layout.

The inventory reproduces, name for name and shape for shape, the variable listing the
reference prints for the shipped config (reference: notebooks/play.ipynb:239-362; total
2 951 522 at N=3, 50x50) and the wiring of sqair/configs/mlp_mnist_model.py:74-150
(discovery and propagation share the input encoder, the glimpse encoder, the mask MLP and
the what-Gaussian head, :112-113).  Each entry carries the TF variable name so a TF
checkpoint can be mapped onto the flat buffer later.

The flat buffer is what crosses the C-ABI (``sqair_bind_params``): entries in the order of
``param_spec`` below, each row-major, fp32, no padding.  ``include/sqair_hip.h`` documents
the same order; ``tests/test_tf_variables.py`` checks names, shapes and scope totals against the
reference's listing, ``tests/test_capi_host.py`` that the C library's table agrees.

Initialisers restate the defaults the reference relies on:
``snt.Linear`` w ~ TruncNormal(0, 1/sqrt(fan_in)) and b = 0 unless overridden
(sqair/modules.py:323-324,508, sqair/core.py:345, sqair/sqair_modules.py:80-83); Sonnet GRU
variables and bare ``tf.get_variable`` fall back to Glorot-uniform; trainable initial states
are zero (sqair/core.py:130, sqair/propagate.py:107).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .flags import get_params, parse_string_flag


def param_spec(F, img_hw):
    """Returns a list of (name, shape, init, tf_name).  ``init`` is one of
    ('lin_w', fan_in) | ('zeros',) | ('const', c) | ('vec', [..]) | ('glorot', fan_in, fan_out)
    | ('mean_img',)."""
    H, W = int(img_hw[0]), int(img_hw[1])
    P = H * W
    p = get_params(F)
    nh = p.n_hidden
    nsp = p.steps_pred_hidden[0]
    G2 = p.glimpse_size[0] * p.glimpse_size[1]
    nw = int(F.n_what)
    N = int(F.n_steps_per_image)
    spec = []

    def lin(name, fin, fout, tf, b_init=("zeros",)):
        spec.append((name + ".w", (fin, fout), ("lin_w", fin), tf + "/w"))
        spec.append((name + ".b", (fout,), b_init, tf + "/b"))

    def gru(name, fin, tf):
        for g in "zrh":
            spec.append(("%s.w%s" % (name, g), (fin, nh), ("glorot", fin, nh), "%s/w%s" % (tf, g)))
            spec.append(("%s.u%s" % (name, g), (nh, nh), ("glorot", nh, nh), "%s/u%s" % (tf, g)))
            spec.append(("%s.b%s" % (name, g), (nh,), ("glorot", nh, nh), "%s/b%s" % (tf, g)))

    # ---- decoder (modules.py:368-467, :131-147)
    spec.append(("dec.mean_img", (H, W), ("mean_img",), "decoder/air_decoder/Variable"))
    lin("dec.l0", nw, nh, "decoder/air_decoder/decoder/mlp/linear")
    lin("dec.l1", nh, nh, "decoder/air_decoder/decoder/mlp/linear_1")
    lin("dec.l2", nh, G2, "decoder/air_decoder/decoder/mlp/linear_2")
    spec.append(("dec.output_scale", (), ("const", float(F.output_scale)),
                 "decoder/air_decoder/decoder/output_scale"))

    # ---- discovery (core.py:146-227, sqair_modules.py:66-229, modules.py:548-630)
    # TF variable scopes, verbatim from the reference's own listing (notebooks/play.ipynb:239-362, committed as
    # tests/golden/tf_variables.json): the shared estimators live under the DiscoveryCore module scope
    # `discovery/discovery_core`, the cells constructed in configs/mlp_mnist_model.py:98,118-125 directly under
    # `discovery/` and `propagation/`, trainable initial states under <calling scope>/<cell scope>_initial_state_i.
    d = "discovery/discovery_core"
    rnn_cell = str(getattr(F, "transition", "VanillaRNN"))   # the slot RNN of both cores (mlp_mnist_model.py:86)
    rnn_lstm, rnn_gru = rnn_cell == "LSTM", rnn_cell == "GRU"
    rmod = "lstm" if rnn_lstm else ("gru" if rnn_gru else "vanilla_rnn")
    spec.append(("disc.rnn_init", (1, nh), ("zeros",), "discovery/discover/discovery/" + rmod + "_initial_state_0/w"))
    if rnn_lstm:
        spec.append(("disc.rnn_init_c", (1, nh), ("zeros",), "discovery/discover/discovery/" + rmod + "_initial_state_1/w"))
    lin("disc.steps_prior.l0", 1, 10, "discovery/discover/mlp/linear")
    lin("disc.steps_prior.l1", 10, N + 1, "discovery/discover/mlp/linear_1")
    rn = "discovery/discover/recurrent_normal_impl"
    spec.append(("disc.rn.init_state", (1, 4), ("zeros",), rn + "/" + rn + "/vanilla_rnn_initial_state_0/w"))
    spec.append(("disc.rn.init_sample", (1, 4), ("glorot", 1, 4), rn + "/init_sample"))
    sp = parse_string_flag(F.scale_prior, num_elements=2)
    lin("disc.rn.readout", 4, 8, rn + "/linear",
        b_init=("vec", list(sp) + [0.0, 0.0] + [1.0, 1.0, 1.0, 1.0]))
    lin("disc.rn.cond", 4 + nh + 1, 128, rn + "/linear_1")
    lin("disc.rn.h2h", 128, 4, rn + "/vanilla_rnn/hidden_to_hidden")
    lin("disc.rn.i2h", 4, 4, rn + "/vanilla_rnn/in_to_hidden")
    lin("enc.what_head", nh, 2 * nw, d + "/air_encoder/gaussian_from_param_vec/linear")
    lin("enc.mask.l0", nh, 128, d + "/air_encoder/mlp/linear")
    lin("enc.mask.l1", 128, G2, d + "/air_encoder/mlp/linear_1", b_init=("const", 1.0))
    lin("enc.input.l0", P, nh, d + "/encoder/mlp/linear")
    lin("enc.input.l1", nh, nh, d + "/encoder/mlp/linear_1")
    lin("enc.glimpse.l0", G2, nh, d + "/encoder_1/mlp/linear")
    lin("enc.glimpse.l1", nh, nh, d + "/encoder_1/mlp/linear_1")
    lin("disc.steps.l0", nh + nw, nsp, d + "/steps_predictor/mlp/linear")
    lin("disc.steps.l1", nsp, 1, d + "/steps_predictor/mlp/linear_1",
        b_init=("const", float(F.disc_step_bias)))
    lin("disc.transform.l0", nh, nh, d + "/stochastic_transform_param/mlp/linear")
    lin("disc.transform.l1", nh, nh, d + "/stochastic_transform_param/mlp/linear_1")
    lin("disc.transform.l2", nh, 8, d + "/stochastic_transform_param/mlp/linear_2")
    spec.append(("disc.transform.scale_offset", (), ("const", float(F.transform_var_bias)),
                 d + "/stochastic_transform_param/scale_offset"))
    fin_d = nh + nh + nw + 4 + 1
    if rnn_lstm:
        spec.append(("disc.rnn_lstm.w", (fin_d + nh, 4 * nh), ("lin_w", fin_d + nh), "discovery/lstm/w_gates"))
        spec.append(("disc.rnn_lstm.b", (4 * nh,), ("zeros",), "discovery/lstm/b_gates"))
    elif rnn_gru:
        gru("disc.rnn_gru", fin_d, "discovery/gru")
    else:
        lin("disc.rnn.h2h", nh, nh, "discovery/vanilla_rnn/hidden_to_hidden")
        lin("disc.rnn.i2h", fin_d, nh, "discovery/vanilla_rnn/in_to_hidden")

    # ---- model-scope categorical step prior (sqair_modules.py:209-221)
    spec.append(("disc.step_prior_bias", (N + 1,), ("zeros",),
                 "model/sequential_air/while/sqair_timestep/discover/step_prior_bias"))
    spec.append(("disc.step_prior_timestep_bias", (N + 1,), ("vec", [10.0] + [0.0] * N),
                 "model/sequential_air/while/sqair_timestep/discover/step_prior_timestep_bias"))

    # ---- propagation (core.py:230-359, propagate.py:46-120)
    pc = "propagation/propagation_core"
    # Sonnet uniquifies module names per class in construction order (configs/mlp_mnist_model.py:116-125: transition cell,
    # temporal cell, prior cell): gru / gru_1, lstm / lstm_1 / lstm_2.  (Names of non-shipped cell choices are a best
    # guess from that rule: no listing of such a model exists in the reference.)
    n_mod = {"lstm": 1 if rnn_lstm else 0, "gru": 1 if rnn_gru else 0, "vanilla_rnn": 0 if (rnn_lstm or rnn_gru) else 1}

    def scope_of(kind):
        n_mod[kind] += 1
        return kind if n_mod[kind] == 1 else "{}_{}".format(kind, n_mod[kind] - 1)

    def lstm_scope():
        return scope_of("lstm")
    time_lstm = str(getattr(F, "time_transition", "GRU")) == "LSTM"
    prior_lstm = str(getattr(F, "prior_transition", "GRU")) == "LSTM"
    time_van = str(getattr(F, "time_transition", "GRU")) == "VanillaRNN"
    prior_van = str(getattr(F, "prior_transition", "GRU")) == "VanillaRNN"
    if time_van:
        tmod = scope_of("vanilla_rnn")
        lin("prop.temporal_rnn.h2h", nh, nh, "propagation/" + tmod + "/hidden_to_hidden")
        lin("prop.temporal_rnn.i2h", nh + 4 + 2 * nw, nh, "propagation/" + tmod + "/in_to_hidden")
    elif time_lstm:
        # snt.LSTM(n_hidden): gates = [x, h] w_gates + b_gates, split (i, j, f, o) (SURVEY Appendix B style restatement)
        fin = nh + 4 + 2 * nw
        tmod = lstm_scope()
        spec.append(("prop.temporal_lstm.w", (fin + nh, 4 * nh), ("lin_w", fin + nh), "propagation/" + tmod + "/w_gates"))
        spec.append(("prop.temporal_lstm.b", (4 * nh,), ("zeros",), "propagation/" + tmod + "/b_gates"))
    else:
        tmod = scope_of("gru")
        gru("prop.temporal_gru", nh + 4 + 2 * nw, "propagation/" + tmod)
    if prior_van:
        pmod = scope_of("vanilla_rnn")
        lin("prop.prior_rnn.h2h", nh, nh, "propagation/" + pmod + "/hidden_to_hidden")
        lin("prop.prior_rnn.i2h", nw + 4, nh, "propagation/" + pmod + "/in_to_hidden")
    elif prior_lstm:
        pmod = lstm_scope()
        scope = "propagation/" + pmod
        spec.append(("prop.prior_lstm.w", (nw + 4 + nh, 4 * nh), ("lin_w", nw + 4 + nh), scope + "/w_gates"))
        spec.append(("prop.prior_lstm.b", (4 * nh,), ("zeros",), scope + "/b_gates"))
    else:
        pmod = scope_of("gru")
        gru("prop.prior_gru", nw + 4, "propagation/" + pmod)
    lin("prop.prior_linear", nh, 2 * (4 + nw) + 1, "propagation/propagate_prior/linear")
    spec.append(("prop.cholesky_scale", (10,), ("glorot", 10, 10),
                 pc + "/affine_diag_normal/cholesky_scale"))
    lin("prop.where_bias.l0", nh, 128, pc + "/rnn_inpt/mlp/linear")
    lin("prop.where_bias.l1", 128, 4, pc + "/rnn_inpt/mlp/linear_1")
    lin("prop.steps.l0", 2 * nh + nw, nsp, pc + "/steps_predictor/mlp/linear")
    lin("prop.steps.l1", nsp, 1, pc + "/steps_predictor/mlp/linear_1",
        b_init=("const", float(F.prop_step_bias)))
    lin("prop.transform.l0", 2 * nh + 4, nh, pc + "/stochastic_transform_param/mlp/linear")
    lin("prop.transform.l1", nh, nh, pc + "/stochastic_transform_param/mlp/linear_1")
    lin("prop.transform.l2", nh, 8, pc + "/stochastic_transform_param/mlp/linear_2")
    spec.append(("prop.transform.scale_offset", (), ("const", float(F.transform_var_bias)),
                 pc + "/stochastic_transform_param/scale_offset"))
    lin("prop.what_head", nh, 2 * nw, pc + "/what/gaussian_from_param_vec/linear")
    lin("prop.gates", nh, 3 * nw, pc + "/what/linear", b_init=("const", 1.0))
    spec.append(("prop.rnn_init", (1, nh), ("zeros",), "propagation/sequential_ssm/propagation/" + rmod + "_initial_state_0/w"))
    fin_p = nw + (nw + 4 + 1) + (nw + 4 + 1) + nh
    if rnn_lstm:
        spec.append(("prop.rnn_init_c", (1, nh), ("zeros",), "propagation/sequential_ssm/propagation/" + rmod + "_initial_state_1/w"))
        spec.append(("prop.rnn_lstm.w", (fin_p + nh, 4 * nh), ("lin_w", fin_p + nh), "propagation/lstm/w_gates"))
        spec.append(("prop.rnn_lstm.b", (4 * nh,), ("zeros",), "propagation/lstm/b_gates"))
    elif rnn_gru:
        gru("prop.rnn_gru", fin_p, "propagation/gru")
    else:
        lin("prop.rnn.h2h", nh, nh, "propagation/vanilla_rnn/hidden_to_hidden")
        lin("prop.rnn.i2h", fin_p, nh, "propagation/vanilla_rnn/in_to_hidden")

    # ---- sequence (sqair_modules.py:332-385)
    # trainable initial states, named after the cell's module name (RNNCore.initial_state(trainable=True)); an
    # LSTMState(hidden, cell) has two variables, kept adjacent ([hidden | cell] is read as one row)
    sq = "sequence/sequential_air/propagation/"
    spec.append(("seq.prior_init", (1, nh), ("zeros",), sq + pmod + "_initial_state_0/w"))
    if prior_lstm:
        spec.append(("seq.prior_init_c", (1, nh), ("zeros",), sq + pmod + "_initial_state_1/w"))
    spec.append(("seq.temporal_init", (1, nh), ("zeros",), sq + tmod + "_initial_state_0/w"))
    if time_lstm:
        spec.append(("seq.temporal_init_c", (1, nh), ("zeros",), sq + tmod + "_initial_state_1/w"))
    lin("seq.latent_enc.l0", nw + 4, nh, "sequence/sequential_air/sqair_timestep/mlp/linear")
    lin("seq.latent_enc.l1", nh, nh, "sequence/sequential_air/sqair_timestep/mlp/linear_1")
    return spec


def param_offsets(spec):
    """name -> (offset, shape) into the flat fp32 buffer, plus the total length."""
    off = OrderedDict()
    o = 0
    for name, shape, _, _ in spec:
        n = int(np.prod(shape)) if len(shape) else 1
        off[name] = (o, tuple(shape))
        o += n
    return off, o


def count_params(F, img_hw):
    return param_offsets(param_spec(F, img_hw))[1]


def _trunc_normal(rng, shape, std):
    # TF's truncated_normal: resample outside two standard deviations
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return x * std


def init_params(F, img_hw, seed=0, mean_img=None, jitter=0.0):
    """Random-initialises every parameter (float64 numpy dict, name -> array).

    ``jitter`` > 0 adds N(0, jitter) noise to the zero / constant initialised entries so that
    parity tests exercise biases, initial states and ``mean_img`` with non-trivial values.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape, init, _ in param_spec(F, img_hw):
        kind = init[0]
        if kind == "lin_w":
            v = _trunc_normal(rng, shape, 1.0 / np.sqrt(init[1]))
        elif kind == "zeros":
            v = np.zeros(shape)
        elif kind == "const":
            v = np.full(shape, init[1], dtype=np.float64)
        elif kind == "vec":
            v = np.asarray(init[1], dtype=np.float64).reshape(shape)
        elif kind == "glorot":
            lim = np.sqrt(6.0 / (init[1] + init[2]))
            v = rng.uniform(-lim, lim, size=shape)
        elif kind == "mean_img":
            v = np.zeros(shape) if mean_img is None else np.asarray(mean_img, dtype=np.float64).reshape(shape)
        else:
            raise ValueError(kind)
        if jitter > 0.0 and kind in ("zeros", "const", "vec", "mean_img"):
            v = v + rng.standard_normal(shape) * jitter
        out[name] = np.asarray(v, dtype=np.float64)
    return out


def flatten_params(params, spec):
    off, total = param_offsets(spec)
    flat = np.zeros(total, dtype=np.float32)
    for name, (o, shape) in off.items():
        n = int(np.prod(shape)) if len(shape) else 1
        flat[o:o + n] = np.asarray(params[name], dtype=np.float32).reshape(-1)
    return flat


def unflatten_params(flat, spec):
    off, total = param_offsets(spec)
    assert flat.shape[0] == total
    out = OrderedDict()
    for name, (o, shape) in off.items():
        n = int(np.prod(shape)) if len(shape) else 1
        out[name] = np.array(flat[o:o + n]).reshape(shape)
    return out
