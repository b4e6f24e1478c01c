"""Host-only check of the weight-packing plan (no GPU): the packed GEMMs of libsqair_hip.so are emulated on the
CPU from the plan tables the library exports (which flat-parameter element lands in which packed slot) and the
composite layers — loop-invariant pre-activations + per-slot partial sums, z-record segments with permuted
rows, fused column blocks — are compared with the reference formulas on the named parameters."""
import ctypes as C

import numpy as np
import pytest

from sqair_amd import _capi
from sqair_amd.flags import make_flags
from sqair_amd.model import make_config
from sqair_amd.params import flatten_params, init_params, param_spec

LAYERS = ("IENC0 IENC1 PREDISC PRIOR_GRU1 PRIOR_GRU2 PRIOR_LIN TAU1 WB2 MASK2 GENC0 GENC1 WHAT_LOC WHAT_HEAD PRE PROP_RNN "
          "PROP_T1 PROP_T2 PROP_T3 PROP_GRU1 PROP_GRU2 PROP_HEADS PROP_S1 LAT0 LAT1 PRED RNCOND DISC_RNN DISC_T1 DISC_T2 "
          "DISC_T3 DISC_S1 DEC0 DEC1 DEC2 PROP_RNN2 DISC_RNN2 WHAT_HEAD_I PROP_HEADS_I").split()
NW, NH = 50, 256


class Plan(object):
    def __init__(self, N=4, hw=(20, 24), **flags):
        self.lib = _capi.lib()
        self.F = make_flags(n_steps_per_image=N, **flags)
        cfg = make_config(self.F, hw)
        self.h = C.c_void_p()
        assert self.lib.sqair_create(C.byref(cfg), C.byref(self.h)) == 0
        assert self.lib.sqair_debug_layers(self.h) == len(LAYERS)
        self.spec = param_spec(self.F, hw)
        self.P = {k: np.asarray(v, dtype=np.float32).astype(np.float64)
                  for k, v in init_params(self.F, hw, seed=3, jitter=0.3).items()}
        self.flat = flatten_params(self.P, self.spec).astype(np.float64)

    def layer(self, name):
        lid = LAYERS.index(name)
        kc, nt, n, nseg = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        sw = (C.c_int * 4)()
        assert self.lib.sqair_debug_layer(self.h, lid, C.byref(kc), C.byref(nt), C.byref(n), C.byref(nseg), sw) == 0
        widx = np.zeros(nt.value * kc.value * 256, dtype=np.int32)
        ba = np.zeros(nt.value * 16, dtype=np.int32)
        bb = np.zeros(nt.value * 16, dtype=np.int32)
        assert self.lib.sqair_debug_plan(self.h, lid, widx.ctypes.data_as(C.POINTER(C.c_int)),
                                         ba.ctypes.data_as(C.POINTER(C.c_int)), bb.ctypes.data_as(C.POINTER(C.c_int))) == 0
        packed = np.where(widx >= 0, self.flat[np.maximum(widx, 0)], 0.0).reshape(nt.value, kc.value, 64, 4)
        # undo the MFMA fragment order: lane l, comp i of chunk c -> k = 16c + 4(l>>4) + i, col = 16 tile + (l&15)
        W = np.zeros((kc.value * 16, nt.value * 16))
        for lane in range(64):
            for comp in range(4):
                W[np.ix_(np.arange(kc.value) * 16 + 4 * (lane >> 4) + comp, np.arange(nt.value) * 16 + (lane & 15))] = \
                    packed[:, :, lane, comp].T
        bias = np.where(ba >= 0, self.flat[np.maximum(ba, 0)], 0.0) + np.where(bb >= 0, self.flat[np.maximum(bb, 0)], 0.0)
        return W, bias, [sw[i] for i in range(nseg.value)], n.value

    def apply(self, name, *segs):
        """y = sum over segments (each zero-padded to a multiple of 16) x_seg @ W_rows + bias, true columns only."""
        W, bias, widths, n = self.layer(name)
        assert len(segs) == len(widths), (name, widths)
        y = np.zeros(segs[0].shape[:-1] + (W.shape[1],))
        k0 = 0
        for x, w in zip(segs, widths):
            assert x.shape[-1] == w, (name, x.shape, w)
            y = y + x @ W[k0:k0 + w]
            k0 += -(-w // 16) * 16
        return (y + bias)[..., :n]


def rec(where=None, what=None, pres=None, logit=None, rows=3):
    r = np.zeros((rows, 56))
    if where is not None:
        r[:, 0:4] = where
    if what is not None:
        r[:, 4:54] = what
    if pres is not None:
        r[:, 54] = pres[:, 0]
    if logit is not None:
        r[:, 55] = logit[:, 0]
    return r


@pytest.fixture(scope="module")
def plan():
    p = Plan()
    yield p
    p.lib.sqair_destroy(p.h)


def lin(P, name, x):
    return x @ P[name + ".w"] + P[name + ".b"]


def test_propagation_composites(plan):
    P, rng, R = plan.P, np.random.default_rng(0), 3
    g = lambda n: rng.standard_normal((R, n))
    loc1, what_km1, where_km1, pres_km1 = g(NW), g(NW), g(4), g(1)
    what_tm1, where_tm1, pres_tm1, logit_tm1, tau, r_prev, r_k = g(NW), g(4), g(1), g(1), g(NH), g(NH), g(NH)
    what_k, where_k, enc = g(NW), g(4), g(2 * NW)
    pre = plan.apply("PRE", loc1, rec(where_tm1, what_tm1, pres_tm1, logit_tm1), tau)
    assert pre.shape[1] == NH + NH + NH // 2 + 2 * NH
    # VanillaRNN pre-activation (core.py:296-304)
    rnn = pre[:, :NH] + plan.apply("PROP_RNN", rec(where_km1, what_km1, pres_km1, g(1)), r_prev) - 0.0
    x = np.concatenate([loc1, what_km1, where_km1, pres_km1, what_tm1, where_tm1, pres_tm1, tau], -1)
    ref = lin(P, "prop.rnn.i2h", x) + lin(P, "prop.rnn.h2h", r_prev)
    assert np.allclose(rnn - plan.layer("PROP_RNN")[1][:NH], ref, atol=1e-9)
    # transform hidden layer 1 (core.py:325-326) and the steps-predictor hidden layer (core.py:312-314)
    t1 = plan.apply("PROP_T1", r_k)
    assert np.allclose(pre[:, NH:2 * NH] + t1[:, :NH], lin(P, "prop.transform.l0", np.concatenate([r_k, where_tm1, tau], -1)), atol=1e-9)
    s1 = pre[:, 2 * NH:2 * NH + NH // 2] + t1[:, NH:] + plan.apply("PROP_S1", rec(where_k, what_k, g(1), g(1)))
    assert np.allclose(s1, lin(P, "prop.steps.l0", np.concatenate([r_k, tau, what_k], -1)), atol=1e-9)
    # temporal GRU gates (core.py:340-341)
    xin = np.concatenate([r_k, where_k, enc], -1)
    g1 = plan.apply("PROP_GRU1", r_k, where_k, enc)
    zr = pre[:, 2 * NH + NH // 2:]
    for i, gate in enumerate("zr"):
        ref = xin @ P["prop.temporal_gru.w" + gate] + tau @ P["prop.temporal_gru.u" + gate] + P["prop.temporal_gru.b" + gate]
        assert np.allclose(g1[:, i * NH:(i + 1) * NH] + zr[:, i * NH:(i + 1) * NH], ref, atol=1e-9)
    assert np.allclose(g1[:, 2 * NH:], xin @ P["prop.temporal_gru.wh"] + P["prop.temporal_gru.bh"], atol=1e-9)
    assert np.allclose(plan.apply("PROP_GRU2", tau), tau @ P["prop.temporal_gru.uh"], atol=1e-9)
    heads = plan.apply("PROP_HEADS", tau)
    assert np.allclose(heads, np.concatenate([lin(P, "prop.what_head", tau), lin(P, "prop.gates", tau)], -1), atol=1e-9)
    # prior GRU on [what, where]_{t-1} (propagate.py:78-81)
    pg = plan.apply("PRIOR_GRU1", rec(where_tm1, what_tm1, pres_tm1, logit_tm1), tau)
    xp = np.concatenate([what_tm1, where_tm1], -1)
    for i, gate in enumerate("zr"):
        ref = xp @ P["prop.prior_gru.w" + gate] + tau @ P["prop.prior_gru.u" + gate] + P["prop.prior_gru.b" + gate]
        assert np.allclose(pg[:, i * NH:(i + 1) * NH], ref, atol=1e-9)
    assert np.allclose(pg[:, 2 * NH:], xp @ P["prop.prior_gru.wh"] + P["prop.prior_gru.bh"], atol=1e-9)
    # where-bias + mask MLP first layers share one launch (core.py:292, modules.py:322-324)
    t = plan.apply("TAU1", tau)
    assert np.allclose(t, np.concatenate([lin(P, "prop.where_bias.l0", tau), lin(P, "enc.mask.l0", tau)], -1), atol=1e-9)
    assert np.allclose(plan.apply("WHAT_LOC", tau), lin(P, "enc.what_head", tau)[:, :NW], atol=1e-9)


def test_discovery_and_sequence_composites(plan):
    P, rng, R = plan.P, np.random.default_rng(1), 3
    g = lambda n: rng.standard_normal((R, n))
    ienc, cond, what_p, where_p, pres_p, r_prev, r_j, what_j = g(NH), g(NH), g(NW), g(4), g(1), g(NH), g(NH), g(NW)
    pre = plan.apply("PREDISC", ienc) + plan.apply("PRED", cond) + plan.apply("DISC_RNN", rec(where_p, what_p, pres_p, g(1)), r_prev)
    x = np.concatenate([ienc, cond, what_p, where_p, pres_p], -1)   # core.py:164-177
    assert np.allclose(pre, lin(P, "disc.rnn.i2h", x) + lin(P, "disc.rnn.h2h", r_prev), atol=1e-9)
    t1 = plan.apply("DISC_T1", r_j)
    assert np.allclose(t1[:, :NH], lin(P, "disc.transform.l0", r_j), atol=1e-9)
    s1 = t1[:, NH:] + plan.apply("DISC_S1", rec(g(4), what_j, g(1), g(1)))
    assert np.allclose(s1, lin(P, "disc.steps.l0", np.concatenate([r_j, what_j], -1)), atol=1e-9)
    # latent summary, decoder, conditioning state of the recurrent where prior
    z = rec(where_p, what_p, pres_p, g(1))
    assert np.allclose(plan.apply("LAT0", z), lin(P, "seq.latent_enc.l0", np.concatenate([what_p, where_p], -1)), atol=1e-9)
    assert np.allclose(plan.apply("DEC0", z), lin(P, "dec.l0", what_p), atol=1e-9)
    init = np.repeat(P["disc.rn.init_state"], R, 0)
    ref = np.concatenate([init, cond], -1) @ P["disc.rn.cond.w"][:4 + NH] + P["disc.rn.cond.b"]
    assert np.allclose(plan.apply("RNCOND", init, cond), ref, atol=1e-9)


@pytest.mark.parametrize("name,pname", [("IENC0", "enc.input.l0"), ("GENC0", "enc.glimpse.l0"), ("MASK2", "enc.mask.l1"),
                                        ("PRIOR_LIN", "prop.prior_linear"), ("DEC2", "dec.l2"), ("PROP_T3", "prop.transform.l2")])
def test_plain_layers(plan, name, pname):
    W = plan.P[pname + ".w"]
    x = np.random.default_rng(2).standard_normal((2, W.shape[0]))
    assert np.allclose(plan.apply(name, x), lin(plan.P, pname, x), atol=1e-9)


@pytest.mark.parametrize("rnn", ["LSTM", "GRU"])
def test_cell_variant_composites(rnn):
    """The packing plans of the non-shipped cell choices (transition in {LSTM, GRU}, time_transition = prior_transition =
    LSTM): hoisted + per-slot GEMMs add up to the cell's full pre-activation."""
    plan = Plan(transition=rnn, time_transition="LSTM", prior_transition="LSTM")
    try:
        P, rng, R = plan.P, np.random.default_rng(2), 3
        g = lambda n: rng.standard_normal((R, n))
        loc1, what_km1, where_km1, pres_km1 = g(NW), g(NW), g(4), g(1)
        what_tm1, where_tm1, pres_tm1, logit_tm1, tau, r_prev, r_k, hid = g(NW), g(4), g(1), g(1), g(NH), g(NH), g(NH), g(NH)
        where_k, enc = g(4), g(2 * NW)
        rw = NH * (4 if rnn == "LSTM" else 3)
        pre = plan.apply("PRE", loc1, rec(where_tm1, what_tm1, pres_tm1, logit_tm1), tau)
        assert pre.shape[1] == rw + NH + NH // 2          # the temporal LSTM's recurrent rows are their own layer
        x = np.concatenate([loc1, what_km1, where_km1, pres_km1, what_tm1, where_tm1, pres_tm1, tau], -1)
        got = pre[:, :rw] + plan.apply("PROP_RNN", rec(where_km1, what_km1, pres_km1, g(1)), r_prev)
        ienc, cond = g(NH), g(NH)
        xd = np.concatenate([ienc, cond, what_km1, where_km1, pres_km1], -1)
        gotd = plan.apply("PREDISC", ienc) + plan.apply("PRED", cond) + plan.apply("DISC_RNN", rec(where_km1, what_km1, pres_km1, g(1)), r_prev)
        if rnn == "LSTM":
            assert np.allclose(got, np.concatenate([x, r_prev], -1) @ P["prop.rnn_lstm.w"] + P["prop.rnn_lstm.b"], atol=1e-9)
            assert np.allclose(gotd, np.concatenate([xd, r_prev], -1) @ P["disc.rnn_lstm.w"] + P["disc.rnn_lstm.b"], atol=1e-9)
        else:
            for core, xx, yy in (("prop", x, got), ("disc", xd, gotd)):
                for i, gate in enumerate("zr"):
                    ref = xx @ P[core + ".rnn_gru.w" + gate] + r_prev @ P[core + ".rnn_gru.u" + gate] + P[core + ".rnn_gru.b" + gate]
                    assert np.allclose(yy[:, i * NH:(i + 1) * NH], ref, atol=1e-9)
                assert np.allclose(yy[:, 2 * NH:], xx @ P[core + ".rnn_gru.wh"] + P[core + ".rnn_gru.bh"], atol=1e-9)
                assert np.allclose(plan.apply(core.upper() + "_RNN2", hid), hid @ P[core + ".rnn_gru.uh"], atol=1e-9)
        # temporal LSTM: input rows per slot + recurrent rows (and the bias) for all slots at once
        xin = np.concatenate([r_k, where_k, enc], -1)
        gates = plan.apply("PROP_GRU1", r_k, where_k, enc) + plan.apply("PROP_GRU2", hid)
        assert np.allclose(gates, np.concatenate([xin, hid], -1) @ P["prop.temporal_lstm.w"] + P["prop.temporal_lstm.b"], atol=1e-9)
        # prior LSTM in one layer
        pg = plan.apply("PRIOR_GRU1", rec(where_tm1, what_tm1, pres_tm1, logit_tm1), hid)
        xp = np.concatenate([what_tm1, where_tm1], -1)
        assert np.allclose(pg, np.concatenate([xp, hid], -1) @ P["prop.prior_lstm.w"] + P["prop.prior_lstm.b"], atol=1e-9)
    finally:
        plan.lib.sqair_destroy(plan.h)


def test_interleaved_forward_packs_hold_the_same_columns(plan):
    """The forward-only packs of the what fusion (sqair_glue.h: WhatArgs) are column permutations of the layers they stand in for:
    WHAT_HEAD_I column 2 c / 2 c + 1 = WHAT_HEAD column c / nw + c (loc, scale of element c); PROP_HEADS_I column 16 t + 5 e + g =
    PROP_HEADS column g nw + 3 t + e (the five pre-activations of element 3 t + e), every other column of a tile empty."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, NH))
    ref, got = plan.apply("WHAT_HEAD", x), plan.apply("WHAT_HEAD_I", x)
    assert got.shape[-1] == 2 * NW
    for c in range(NW):
        assert np.array_equal(got[:, 2 * c], ref[:, c]) and np.array_equal(got[:, 2 * c + 1], ref[:, NW + c])
    ref = plan.apply("PROP_HEADS", x)
    W, bias, widths, n = plan.layer("PROP_HEADS_I")
    full = x @ W[:NH] + bias
    nt = (NW + 2) // 3
    assert n == 16 * nt and W.shape[1] == 16 * nt
    for t in range(nt):
        for e in range(3):
            c = 3 * t + e
            for g in range(5):
                col = full[:, 16 * t + 5 * e + g]
                assert np.array_equal(col, ref[:, g * NW + c]) if c < NW else not col.any()
        assert not full[:, 16 * t + 15].any()
