"""Known-answer tests that pin the oracle to identities derivable from the reference's own
formulas; the reference ships no tests or golden vectors."""
import math

import numpy as np
import pytest
import torch

from oracle import sqair_oracle as O
from sqair_amd.flags import make_flags
from sqair_amd.params import count_params, init_params, param_spec

D = torch.float64


def test_param_count_matches_reference_listing():
    # reference: notebooks/play.ipynb:362 (2 951 522 at N=3, 50x50) and per-scope :249,296,301,349,357
    F = make_flags(n_steps_per_image=3)
    assert count_params(F, (50, 50)) == 2951522
    spec = param_spec(F, (50, 50))
    scope = {}
    for name, shape, _, tf in spec:
        scope[tf.split("/")[0]] = scope.get(tf.split("/")[0], 0) + int(np.prod(shape))
    assert scope == {"decoder": 184149, "discovery": 1403398, "model": 8, "propagation": 1283583,
                     "sequence": 80384}
    # N-dependence: only the [N+1] vectors and the [10, N+1] layer
    F4 = make_flags(n_steps_per_image=4)
    assert count_params(F4, (50, 50)) - 2951522 == 2 + 10 + 1


def test_st_crop_identity():
    # (sx,sy,tx,ty) = (1,1,0,0) with G=H=W returns the image (modules.py:170-218)
    img = torch.rand(2, 9, 9, dtype=D)
    big = 40.0  # sigmoid(40) == 1.0 in fp64
    where = torch.tensor([[big, big, 0.0, 0.0]] * 2, dtype=D)
    assert torch.allclose(O.st_crop(img, where, 9), img, atol=1e-12)


def test_st_crop_matches_grid_sample():
    # resampler == grid_sample(bilinear, zeros, align_corners=True) on (sx xn + tx, sy yn + ty)
    torch.manual_seed(0)
    img = torch.rand(5, 13, 17, dtype=D)
    where = torch.randn(5, 4, dtype=D) * 1.5
    G = 6
    sx, sy, tx, ty = O.to_coords(where)
    g = torch.linspace(-1, 1, G, dtype=D)
    gx = (sx[:, None] * g[None] + tx[:, None])[:, None, :].expand(-1, G, -1)
    gy = (sy[:, None] * g[None] + ty[:, None])[:, :, None].expand(-1, -1, G)
    ref = torch.nn.functional.grid_sample(img[:, None], torch.stack([gx, gy], -1), mode="bilinear",
                                          padding_mode="zeros", align_corners=True)[:, 0]
    assert torch.allclose(O.st_crop(img, where, G), ref, atol=1e-12)


def test_insert_after_crop_is_identity_on_support():
    # a glimpse that covers the whole image at the same resolution: insert(crop(x)) == x
    img = torch.rand(1, 7, 7, dtype=D)
    where = torch.tensor([[40.0, 40.0, 0.0, 0.0]], dtype=D)
    g = O.st_crop(img, where, 7)
    assert torch.allclose(O.st_insert(g, where, 7, 7), img, atol=1e-10)


def test_insert_zero_outside_glimpse():
    ones = torch.ones(1, 20, 20, dtype=D)
    where = torch.tensor(O.to_logits([[0.3, 0.3, 0.2, -0.1]]), dtype=D)
    m = O.st_insert(ones, where, 50, 50)[0]
    y, x, h, w = O.stn_to_pixel_coords(np.array([0.3, 0.3, 0.2, -0.1]), (50, 50))
    assert m.max() <= 1 + 1e-12 and m.min() >= 0
    inside = m[int(y + 3):int(y + 10), int(x + 3):int(x + 10)]
    assert torch.allclose(inside, torch.ones_like(inside), atol=1e-9)
    assert float(m[0, 0]) == 0.0 and float(m[-1, -1]) == 0.0


def test_stn_pixel_round_trip_and_logits():
    # modules.py:246-280 and :221-243
    stn = np.array([[0.4, 0.25, -0.3, 0.55]])
    px = O.stn_to_pixel_coords(stn, (50, 40))
    assert np.allclose(O.pixel_to_stn_coords(px, (50, 40)), stn)
    lg = O.to_logits(stn)
    sx, sy, tx, ty = O.to_coords(torch.tensor(lg, dtype=D))
    assert np.allclose(torch.stack([sx, sy, tx, ty], -1).numpy(), stn, atol=1e-12)


def test_modified_geometric():
    # prior.py:61-67
    j = O.bernoulli_to_modified_geometric(torch.tensor([[0.5, 0.5, 0.5]], dtype=D))
    assert torch.allclose(j, torch.tensor([[0.5, 0.25, 0.125, 0.125]], dtype=D))
    lp = O.num_steps_log_prob(j, torch.tensor([2.0], dtype=D))
    assert abs(float(lp) - math.log(0.125)) < 1e-12
    # zero probability is clipped at 1e-16 (prior.py:98)
    j0 = O.bernoulli_to_modified_geometric(torch.tensor([[0.0, 0.3]], dtype=D))
    assert abs(float(O.num_steps_log_prob(j0, torch.tensor([2.0], dtype=D))) - math.log(1e-16)) < 1e-9


def test_iwae_and_vimco_control_variate():
    # targets.py:38-59
    w = torch.full((3, 5), -7.25, dtype=D)
    assert torch.allclose(O.iwae(w), torch.full((3,), -7.25, dtype=D))
    w2 = torch.tensor([[1.0, -2.0]], dtype=D)
    cv = O.vimco_control_variate(w2)  # K=2: replacing w_j by the other one -> logmeanexp = other
    assert torch.allclose(cv, torch.tensor([[-2.0, 1.0]], dtype=D))


def test_vimco_gradient_structure():
    # targets.py:62-75: d loss / d log_probs = -signal / (B K), signal = stopgrad(log_w - cv)
    torch.manual_seed(1)
    lw = torch.randn(4, 3, dtype=D, requires_grad=True)
    lp = torch.randn(4, 3, dtype=D, requires_grad=True)
    loss = O.vimco(lw, lp)
    loss.backward()
    sig = (lw - O.vimco_control_variate(lw)).detach()
    assert torch.allclose(lp.grad, -sig / 12.0)
    assert torch.allclose(lw.grad, -torch.softmax(lw.detach(), -1) / 4.0)


def test_select_present_and_ids():
    # index.py:132-165, :198-221
    x = torch.tensor([[[1.0], [2.0], [3.0], [4.0]]], dtype=D)
    pres = torch.tensor([[0.0, 1.0, 0.0, 1.0]], dtype=D)
    assert O.select_present(x, pres).flatten().tolist() == [2.0, 4.0, 1.0, 3.0]
    last, ids = O.compute_object_ids(torch.tensor([[2.0]], dtype=D), torch.tensor([[[5.0], [7.0]]], dtype=D),
                                     torch.tensor([[[1.0], [0.0]]], dtype=D), torch.tensor([[[1.0], [1.0]]], dtype=D))
    assert last.flatten().tolist() == [4.0]
    assert ids.flatten().tolist() == [5.0, -1.0, 3.0, 4.0]


def test_tile_input_ordering():
    # index.py:106-129: b' = b*K + k
    x = torch.arange(6, dtype=D).reshape(1, 3, 2)
    t = O.tile_input_for_iwae(x, 2)
    assert t[0, :, 0].tolist() == [0, 0, 2, 2, 4, 4]


def test_fill_triangular_convention():
    t = O.fill_triangular(torch.arange(1.0, 7.0, dtype=D), 3)
    assert t.tolist() == [[4, 0, 0], [6, 5, 0], [3, 2, 1]]


def test_bernoulli_and_absent_slots_stay_absent():
    # modules.py:513, propagate.py:86: absent slot -> logit -88 -> presence 0, log-prob ~ 0
    lp = O.bernoulli_log_prob(torch.tensor(0.0, dtype=D), torch.tensor(-88.0, dtype=D))
    assert abs(float(lp)) < 1e-30
    F = make_flags(n_steps_per_image=2, k_particles=1)
    cfg = O.make_cfg(F, (12, 12))
    P = init_params(F, (12, 12), seed=3, jitter=0.1)
    orc = O.SqairOracle(P, cfg)
    rng = np.random.default_rng(0)
    obs = rng.uniform(size=(2, 3, 12, 12))
    nz = rng.standard_normal((2, 3, 2, 2, O.noise_width(cfg)))
    nz[..., -1] = 0.0  # u = 0: every live Bernoulli fires, dead ones (logit -88) must not
    m = orc.model(obs, nz)
    assert float(m.prop_pres[0].abs().sum()) == 0.0  # nothing to propagate at t=0
    assert torch.all(m.disc_pres[0] == 1.0)
    assert torch.all(m.obj_id[0] == torch.tensor([0.0, 1.0], dtype=D))
    # at t=1 both slots are propagated (u=0), discoveries are all present but truncated away
    assert torch.all(m.prop_pres[1] == 1.0) and torch.all(m.presence[1] == 1.0)
    assert torch.all(m.obj_id[1] == torch.tensor([0.0, 1.0], dtype=D))
    assert torch.all(m.num_steps_per_sample == 2.0)


def test_every_parameter_receives_gradient():
    # reference asserts the same at model.py:163-166
    F = make_flags(n_steps_per_image=2, k_particles=2)
    cfg = O.make_cfg(F, (16, 16))
    orc = O.SqairOracle(init_params(F, (16, 16), seed=0, jitter=0.05), cfg, requires_grad=True)
    rng = np.random.default_rng(1)
    obs = rng.uniform(size=(3, 2, 16, 16))
    nz = rng.standard_normal((3, 4, 2, 2, O.noise_width(cfg)))
    nz[..., -1] = rng.uniform(size=nz.shape[:-1])
    m = orc.model(obs, nz)
    orc.make_target(m).backward()
    assert [k for k, v in orc.P.items() if v.grad is None] == []
    assert all(torch.isfinite(v.grad).all() for v in orc.P.values())


def test_fp32_mode_close_to_fp64():
    F = make_flags(n_steps_per_image=2, k_particles=2)
    cfg = O.make_cfg(F, (16, 16))
    P = init_params(F, (16, 16), seed=0, jitter=0.05)
    rng = np.random.default_rng(1)
    obs = rng.uniform(size=(3, 2, 16, 16))
    nz = rng.standard_normal((3, 4, 2, 2, O.noise_width(cfg)))
    nz[..., -1] = rng.uniform(size=nz.shape[:-1])
    a = O.SqairOracle(P, cfg, torch.float64).model(obs, nz)
    b = O.SqairOracle(P, cfg, torch.float32).model(obs, nz)
    if torch.equal(a.presence, b.presence.double()):
        assert abs(float(a.elbo_iwae) - float(b.elbo_iwae)) <= 1e-4 * abs(float(a.elbo_iwae))


# ---------------------------------------------------------------------------------------------------------------------
# Cross-checks of the oracle's primitives against INDEPENDENT implementations of the same published definitions
# (torch.distributions / torch.nn / scipy were not written from this repo's restatement): as close as this container
# gets to pinning the third-party arithmetic (TF 1.6 contrib.distributions, Sonnet cells) without TF.
# ---------------------------------------------------------------------------------------------------------------------
def test_log_probs_match_torch_distributions():
    import torch.distributions as D
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 5, dtype=torch.float64, generator=g)
    loc = torch.randn(7, 5, dtype=torch.float64, generator=g)
    scale = torch.rand(7, 5, dtype=torch.float64, generator=g) + 0.1
    assert torch.allclose(O.normal_log_prob(x, loc, scale), D.Normal(loc, scale).log_prob(x), atol=1e-12)
    logits = torch.randn(9, dtype=torch.float64, generator=g) * 5
    for v in (0.0, 1.0):
        b = torch.full_like(logits, v)
        assert torch.allclose(O.bernoulli_log_prob(b, logits), D.Bernoulli(logits=logits).log_prob(b), atol=1e-12)
    # extreme logits (the -88 of absent slots) stay finite and exact
    big = torch.tensor([-88.0, 88.0], dtype=torch.float64)
    assert torch.allclose(O.bernoulli_log_prob(torch.tensor([0.0, 1.0], dtype=torch.float64), big),
                          torch.zeros(2, dtype=torch.float64), atol=1e-30)
    # MultivariateNormalTriL
    L = torch.tril(torch.randn(6, 4, 4, dtype=torch.float64, generator=g) * 0.3, diagonal=-1) + torch.diag_embed(
        torch.rand(6, 4, dtype=torch.float64, generator=g) + 0.5)
    xm, lm = torch.randn(6, 4, dtype=torch.float64, generator=g), torch.randn(6, 4, dtype=torch.float64, generator=g)
    want = D.MultivariateNormal(lm, scale_tril=L).log_prob(xm)
    assert torch.allclose(O.SqairOracle.mvn_tril_log_prob(xm, lm, L), want, atol=1e-10)


def test_fill_triangular_matches_the_documented_examples():
    """tfd.fill_triangular docstring: [1..6] -> [[4,0,0],[6,5,0],[3,2,1]]; [1..10] (n = 4) follows the same clockwise
    spiral: rows [7,0,0,0], [8? ...] are fixed by reshape(concat(v[n:], reverse(v)), [n, n])."""
    m3 = O.fill_triangular(torch.arange(1.0, 7.0), 3)
    assert m3.tolist() == [[4.0, 0.0, 0.0], [6.0, 5.0, 0.0], [3.0, 2.0, 1.0]]
    m4 = O.fill_triangular(torch.arange(1.0, 11.0), 4)
    assert m4.tolist() == [[5.0, 0.0, 0.0, 0.0], [9.0, 10.0, 0.0, 0.0], [8.0, 7.0, 6.0, 0.0], [4.0, 3.0, 2.0, 1.0]]


def test_vanilla_rnn_matches_torch_rnncell_and_gru_matches_its_equations():
    g = torch.Generator().manual_seed(1)
    nin, nh, B = 7, 5, 3
    P = {"c.i2h.w": torch.randn(nin, nh, dtype=torch.float64, generator=g), "c.i2h.b": torch.randn(nh, dtype=torch.float64, generator=g),
         "c.h2h.w": torch.randn(nh, nh, dtype=torch.float64, generator=g), "c.h2h.b": torch.randn(nh, dtype=torch.float64, generator=g)}
    x, h = torch.randn(B, nin, dtype=torch.float64, generator=g), torch.randn(B, nh, dtype=torch.float64, generator=g)
    cell = torch.nn.RNNCell(nin, nh, nonlinearity="tanh").double()
    with torch.no_grad():
        cell.weight_ih.copy_(P["c.i2h.w"].T); cell.weight_hh.copy_(P["c.h2h.w"].T)
        cell.bias_ih.copy_(P["c.i2h.b"]); cell.bias_hh.copy_(P["c.h2h.b"])
        assert torch.allclose(O.vanilla_rnn(P, "c", x, h), cell(x, h), atol=1e-12)
    # snt.GRU is NOT torch.nn.GRUCell (Sonnet applies the reset gate before the recurrent matmul): check the limits
    # that pin the equations instead: z -> 0 keeps the state, z -> 1 with r -> 0 gives tanh(x W_h + b_h)
    G = {"g.w" + k: torch.randn(nin, nh, dtype=torch.float64, generator=g) for k in "zrh"}
    G.update({"g.u" + k: torch.randn(nh, nh, dtype=torch.float64, generator=g) for k in "zrh"})
    G.update({"g.b" + k: torch.zeros(nh, dtype=torch.float64) for k in "zrh"})
    keep = dict(G); keep["g.bz"] = torch.full((nh,), -60.0, dtype=torch.float64)
    assert torch.allclose(O.gru(keep, "g", x, h), h, atol=1e-12)
    new = dict(G); new["g.bz"] = torch.full((nh,), 60.0, dtype=torch.float64); new["g.br"] = torch.full((nh,), -60.0, dtype=torch.float64)
    assert torch.allclose(O.gru(new, "g", x, h), torch.tanh(x @ G["g.wh"]), atol=1e-10)
    # and with r -> 1 the candidate sees the full recurrent term
    full = dict(new); full["g.br"] = torch.full((nh,), 60.0, dtype=torch.float64)
    assert torch.allclose(O.gru(full, "g", x, h), torch.tanh(x @ G["g.wh"] + h @ G["g.uh"]), atol=1e-10)


def test_lstm_matches_torch_lstmcell_with_the_forget_bias_folded_in():
    """snt.LSTM: gate order (i, j, f, o), forget bias +1 added inside the cell; torch.nn.LSTMCell: (i, f, g, o), no forget
    bias.  Same function once the columns are permuted and +1 is folded into the forget-gate bias."""
    g = torch.Generator().manual_seed(2)
    nin, nh, B = 6, 5, 4
    W = torch.randn(nin + nh, 4 * nh, dtype=torch.float64, generator=g)
    b = torch.randn(4 * nh, dtype=torch.float64, generator=g)
    x, h, c = (torch.randn(B, n, dtype=torch.float64, generator=g) for n in (nin, nh, nh))
    i_, j_, f_, o_ = (slice(k * nh, (k + 1) * nh) for k in range(4))
    perm = torch.cat([torch.arange(4 * nh)[sl] for sl in (i_, f_, j_, o_)])   # snt (i, j, f, o) -> torch (i, f, g, o)
    cell = torch.nn.LSTMCell(nin, nh).double()
    with torch.no_grad():
        cell.weight_ih.copy_(W[:nin, perm].T); cell.weight_hh.copy_(W[nin:, perm].T)
        bb = b.clone(); bb[f_] += 1.0
        cell.bias_ih.copy_(bb[perm]); cell.bias_hh.zero_()
        h_t, c_t = cell(x, (h, c))
        h_o, c_o = O.lstm({"l.w": W, "l.b": b}, "l", x, h, c)
    assert torch.allclose(h_o, h_t, atol=1e-12) and torch.allclose(c_o, c_t, atol=1e-12)


def test_step_distributions_match_independent_formulas():
    import torch.distributions as D
    from scipy.special import logsumexp
    # Categorical(logits).log_prob = log_softmax
    lg = torch.tensor([[0.3, -1.2, 2.0, 0.0]], dtype=torch.float64)
    want = lg.numpy()[0] - logsumexp(lg.numpy()[0])
    assert np.allclose(D.Categorical(logits=lg).log_prob(torch.tensor([2])).item(), want[2])
    assert np.allclose(torch.log_softmax(lg, -1).numpy()[0], want)
    # tfd.Geometric(probs=p) counts failures before the first success: log p(k) = k log(1 - p) + log p
    p = 0.25
    for k in range(4):
        assert np.isclose(D.Geometric(probs=torch.tensor(p)).log_prob(torch.tensor(float(k))).item(), k * np.log(1 - p) + np.log(p))
    # NumStepsDistribution: the modified geometric built from Bernoulli step probabilities sums to one and its last bin
    # is the probability of all steps succeeding
    pr = torch.tensor([[0.9, 0.5, 0.2]], dtype=torch.float64)
    joint = O.bernoulli_to_modified_geometric(pr)
    assert np.isclose(joint.sum().item(), 1.0) and np.isclose(joint[0, -1].item(), 0.9 * 0.5 * 0.2)
    assert np.isclose(joint[0, 1].item(), 0.9 * (1 - 0.5))
    assert np.isclose(O.num_steps_log_prob(joint, torch.tensor([1.0])).item(), np.log(0.45))


def _gen_case(**flags):
    F = make_flags(n_steps_per_image=3, k_particles=2, **flags)
    hw = (20, 20)
    cfg = O.make_cfg(F, hw)
    P = init_params(F, hw, seed=5, jitter=0.2)
    rng = np.random.default_rng(3)
    T, B = 4, 3
    obs = rng.uniform(size=(T, B, *hw))
    nz = rng.standard_normal((T, B * 2, 2, 3, O.noise_width(cfg)))
    nz[..., -1] = rng.uniform(size=nz.shape[:-1])
    gn = rng.standard_normal(nz.shape)
    gn[..., -1] = rng.uniform(size=nz.shape[:-1])
    return O.SqairOracle(P, cfg), obs, nz, gn


def test_sample_from_prior_only_moves_the_posterior_evaluation_points():
    """mlp_mnist_model.py:51 + sqair_modules.py:294-302 with generate_after unset (the only combination the flags can
    reach): latents, canvases, priors are unchanged; the propagation posterior log-probs are evaluated at prior samples."""
    base, obs, nz, gn = _gen_case()
    sfp, _, _, _ = _gen_case(sample_from_prior=True)
    a, b = base.model(obs, nz), sfp.model(obs, nz, gen_noise=gn)
    for k in ("what", "where", "presence", "canvas", "obj_id", "prop_what_prior_log_prob", "prop_prior_log_prob",
              "disc_what_log_prob", "disc_log_prob", "log_p_z_per_sample", "data_ll_per_sample"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert not torch.equal(a.prop_what_log_prob[1:], b.prop_what_log_prob[1:])
    assert not torch.equal(a.prop_where_log_prob[1:], b.prop_where_log_prob[1:])
    assert not torch.equal(a.log_q_z_given_x_per_sample, b.log_q_z_given_x_per_sample)


def test_generation_after_t_samples_the_priors():
    """seq.py:198-200: frames t > generate_after take what / where / presence of propagated objects from the prior and
    discover nothing (sqair_modules.py:157-170: presence `* 0.`); earlier frames are untouched."""
    base, obs, nz, gn = _gen_case(sample_from_prior=True)
    gen, _, _, _ = _gen_case(sample_from_prior=True, generate_after=1)
    a, b = base.model(obs, nz, gen_noise=gn), gen.model(obs, nz, gen_noise=gn)
    for k in ("what", "where", "presence", "canvas", "log_weights_per_timestep"):
        assert torch.equal(getattr(a, k)[:2], getattr(b, k)[:2]), k          # t = 0, 1 (t > 1 generates)
    assert float(b.disc_pres[2:].abs().sum()) == 0.0
    assert float(b.num_disc_steps_per_sample[:2].sum()) > 0                  # discovery still counts its own steps
    # a propagated, generated object is a prior sample: its presence is the prior Bernoulli's draw
    assert set(np.unique(b.prop_pres[2:].numpy())) <= {0.0, 1.0}
    assert not torch.equal(a.what[2:], b.what[2:]) or float(a.presence[2:].sum()) == 0.0
    assert torch.isfinite(b.log_weights).all()


@pytest.mark.parametrize("cells", [("LSTM", "GRU"), ("GRU", "LSTM"), ("LSTM", "LSTM"), ("GRU", "GRU", "LSTM"), ("LSTM", "LSTM", "LSTM"),
                                   ("GRU", "GRU", "GRU"), ("VanillaRNN", "VanillaRNN", "VanillaRNN")])
def test_lstm_cells_are_wired_into_the_model(cells):
    """time_transition / prior_transition = LSTM (configs/mlp_mnist_model.py:86-87,125): the recurrent states double to
    [hidden | cell], every parameter of the LSTM variant receives a gradient, and the parameter table swaps the nine GRU
    matrices for {w_gates, b_gates} + a second trainable initial state."""
    from sqair_amd.data import make_sequences, to_float
    from sqair_amd.flags import make_flags
    from sqair_amd.params import init_params, param_spec
    rnn = cells[2] if len(cells) > 2 else "VanillaRNN"
    F = make_flags(k_particles=2, n_steps_per_image=3, time_transition=cells[0], prior_transition=cells[1], transition=rnn)
    hw, T, B = (32, 40), 3, 2
    d = make_sequences(B, T=T, canvas=hw, n_objects=(1, 2), obj_size=20, seed=9)
    obs = to_float(d["imgs"])
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
    rng = np.random.default_rng(0)
    noise = rng.standard_normal((T, B * 2, 2, 3, 55)).astype(np.float32)
    noise[..., -1] = rng.uniform(size=noise.shape[:-1])
    orc = O.SqairOracle(P, O.make_cfg(F, hw), torch.float64, requires_grad=True)
    m = orc.model(obs, noise)
    assert float(m.prop_pres.detach().sum()) > 0
    assert m.outputs["_final_temporal_state"].shape[-1] == (512 if cells[0] == "LSTM" else 256)
    assert m.outputs["_final_prior_state"].shape[-1] == (512 if cells[1] == "LSTM" else 256)
    orc.make_target(m).backward()
    assert all(v.grad is not None and float(v.grad.abs().max()) > 0 for v in orc.P.values())
    names = [s[0] for s in param_spec(F, hw)]
    tf_names = [s[3] for s in param_spec(F, hw)]
    assert len(set(tf_names)) == len(tf_names)
    assert ("prop.temporal_lstm.w" in names) == (cells[0] == "LSTM") and ("prop.temporal_gru.wz" in names) == (cells[0] == "GRU")
    assert ("prop.temporal_rnn.i2h.w" in names) == (cells[0] == "VanillaRNN") and ("prop.prior_rnn.h2h.b" in names) == (cells[1] == "VanillaRNN")
    assert ("prop.prior_lstm.w" in names) == (cells[1] == "LSTM") and ("seq.prior_init_c" in names) == (cells[1] == "LSTM")
    assert ("prop.rnn_lstm.w" in names) == (rnn == "LSTM") == ("disc.rnn_init_c" in names)
    assert ("prop.rnn.i2h.w" in names) == (rnn == "VanillaRNN") and ("disc.rnn_gru.uh" in names) == (rnn == "GRU")
