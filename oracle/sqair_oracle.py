"""CPU oracle for the SQAIR Discover/Propagate hot path.   *** TEST INFRASTRUCTURE ***

This file is a checker, not a product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it; ``sqair_amd`` never does and fails
loudly when its HIP library is missing.

What it is: a PyTorch-CPU restatement (float64 master, float32 switchable) of the forward
pass the reference builds as a TF1 graph, *at the reference's op granularity* (the input
encoder is re-evaluated at every discovery step, the mask MLP twice per propagation slot,
glimpses and canvases are materialised) so that its cost structure mirrors the reference
CPU path.  Every stochastic site consumes caller-supplied noise (eps for Normals, u for
Bernoullis) so that a GPU run can be compared sample for sample.  Autograd gives the
gradients of the VIMCO target for free (stop-gradients restated where the reference has
them).

PARITY UNPINNED: the reference is Python-2 / TensorFlow-1.6 / Sonnet-1.14 code with no tests,
no golden vectors and no fixtures, and none of those packages can be imported here
(SURVEY.md section 8(c)).  The semantics of the third-party ops the arithmetic lives in
(dm_sonnet==1.14: AffineGridWarper(+inverse), VanillaRNN, GRU, Linear; TensorFlow 1.6:
contrib.resampler, contrib.distributions Normal/Bernoulli/Categorical/MultivariateNormalTriL/
fill_triangular, dynamic_partition) are restated from their published definitions
(SURVEY.md Appendix B).  What *is* pinned: the variable inventory (2 951 522 parameters,
reference notebooks/play.ipynb:239-362) and the known-answer identities of
tests/test_oracle_known_answers.py, all derived from the reference's own formulas.

Function docstrings cite the reference file:line they follow (paths under
/root/reference/sqair/).

Noise layout: ``noise[t, b', s, k, :]`` with s = 0 (propagation slot k) or 1 (discovery step
k); entries 0:4 = eps for ``where``, 4:4+n_what = eps for ``what``, last = uniform u of the
presence Bernoulli.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as Fn

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------- config
def make_cfg(F, img_hw):
    """Collects the scalars the forward pass needs from a flags object
    (sqair_amd.flags.Flags or anything with the same attribute names)."""
    sp = [float(s) for s in str(F.scale_prior).split(",")]
    if len(sp) == 1:
        sp = sp * 2
    # modules.py:419-422: std is stored as sqrt(value) in fp32 and squared again in fp32
    std = float(np.float32(np.float32(np.sqrt(F.output_std)) ** np.float32(2.0)))
    return SimpleNamespace(
        H=int(img_hw[0]), W=int(img_hw[1]), G=int(F.glimpse_size), N=int(F.n_steps_per_image),
        n_what=int(F.n_what), n_hidden=32 * int(F.n_units), K=int(F.k_particles),
        prop_prior_step_bias=float(F.prop_prior_step_bias), prop_prior_type=str(F.prop_prior_type),
        masked_glimpse=bool(F.masked_glimpse), disc_prior_type=str(F.disc_prior_type),
        step_success_prob=float(F.step_success_prob), rec_where_prior=bool(F.rec_where_prior),
        where_prior_mean=sp + [0.0, 0.0], output_std=std, background_std=std,
        transition=str(F.transition), time_transition=str(F.time_transition),
        prior_transition=str(F.prior_transition), where_update_scale=1.0, min_std=1e-2,
        # generation modes (mlp_mnist_model.py:51, seq.py:46, :198-200); generate_after is a constructor argument of
        # SequentialAIR only (no flag), carried here as an optional attribute of the flags object
        sample_from_prior=bool(getattr(F, "sample_from_prior", False)), generate_after=int(getattr(F, "generate_after", -1)),
    )


def noise_width(cfg):
    return 4 + cfg.n_what + 1


# ----------------------------------------------------------------------------- primitives
def elu(x):
    return Fn.elu(x)


def softplus(x):
    return Fn.softplus(x, beta=1.0, threshold=1e9)


def linear(P, name, x):
    """snt.Linear: x W + b (neural.py:34-47)."""
    return x @ P[name + ".w"] + P[name + ".b"]


def mlp2_hidden(P, name, x):
    """MLP([n, n]): two ELU layers, no output layer (neural.py:34-116; used by Encoder
    modules.py:100-112 and the latent encoder sqair_modules.py:347-349)."""
    return elu(linear(P, name + ".l1", elu(linear(P, name + ".l0", x))))


def mlp_1hidden_out(P, name, x, transfer=None):
    """MLP(n_hidden, n_out=...): one ELU layer + linear/``transfer`` output (neural.py:90-98)."""
    y = linear(P, name + ".l1", elu(linear(P, name + ".l0", x)))
    return transfer(y) if transfer is not None else y


def vanilla_rnn(P, name, x, h):
    """snt.VanillaRNN: tanh(in_to_hidden(x) + hidden_to_hidden(h))."""
    return torch.tanh(linear(P, name + ".i2h", x) + linear(P, name + ".h2h", h))


def gru(P, name, x, h):
    """snt.GRU: z, r gates; candidate uses (r*h) U_h; h' = (1-z) h + z h~."""
    z = torch.sigmoid(x @ P[name + ".wz"] + h @ P[name + ".uz"] + P[name + ".bz"])
    r = torch.sigmoid(x @ P[name + ".wr"] + h @ P[name + ".ur"] + P[name + ".br"])
    hc = torch.tanh(x @ P[name + ".wh"] + (r * h) @ P[name + ".uh"] + P[name + ".bh"])
    return (1.0 - z) * h + z * hc


def lstm(P, name, x, h, c, forget_bias=1.0):
    """snt.LSTM (dm_sonnet 1.14, restated): gates = [x, h] w_gates + b_gates, split into (i, j, f, o);
    c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j); h' = tanh(c') sigmoid(o).  Returns (h', c')."""
    g = torch.cat([x, h], -1) @ P[name + ".w"] + P[name + ".b"]
    i, j, f, o = torch.chunk(g, 4, -1)
    c2 = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
    return torch.tanh(c2) * torch.sigmoid(o), c2


def normal_log_prob(x, loc, scale):
    """tfd.Normal.log_prob."""
    return -0.5 * ((x - loc) / scale) ** 2 - torch.log(scale) - 0.5 * LOG_2PI


def bernoulli_log_prob(x, logits):
    """tfd.Bernoulli(logits).log_prob = -sigmoid_cross_entropy_with_logits(labels=x)."""
    return -(torch.clamp(logits, min=0.0) - logits * x + torch.log1p(torch.exp(-torch.abs(logits))))


def clip_preserve_min(x, lo):
    """ops.clip_preserve(x, lo, x) (ops.py:33-42): forward max(x, lo), gradient of identity."""
    return x + (torch.clamp(x, min=lo) - x).detach()


def fill_triangular(v, n):
    """tfd.fill_triangular (lower): reshape(concat(v[n:], reverse(v)), [n, n]) then lower band
    (e.g. [1..6] -> [[4,0,0],[6,5,0],[3,2,1]])."""
    m = torch.cat([v[n:], torch.flip(v, dims=[0])]).reshape(n, n)
    return torch.tril(m)


# ----------------------------------------------------------------------------- spatial transformer
def to_coords(where_logits):
    """SpatialTransformer.to_coords (modules.py:220-227) + the clip of modules.py:205-206:
    (sx, sy, tx, ty) = (sigmoid, sigmoid, tanh, tanh), sx, sy >= 1e-4."""
    sx = clip_preserve_min(torch.sigmoid(where_logits[..., 0]), 1e-4)
    sy = clip_preserve_min(torch.sigmoid(where_logits[..., 1]), 1e-4)
    tx = torch.tanh(where_logits[..., 2])
    ty = torch.tanh(where_logits[..., 3])
    return sx, sy, tx, ty


def bilinear_gather(src, x, y):
    """tf.contrib.resampler semantics on an axis-aligned grid (modules.py:170-173): bilinear,
    every out-of-range tap contributes zero.  src [R,Hs,Ws]; x [R,Wo], y [R,Ho] in source
    pixels; returns [R,Ho,Wo]."""
    R, Hs, Ws = src.shape
    Ho, Wo = y.shape[1], x.shape[1]
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    wx1 = x - x0
    wy1 = y - y0
    flat = src.reshape(R, Hs * Ws)
    out = None
    for yi, wy in ((y0, 1.0 - wy1), (y0 + 1.0, wy1)):
        vy = ((yi >= 0) & (yi <= Hs - 1)).to(src.dtype)
        yc = yi.clamp(0, Hs - 1).long()
        for xi, wx in ((x0, 1.0 - wx1), (x0 + 1.0, wx1)):
            vx = ((xi >= 0) & (xi <= Ws - 1)).to(src.dtype)
            xc = xi.clamp(0, Ws - 1).long()
            idx = (yc[:, :, None] * Ws + xc[:, None, :]).reshape(R, Ho * Wo)
            vals = flat.gather(1, idx).reshape(R, Ho, Wo)
            term = (wy * vy)[:, :, None] * (wx * vx)[:, None, :] * vals
            out = term if out is None else out + term
    return out


def st_crop(img, where_logits, G):
    """Forward spatial transformer (modules.py:150-218, Sonnet AffineGridWarper no_shear_2d):
    glimpse[i,j] = bilinear(img, x=(W-1)/2 (sx xn_j + tx + 1), y=(H-1)/2 (sy yn_i + ty + 1)),
    xn, yn = linspace(-1, 1, G).  img [R,H,W]; where_logits [R,4] -> [R,G,G]."""
    R, H, W = img.shape
    sx, sy, tx, ty = to_coords(where_logits)
    g = torch.linspace(-1.0, 1.0, G, dtype=img.dtype)
    x = 0.5 * (W - 1) * (sx[:, None] * g[None, :] + tx[:, None] + 1.0)
    y = 0.5 * (H - 1) * (sy[:, None] * g[None, :] + ty[:, None] + 1.0)
    return bilinear_gather(img, x, y)


def st_insert(glimpse, where_logits, H, W):
    """Inverse spatial transformer (modules.py:165-168, AffineGridWarper.inverse()): every
    canvas pixel samples the glimpse at xg = (G-1)/2 ((Xn - tx)/sx + 1), zero outside.
    glimpse [R,G,G] -> [R,H,W]."""
    R, G, _ = glimpse.shape
    sx, sy, tx, ty = to_coords(where_logits)
    Xn = torch.linspace(-1.0, 1.0, W, dtype=glimpse.dtype)
    Yn = torch.linspace(-1.0, 1.0, H, dtype=glimpse.dtype)
    xg = 0.5 * (G - 1) * ((Xn[None, :] - tx[:, None]) / sx[:, None] + 1.0)
    yg = 0.5 * (G - 1) * ((Yn[None, :] - ty[:, None]) / sy[:, None] + 1.0)
    return bilinear_gather(glimpse, xg, yg)


def stn_to_pixel_coords(stn, img_size):
    """modules.py:246-262 -> (y, x, h, w)."""
    sx, sy, tx, ty = [stn[..., i] for i in range(4)]

    def conv(s, t, L):
        return 0.5 * (L - 1.0) * (t - s + 1.0), (L + 1.0) * s
    y, h = conv(sy, ty, img_size[0])
    x, w = conv(sx, tx, img_size[1])
    return np.stack([y, x, h, w], -1)


def pixel_to_stn_coords(yxhw, img_size):
    """modules.py:264-280 -> (sx, sy, tx, ty)."""
    yxhw = np.asarray(yxhw, dtype=np.float64)
    size = np.asarray(img_size, dtype=np.float64)
    scale = yxhw[..., 2:] / (size + 1.0)
    shift = 2.0 * yxhw[..., :2] / (size - 1.0) + scale - 1.0
    sy, sx = scale[..., 0], scale[..., 1]
    ty, tx = shift[..., 0], shift[..., 1]
    return np.stack([sx, sy, tx, ty], -1)


def to_logits(coords, eps=1e-4):
    """modules.py:229-243."""
    coords = np.asarray(coords, dtype=np.float64)
    scale, shift = coords[..., :2], coords[..., 2:]
    s = np.clip(scale, eps, 1.0 - eps)
    sl = np.log(s / (1.0 - s))
    sh = np.clip(shift, eps - 1.0, 1.0 - eps)
    shl = 0.5 * (np.log(1.0 + sh) - np.log(1.0 - sh))
    return np.concatenate([sl, shl], -1)


# ----------------------------------------------------------------------------- priors / index helpers
def bernoulli_to_modified_geometric(p):
    """prior.py:61-67, computed in float64 and cast back: [1-p1, p1(1-p2), ..., prod p] / sum."""
    dt = p.dtype
    p = p.double()
    inv = 1.0 - p
    prob = torch.cumprod(p, dim=-1)
    mod = torch.cat([inv[..., :1], inv[..., 1:] * prob[..., :-1], prob[..., -1:]], -1)
    mod = mod / mod.sum(-1, keepdim=True)
    return mod.to(dt)


def num_steps_log_prob(joint, n):
    """NumStepsDistribution.log_prob (prior.py:93-101, index.py:48-71):
    log clip_preserve(joint[n], 1e-16, 1)."""
    pr = joint.gather(-1, n.long().unsqueeze(-1)).squeeze(-1)
    pr = pr + (pr.clamp(1e-16, 1.0) - pr).detach()
    return torch.log(pr)


def tile_input_for_iwae(x, K, with_time=True):
    """index.py:106-129: repeats every sequence K times contiguously (b' = b*K + k)."""
    ax = 1 if with_time else 0
    return torch.repeat_interleave(x, K, dim=ax)


def select_present(x, presence):
    """index.py:132-165: per-row stable partition, present entries first.  x [B,S,d],
    presence [B,S]."""
    order = torch.sort(1.0 - presence, dim=1, stable=True).indices
    return x.gather(1, order.unsqueeze(-1).expand(-1, -1, x.shape[-1]))


def compute_object_ids(last_used_id, prev_ids, prop_pres, disc_pres):
    """index.py:198-221.  last_used_id [B,1]; prev_ids, *_pres [B,N,1]."""
    prop_ids = prev_ids * prop_pres - (1.0 - prop_pres)
    inc = torch.cumsum(disc_pres, 1)
    disc_ids = inc + last_used_id[:, None]
    last_used_id = last_used_id + inc[:, -1]
    disc_ids = disc_ids * disc_pres - (1.0 - disc_pres)
    return last_used_id, torch.cat([prop_ids, disc_ids], 1)


# ----------------------------------------------------------------------------- targets (targets.py)
def iwae(log_w):
    """targets.py:38-43."""
    return torch.logsumexp(log_w, -1) - math.log(float(log_w.shape[-1]))


def vimco_control_variate(t):
    """targets.py:46-59."""
    K = t.shape[-1]
    s = t.sum(-1, keepdim=True)
    abo = (s - t) / (K - 1.0)
    base = t[..., None] + torch.diag_embed(abo - t)
    return torch.logsumexp(base, -2) - math.log(float(K))


def vimco(log_w, log_probs, elbo_iwae=None):
    """targets.py:62-75 (learning signal = stop_gradient(log_w - control variate))."""
    cv = vimco_control_variate(log_w)
    signal = (log_w - cv).detach()
    log_probs = log_probs.reshape(log_w.shape)
    if elbo_iwae is None:
        elbo_iwae = iwae(log_w)
    return (-elbo_iwae[..., None] - signal * log_probs).mean()


def reinforce(log_w, log_probs, elbo_iwae=None):
    """targets.py:78-89 (learning signal = stop_gradient(log_w): no control variate)."""
    signal = log_w.detach()
    log_probs = log_probs.reshape(log_w.shape)
    if elbo_iwae is None:
        elbo_iwae = iwae(log_w)
    return (-elbo_iwae[..., None] - signal * log_probs).mean()


def ess(w):
    """ops.py:52-59 with average=True."""
    return (w.sum(-1) ** 2 / (w ** 2).sum(-1)).mean()


# ----------------------------------------------------------------------------- the model
class SqairOracle(object):
    """Restatement of Model(SequentialAIR(SQAIRTimestep(Discover, Propagate), AIRDecoder))
    as wired by configs/mlp_mnist_model.py:74-150."""

    def __init__(self, params, cfg, dtype=torch.float64, requires_grad=False):
        self.cfg = cfg
        self.dtype = dtype
        if cfg.transition not in ("VanillaRNN", "GRU", "LSTM") or cfg.time_transition not in ("VanillaRNN", "GRU", "LSTM") or cfg.prior_transition not in ("VanillaRNN", "GRU", "LSTM"):
            raise NotImplementedError("oracle restates transition / time_transition / prior_transition in {VanillaRNN, GRU, LSTM}")
        if cfg.prop_prior_type not in ("rnn", "rw", "guided"):
            raise ValueError('Invalid prior type: "{}"'.format(cfg.prop_prior_type))  # propagate.py:42-43
        if cfg.disc_prior_type not in ("cat", "geom"):
            raise ValueError("Invalid prior type: {}".format(cfg.disc_prior_type))  # sqair_modules.py:224
        self.P = OrderedDict()
        for k, v in params.items():
            t = torch.as_tensor(np.asarray(v), dtype=dtype).clone()
            t.requires_grad_(requires_grad)
            self.P[k] = t

    # ---- shared encoders -------------------------------------------------------------
    def input_encoder(self, img):
        """Encoder(n_hiddens) on the flattened image (modules.py:100-112)."""
        return mlp2_hidden(self.P, "enc.input", img.reshape(img.shape[0], -1))

    def glimpse_mask(self, mask_inpt):
        """modules.py:322-324,350-356: sigmoid(MLP[256->128->G*G]), out-bias init 1."""
        return mlp_1hidden_out(self.P, "enc.mask", mask_inpt, torch.sigmoid)

    def air_encoder(self, img, where, mask_inpt=None):
        """AIREncoder._build (modules.py:326-364) + GaussianFromParamVec (:39-74)."""
        c = self.cfg
        glimpse = st_crop(img, where, c.G).reshape(img.shape[0], c.G * c.G)
        if c.masked_glimpse and mask_inpt is not None:
            glimpse = glimpse * self.glimpse_mask(mask_inpt)
        feat = mlp2_hidden(self.P, "enc.glimpse", glimpse)
        return self.gaussian_head("enc.what_head", feat)

    def gaussian_head(self, name, feat):
        s = linear(self.P, name, feat)
        n = s.shape[-1] // 2
        return s[..., :n], softplus(s[..., n:] + 0.0) + self.cfg.min_std

    def steps_logit(self, name, prev_presence, feats):
        """StepsPredictor._build (modules.py:506-524) up to the Bernoulli."""
        logit = mlp_1hidden_out(self.P, name, torch.cat(feats, -1))
        return prev_presence * logit + (prev_presence - 1.0) * 88.0

    def compute_presence(self, name, prev_presence, feats, u):
        """core.py:141-144: Bernoulli(logits).sample() * previous presence; sample = 1[u < p]."""
        logit = self.steps_logit(name, prev_presence, feats)
        prob = torch.sigmoid(logit)
        pres = (u < prob).to(self.dtype).detach() * prev_presence
        return pres, prob, logit

    # ---- propagation -----------------------------------------------------------------
    def propagate_prior(self, z_tm1, prior_state):
        """PropagatePrior._build (propagate.py:68-98) and the rw/guided variants (:123-158).
        Shapes [B,N,.]."""
        c = self.cfg
        what_tm1, where_tm1, pres_tm1, logit_tm1 = z_tm1
        B, N = what_tm1.shape[:2]
        x = torch.cat([what_tm1, where_tm1], -1).reshape(B * N, -1)
        ps = prior_state.reshape(B * N, -1)
        if c.prior_transition == "LSTM":  # state travels as [hidden | cell]; the Linear reads the cell OUTPUT = new hidden
            nh = c.n_hidden
            h, c2 = lstm(self.P, "prop.prior_lstm", x, ps[:, :nh], ps[:, nh:])
            new_state = torch.cat([h, c2], -1).reshape(B, N, -1)
        else:
            h = vanilla_rnn(self.P, "prop.prior_rnn", x, ps) if c.prior_transition == "VanillaRNN" else gru(self.P, "prop.prior_gru", x, ps)
            new_state = h.reshape(B, N, -1)
        stats = linear(self.P, "prop.prior_linear", h).reshape(B, N, -1)
        logit = stats[..., :1] + c.prop_prior_step_bias
        logit = pres_tm1 * logit + (pres_tm1 - 1.0) * 88.0
        rest = stats[..., 1:]
        half = rest.shape[-1] // 2
        locs, scales = rest[..., :half], rest[..., half:]
        where_loc, what_loc = locs[..., :4], locs[..., 4:]
        where_scale = softplus(scales[..., :4]) + 1e-2
        what_scale = softplus(scales[..., 4:]) + 1e-2
        if c.prop_prior_type == "rw":
            where_loc, what_loc, logit = where_tm1, what_tm1, logit_tm1 + 0.1 * logit
        elif c.prop_prior_type == "guided":
            where_loc = where_tm1 + 0.1 * where_loc
            what_loc = what_tm1 + 0.1 * what_loc
            logit = logit_tm1 + 0.1 * logit
        return (where_loc, where_scale, what_loc, what_scale, logit), new_state

    def where_tril(self, scale):
        """AffineDiagNormal._build (modules.py:535-545): L = T * scale[:, None] + diag(scale)."""
        T = fill_triangular(self.P["prop.cholesky_scale"], 4)
        return T[None] * scale[..., :, None] + torch.diag_embed(scale)

    @staticmethod
    def mvn_tril_log_prob(x, loc, L):
        """tfd.MultivariateNormalTriL.log_prob."""
        d = (x - loc).unsqueeze(-1)
        sol = torch.linalg.solve_triangular(L, d, upper=False).squeeze(-1)
        logdet = torch.log(torch.abs(torch.diagonal(L, dim1=-2, dim2=-1))).sum(-1)
        return -0.5 * (sol ** 2).sum(-1) - logdet - 0.5 * x.shape[-1] * LOG_2PI

    def propagation_core(self, img, z_tm1_k, temporal_state, state, eps_where, eps_what, u):
        """PropagationCore._build/_compute_where/_compute_what (core.py:280-359), one slot."""
        c = self.cfg
        P = self.P
        what_tm1, where_tm1, pres_tm1, logit_tm1 = z_tm1_k
        what_km1, where_km1, pres_km1, hidden = state
        # core.py:284: temporal_state = nest.flatten(temporal_hidden_state)[-1] — the GRU state itself, the CELL state of an
        # LSTM.  An LSTMState travels here as one tensor [hidden | cell] (so that compaction treats it like any feature).
        lstm_time = c.time_transition == "LSTM"
        temporal_full = temporal_state
        if lstm_time:
            temporal_state = temporal_full[..., c.n_hidden:]
        # rnn_inpt (core.py:291-304)
        where_bias = mlp_1hidden_out(P, "prop.where_bias", temporal_state) * 0.1
        loc1, _ = self.air_encoder(img, where_tm1 + where_bias, mask_inpt=temporal_state)
        rnn_inpt = torch.cat([loc1, what_km1, where_km1, pres_km1, what_tm1, where_tm1, pres_tm1,
                              temporal_state], -1)
        hidden, hidden_state = self.slot_rnn("prop", rnn_inpt, hidden)
        # where (core.py:323-334, modules.py:89-97)
        tp = linear(P, "prop.transform.l2", mlp2_hidden(P, "prop.transform",
                                                         torch.cat([hidden, where_tm1, temporal_state], -1)))
        loc = where_tm1 + c.where_update_scale * tp[..., :4]
        scale = softplus(tp[..., 4:] + P["prop.transform.scale_offset"] - 1.0) + 1e-2
        L = self.where_tril(scale)
        where = loc + (L @ eps_where.unsqueeze(-1)).squeeze(-1)
        # what (core.py:336-359)
        loc2, scale2 = self.air_encoder(img, where, mask_inpt=temporal_state)
        cell_inpt = torch.cat([hidden, where, loc2, scale2], -1)
        if lstm_time:
            temporal_out, cell_new = lstm(P, "prop.temporal_lstm", cell_inpt, temporal_full[..., :c.n_hidden], temporal_state)
            temporal_new = torch.cat([temporal_out, cell_new], -1)
        elif c.time_transition == "VanillaRNN":
            temporal_new = temporal_out = vanilla_rnn(P, "prop.temporal_rnn", cell_inpt, temporal_state)
        else:
            temporal_new = temporal_out = gru(P, "prop.temporal_gru", cell_inpt, temporal_state)
        t_loc, t_scale = self.gaussian_head("prop.what_head", temporal_out)
        gates = torch.sigmoid(linear(P, "prop.gates", temporal_out)) * 0.9999
        nw = c.n_what
        fg, ig, tg = gates[..., :nw], gates[..., nw:2 * nw], gates[..., 2 * nw:]
        what_loc = fg * what_tm1 + (1.0 - ig) * loc2 + (1.0 - tg) * t_loc
        what_scale = (1.0 - ig) * scale2 + (1.0 - tg) * t_scale
        what = what_loc + what_scale * eps_what
        # presence (core.py:312-314): features = (hidden, temporal_state (pre-update), what)
        pres, prob, logit = self.compute_presence("prop.steps", pres_tm1, [hidden, temporal_state, what], u)
        out = dict(what=what, what_loc=what_loc, what_scale=what_scale, where=where, where_loc=loc,
                   where_scale=scale, presence_prob=prob, presence=pres, presence_logit=logit,
                   temporal_state=temporal_new)
        return out, (what, where, pres, hidden_state)

    def slot_rnn(self, core, rnn_inpt, state):
        """core.py:187-189 / :304-305: hidden_output, hidden_state = cell(rnn_inpt, hidden_state).  VanillaRNN: both are the
        new hidden vector; LSTM (transition=LSTM): the state travels as [hidden | cell], the output is the new hidden."""
        if self.cfg.transition == "LSTM":
            nh = self.cfg.n_hidden
            h, c2 = lstm(self.P, core + ".rnn_lstm", rnn_inpt, state[..., :nh], state[..., nh:])
            return h, torch.cat([h, c2], -1)
        if self.cfg.transition == "GRU":
            h = gru(self.P, core + ".rnn_gru", rnn_inpt, state)
        else:
            h = vanilla_rnn(self.P, core + ".rnn", rnn_inpt, state)
        return h, h

    def initial_rnn_state(self, core):
        if self.cfg.transition == "LSTM":
            return torch.cat([self.P[core + ".rnn_init"], self.P[core + ".rnn_init_c"]], -1)
        return self.P[core + ".rnn_init"]

    def propagate(self, img, z_tm1, temporal_state, prior_state, noise, gen_noise=None, do_generate=False):
        """Propagate._build/_compute_log_probs (sqair_modules.py:250-329) + SequentialSSM
        (propagate.py:168-184)."""
        c = self.cfg
        B = img.shape[0]
        N, nw = c.N, c.n_what
        prior_stats, prior_state = self.propagate_prior(z_tm1, prior_state)
        zeros = lambda n: torch.zeros(B, n, dtype=self.dtype)
        state = (zeros(nw), zeros(4), zeros(1), self.initial_rnn_state("prop").expand(B, -1))
        outs = []
        for k in range(N):
            z_k = tuple(z[:, k] for z in z_tm1)
            o, state = self.propagation_core(img, z_k, temporal_state[:, k], state,
                                             noise[:, k, 0:4], noise[:, k, 4:4 + nw], noise[:, k, 4 + nw:])
            outs.append(o)
        ho = {k: torch.stack([o[k] for o in outs], 1) for k in outs[0]}
        pres = ho["presence"].squeeze(-1)  # the posterior path's presence: masks and prior Bernoulli use it even when
        pres_tm1 = z_tm1[2].squeeze(-1)    # the hidden outputs are replaced below (sqair_modules.py:286, :310-318)
        pw_loc, pw_scale, pa_loc, pa_scale, p_logit = prior_stats
        s_what, s_where, s_pres = ho["what"], ho["where"], pres
        if c.sample_from_prior:
            # sqair_modules.py:294-302: samples = [p.sample() for p in priors] replace the points at which the
            # POSTERIORS are evaluated; the hidden outputs themselves only when do_generate
            s_what = pa_loc + pa_scale * gen_noise[..., 4:4 + nw]
            s_where = pw_loc + pw_scale * gen_noise[..., 0:4]
            s_pres = (gen_noise[..., 4 + nw] < torch.sigmoid(p_logit.squeeze(-1))).to(self.dtype)
            if do_generate:
                ho["what"], ho["where"], ho["presence"] = s_what, s_where, s_pres.unsqueeze(-1)
        # posteriors (sqair_modules.py:49-60, :328-329)
        q_what = normal_log_prob(s_what, ho["what_loc"], ho["what_scale"]).sum(-1)
        q_where = self.mvn_tril_log_prob(s_where, ho["where_loc"], self.where_tril(ho["where_scale"]))
        q_pres = bernoulli_log_prob(s_pres, ho["presence_logit"].squeeze(-1))
        p_what = normal_log_prob(ho["what"], pa_loc, pa_scale).sum(-1)
        p_where = normal_log_prob(ho["where"], pw_loc, pw_scale).sum(-1)
        p_pres = bernoulli_log_prob(pres, p_logit.squeeze(-1))
        prop_prob = torch.exp(q_pres) * pres_tm1
        m = pres_tm1 * pres
        q_what, q_where, p_what, p_where = q_what * m, q_where * m, p_what * m, p_where * m
        q_pres = (q_pres * pres_tm1).sum(-1)
        p_pres = (p_pres * pres_tm1).sum(-1)
        o = dict(ho)
        o.update(prior_stats=prior_stats, prior_state=prior_state, num_steps=pres.sum(-1),
                 q_z_given_x=(q_what + q_where).sum(-1) + q_pres, p_z=(p_what + p_where).sum(-1) + p_pres,
                 what_log_prob=q_what, where_log_prob=q_where, prop_log_prob=q_pres,
                 what_prior_log_prob=p_what, where_prior_log_prob=p_where, prop_prior_log_prob=p_pres,
                 prop_prob=prop_prob)
        return o

    # ---- discovery -------------------------------------------------------------------
    def discovery_core(self, img, conditioning, state, eps_where, eps_what, u):
        """DiscoveryCore._build (core.py:192-227); the input encoder is re-evaluated at every
        step exactly as the reference graph does (core.py:165)."""
        P = self.P
        what_prev, where_prev, pres_prev, hidden = state
        rnn_inpt = torch.cat([self.input_encoder(img), conditioning, what_prev, where_prev, pres_prev], -1)
        hidden, hidden_state = self.slot_rnn("disc", rnn_inpt, hidden)
        tp = linear(P, "disc.transform.l2", mlp2_hidden(P, "disc.transform", hidden))
        where_loc = tp[..., :4]
        where_scale = softplus(tp[..., 4:] + P["disc.transform.scale_offset"]) + 1e-2
        where = where_loc + where_scale * eps_where
        what_loc, what_scale = self.air_encoder(img, where)
        what = what_loc + what_scale * eps_what
        pres, prob, logit = self.compute_presence("disc.steps", pres_prev, [hidden, what], u)
        out = dict(what=what, what_loc=what_loc, what_scale=what_scale, where=where, where_loc=where_loc,
                   where_scale=where_scale, presence_prob=prob, presence=pres, presence_logit=logit)
        return out, (what, where, pres, hidden_state)

    def recurrent_normal_log_prob(self, samples, conditioning):
        """RecurrentNormal.log_prob -> RecurrentNormalImpl (modules.py:548-611).  The RNN state is
        computed once from the conditioning and never advanced in the loop (:582-593)."""
        P = self.P
        B, N = samples.shape[:2]
        state = torch.cat([P["disc.rn.init_state"].expand(B, -1), conditioning], -1)
        state = elu(linear(P, "disc.rn.cond", state))
        sample = P["disc.rn.init_sample"].expand(B, -1)
        lps = []
        for k in range(N):
            o = vanilla_rnn(P, "disc.rn", sample, state)
            st = linear(P, "disc.rn.readout", o)
            loc, scale = st[..., :4], softplus(st[..., 4:]) + 1e-2
            sample = samples[:, k]
            lps.append(normal_log_prob(sample, loc, scale))
        return torch.stack(lps, 1)

    def recurrent_normal_sample(self, eps, conditioning):
        """RecurrentNormal.sample (modules.py:619-629): x_k = loc(x_{k-1}) + scale(x_{k-1}) eps_k, same never-advanced
        RNN state as log_prob."""
        P = self.P
        B, N = eps.shape[:2]
        state = torch.cat([P["disc.rn.init_state"].expand(B, -1), conditioning], -1)
        state = elu(linear(P, "disc.rn.cond", state))
        sample = P["disc.rn.init_sample"].expand(B, -1)
        xs = []
        for k in range(N):
            o = vanilla_rnn(P, "disc.rn", sample, state)
            st = linear(P, "disc.rn.readout", o)
            loc, scale = st[..., :4], softplus(st[..., 4:]) + 1e-2
            sample = loc + scale * eps[:, k]
            xs.append(sample)
        return torch.stack(xs, 1)

    def discover(self, img, conditioning, prior_conditioning, t, noise, gen_noise=None, do_generate=False):
        """Discover._build/_discover/_compute_log_probs/_make_priors (sqair_modules.py:94-229)."""
        c = self.cfg
        P = self.P
        B = img.shape[0]
        N, nw = c.N, c.n_what
        zeros = lambda n: torch.zeros(B, n, dtype=self.dtype)
        state = (zeros(nw), zeros(4), torch.ones(B, 1, dtype=self.dtype), self.initial_rnn_state("disc").expand(B, -1))
        outs = []
        for j in range(N):
            o, state = self.discovery_core(img, conditioning, state, noise[:, j, 0:4], noise[:, j, 4:4 + nw],
                                           noise[:, j, 4 + nw:])
            outs.append(o)
        ho = {k: torch.stack([o[k] for o in outs], 1) for k in outs[0]}
        num_steps = ho["presence"].squeeze(-1).sum(-1)  # _discover (sqair_modules.py:146): before any replacement
        where_cond = torch.cat([conditioning, prior_conditioning], -1)
        if c.sample_from_prior and do_generate:
            # sqair_modules.py:157-170: what ~ N(0, I), where ~ the where prior, presence zeroed (`* 0.`)
            ho["what"] = gen_noise[..., 4:4 + nw]
            if c.rec_where_prior:
                ho["where"] = self.recurrent_normal_sample(gen_noise[..., 0:4], where_cond)
            else:
                ho["where"] = torch.tensor(c.where_prior_mean, dtype=self.dtype) + gen_noise[..., 0:4]
            ho["presence"] = torch.zeros_like(ho["presence"])
        pres = ho["presence"].squeeze(-1)
        joint = bernoulli_to_modified_geometric(ho["presence_prob"].squeeze(-1))
        q_what = normal_log_prob(ho["what"], ho["what_loc"], ho["what_scale"]).sum(-1) * pres
        q_where = normal_log_prob(ho["where"], ho["where_loc"], ho["where_scale"]).sum(-1) * pres
        q_num = num_steps_log_prob(joint, num_steps)
        # priors (sqair_modules.py:199-226)
        p_what = normal_log_prob(ho["what"], torch.zeros((), dtype=self.dtype),
                                 torch.ones((), dtype=self.dtype)).sum(-1) * pres
        if c.rec_where_prior:
            p_where = self.recurrent_normal_log_prob(ho["where"], where_cond).sum(-1) * pres
        else:
            mean = torch.tensor(c.where_prior_mean, dtype=self.dtype)
            p_where = normal_log_prob(ho["where"], mean, torch.ones(4, dtype=self.dtype)).sum(-1) * pres
        if c.disc_prior_type == "geom":
            pr = torch.tensor(1.0 - c.step_success_prob, dtype=self.dtype)
            p_num = num_steps * torch.log1p(-pr) + torch.log(pr)
        else:
            logits = P["disc.step_prior_bias"] + (0.0 if t == 0 else 1.0) * P["disc.step_prior_timestep_bias"]
            logits = logits[None] + mlp_1hidden_out(P, "disc.steps_prior", prior_conditioning)
            logits = elu(logits)
            p_num = torch.log_softmax(logits, -1).gather(-1, num_steps.long().unsqueeze(-1)).squeeze(-1)
        o = dict(ho)
        o.update(num_steps=num_steps, q_z_given_x=(q_what + q_where).sum(-1) + q_num,
                 p_z=(p_what + p_where).sum(-1) + p_num,
                 what_log_prob=q_what, where_log_prob=q_where, num_step_log_prob=q_num,
                 what_prior_log_prob=p_what, where_prior_log_prob=p_where, num_step_prior_log_prob=p_num,
                 num_steps_prob=joint)
        return o

    # ---- one time step ---------------------------------------------------------------
    def encode_latents(self, what, where, presence):
        """AbstractTimstepModule._encode_latents (sqair_modules.py:368-385), no relation embedding."""
        B, N = what.shape[:2]
        f = mlp2_hidden(self.P, "seq.latent_enc", torch.cat([what, where], -1).reshape(B * N, -1))
        return (f.reshape(B, N, -1) * presence).sum(-2)

    def initial_prior_state(self):
        """initial_prior_state (sqair_modules.py:352-358): trainable; [hidden | cell] for an LSTM."""
        if self.cfg.prior_transition == "LSTM":
            return torch.cat([self.P["seq.prior_init"], self.P["seq.prior_init_c"]], -1)
        return self.P["seq.prior_init"]

    def initial_temporal_state(self):
        """initial_temporal_state (sqair_modules.py:352-366): trainable; [hidden | cell] for an LSTM."""
        if self.cfg.time_transition == "LSTM":
            return torch.cat([self.P["seq.temporal_init"], self.P["seq.temporal_init_c"]], -1)
        return self.P["seq.temporal_init"]

    def timestep(self, img, z_tm1, temporal_state, prior_state, last_used_id, prev_ids, t, noise, gen_noise=None):
        """SQAIRTimestep._build/_propagate_and_discover/_choose_latents (sqair_modules.py:446-582)."""
        c = self.cfg
        B, N = img.shape[0], c.N
        do_generate = c.generate_after > 0 and t > c.generate_after  # seq.py:198-200
        gn = gen_noise if gen_noise is not None else torch.zeros_like(noise)
        prop = self.propagate(img, z_tm1, temporal_state, prior_state, noise[:, 0], gn[:, 0], do_generate)
        cond = self.encode_latents(prop["what"], prop["where"], prop["presence"])
        prior_logits = prop["prior_stats"][-1].squeeze(-1)
        exp_steps = ((torch.sigmoid(prior_logits) - 0.5) / N).sum(-1, keepdim=True)
        disc = self.discover(img, cond, exp_steps, t, noise[:, 1], gn[:, 1], do_generate)
        # merge (sqair_modules.py:514-582)
        names = "what what_loc what_scale where where_loc where_scale presence_prob presence presence_logit".split()
        init_temporal = self.initial_temporal_state()[None].expand(B, N, -1)
        init_prior = self.initial_prior_state()[None].expand(B, N, -1)
        temporal_cat = torch.cat([prop["temporal_state"], init_temporal], 1)
        prior_cat = torch.cat([prop["prior_state"], init_prior], 1)
        hidden = [torch.cat([prop[n], disc[n]], 1) for n in names]
        last_used_id, new_ids = compute_object_ids(last_used_id, prev_ids, prop["presence"], disc["presence"])
        parts = hidden + [new_ids, prior_cat, temporal_cat]
        widths = [p.shape[-1] for p in parts]
        merged = select_present(torch.cat(parts, -1), hidden[7].squeeze(-1))[:, :N]
        split = torch.split(merged, widths, -1)
        ho = dict(zip(names, split[:len(names)]))
        obj_ids, prior_state, temporal_state = split[len(names):]
        z_t = (ho["what"], ho["where"], ho["presence"], ho["presence_logit"])
        out = dict(ho)
        out.update(z_t=z_t, obj_ids=obj_ids, prior_state=prior_state, temporal_state=temporal_state,
                   last_used_id=last_used_id, prop=prop, disc=disc,
                   presence_log_prob=prop["prop_log_prob"] + disc["num_step_log_prob"],
                   p_z=disc["p_z"] + prop["p_z"], q_z_given_x=disc["q_z_given_x"] + prop["q_z_given_x"],
                   num_steps=ho["presence"].squeeze(-1).sum(-1))
        return out

    # ---- decoder ---------------------------------------------------------------------
    def decode(self, what, where, presence):
        """AIRDecoder._build/_decode/_add_mean_image (modules.py:435-467) + Decoder (:131-147).
        Returns canvas mean [B,H,W], per-pixel std, decoded glimpses [B,N,G,G]."""
        c = self.cfg
        P = self.P
        B, N = what.shape[:2]
        g = linear(P, "dec.l2", mlp2_hidden(P, "dec", what.reshape(B * N, -1))) * P["dec.output_scale"]
        glimpse = g.reshape(B * N, c.G, c.G)
        wl = where.reshape(B * N, 4)
        inv = st_insert(glimpse, wl, c.H, c.W).reshape(B, N, c.H, c.W) * presence[..., None]
        canvas = inv.sum(1)
        ones = torch.ones(B * N, c.G, c.G, dtype=self.dtype)
        nz = (st_insert(ones, wl, c.H, c.W).reshape(B, N, c.H, c.W) * presence[..., None]).sum(1)
        nz = torch.sigmoid(-10.0 + nz * 20.0)
        canvas = canvas + P["dec.mean_img"][None] * nz
        std = nz * c.output_std + (1.0 - nz) * c.background_std
        return canvas, std, glimpse.reshape(B, N, c.G, c.G)

    # ---- sequence unroll -------------------------------------------------------------
    def sequence(self, tiled_obs, noise, gen_noise=None):
        """SequentialAIR._build/_prepare_loop_vars/_loop_body/_compute_log_weights
        (seq.py:69-279).  tiled_obs [T,B',H,W]; noise [T,B',2,N,4+n_what+1]."""
        c = self.cfg
        T, B = tiled_obs.shape[:2]
        N, nw, nh = c.N, c.n_what, c.n_hidden
        dt = self.dtype
        z = (torch.zeros(B, N, nw, dtype=dt), torch.zeros(B, N, 4, dtype=dt), torch.zeros(B, N, 1, dtype=dt),
             torch.zeros(B, N, 1, dtype=dt))
        temporal = self.initial_temporal_state()[None].expand(B, N, -1)
        prior = self.initial_prior_state()[None].expand(B, N, -1)
        prev_ids = -torch.ones(B, N, 1, dtype=dt)
        last_id = -torch.ones(B, 1, dtype=dt)
        tas = OrderedDict()

        def write(name, val):
            if val.dim() > 1 and val.shape[-1] == 1:
                val = val.squeeze(-1)
            tas.setdefault(name, []).append(val)

        for t in range(T):
            img = tiled_obs[t]
            o = self.timestep(img, z, temporal, prior, last_id, prev_ids, t, noise[t], None if gen_noise is None else gen_noise[t])
            z_t = o["z_t"]
            canvas, std, glimpse = self.decode(z_t[0], z_t[1], z_t[2])
            data_ll = normal_log_prob(img, canvas, std).sum((1, 2))
            kl = o["q_z_given_x"] - o["p_z"]
            log_w = data_ll - kl
            prop, disc = o["prop"], o["disc"]
            for n in "what what_loc what_scale where where_loc where_scale presence_prob presence presence_logit".split():
                write(n, o[n])
            write("obj_id", o["obj_ids"])
            write("step_log_prob", o["presence_log_prob"])
            write("canvas", canvas)
            write("glimpse", glimpse)
            write("disc_what_log_prob", disc["what_log_prob"])
            write("disc_where_log_prob", disc["where_log_prob"])
            write("disc_what_prior_log_prob", disc["what_prior_log_prob"])
            write("disc_where_prior_log_prob", disc["where_prior_log_prob"])
            write("disc_log_prob", disc["num_step_log_prob"])
            write("disc_prior_log_prob", disc["num_step_prior_log_prob"])
            write("disc_prob", disc["num_steps_prob"])
            write("prop_what_log_prob", prop["what_log_prob"])
            write("prop_where_log_prob", prop["where_log_prob"])
            write("prop_what_prior_log_prob", prop["what_prior_log_prob"])
            write("prop_where_prior_log_prob", prop["where_prior_log_prob"])
            write("prop_log_prob", prop["prop_log_prob"])
            write("prop_prior_log_prob", prop["prop_prior_log_prob"])
            write("prop_prob", prop["prop_prob"])
            write("discrete_log_prob", prop["prop_log_prob"] + disc["num_step_log_prob"])
            write("num_prop_steps_per_sample", prop["num_steps"])
            write("num_disc_steps_per_sample", disc["num_steps"])
            write("num_steps_per_sample", o["num_steps"])
            write("prop_pres", prop["presence"])
            write("disc_pres", disc["presence"])
            write("data_ll_per_sample", data_ll)
            write("kl_per_sample", kl)
            write("log_q_z_given_x_per_sample", o["q_z_given_x"])
            write("log_p_z_per_sample", o["p_z"])
            write("log_weights_per_timestep", log_w)
            # extras (not reference outputs): pre-merge Bernoulli probabilities, used by the fixture
            # generator to reject noise draws whose |u - p| margin is too small to be decision-stable
            write("_prop_presence_prob", prop["presence_prob"])
            write("_disc_presence_prob", disc["presence_prob"])
            write("_prop_prev_presence", z[2])
            # probability of the Bernoulli the generation modes draw from the propagation PRIOR (sqair_modules.py:294-302):
            # lets the tests measure how far those decisions are from flipping, like the two posterior ones above
            write("_prop_prior_presence_prob", torch.sigmoid(prop["prior_stats"][4]))
            z, temporal, prior = z_t, o["temporal_state"], o["prior_state"]
            prev_ids, last_id = o["obj_ids"], o["last_used_id"]
        out = OrderedDict((k, torch.stack(v, 0)) for k, v in tas.items())
        # final recurrent state, exposed for state-level parity checks of the HIP path
        out["_final_temporal_state"] = temporal
        out["_final_prior_state"] = prior
        out["_final_last_used_id"] = last_id
        return out

    # ---- Model (model.py) ------------------------------------------------------------
    def model(self, obs, noise, num=None, resample_u=None, gen_noise=None):
        """Model.__init__/_build (model.py:43-148).  obs [T,B,H,W] in [0,1]; ``num`` is the
        ground-truth prefix-ones presence [T,B,n_max+1] used for the step accuracy."""
        c = self.cfg
        K = c.K
        obs = torch.as_tensor(np.asarray(obs), dtype=self.dtype)
        noise = torch.as_tensor(np.asarray(noise), dtype=self.dtype)
        T, B = obs.shape[:2]
        tiled = tile_input_for_iwae(obs, K)
        if gen_noise is not None:
            gen_noise = torch.as_tensor(np.asarray(gen_noise), dtype=self.dtype)
        if c.sample_from_prior and gen_noise is None:
            raise ValueError("sample_from_prior needs gen_noise (the prior samples' eps / u), same layout as noise")
        o = self.sequence(tiled, noise, gen_noise)
        m = SimpleNamespace(outputs=o, **{k: v for k, v in o.items() if not k.startswith("_")})
        m.log_weights = o["log_weights_per_timestep"].sum(0).reshape(B, K)
        m.elbo_vae = m.log_weights.mean()
        m.elbo_iwae_per_example = iwae(m.log_weights)
        m.elbo_iwae = m.elbo_iwae_per_example.mean()
        m.normalised_elbo_vae = m.elbo_vae / float(T)
        m.normalised_elbo_iwae = m.elbo_iwae / float(T)
        m.importance_weights = torch.softmax(m.log_weights, -1).detach()
        m.ess = ess(m.importance_weights)
        if resample_u is not None:
            cdf = torch.cumsum(m.importance_weights, -1)
            ru = torch.as_tensor(np.asarray(resample_u), dtype=self.dtype).reshape(B, 1)
            m.iw_resampling_idx = (cdf <= ru).sum(-1).clamp(max=K - 1)

        def iw_mean(x):  # model.py:202-205
            x = x.reshape(-1, B, K).mean(0)
            return (m.importance_weights * x * K).mean()

        m.data_ll = iw_mean(o["data_ll_per_sample"])
        m.log_p_z = iw_mean(o["log_p_z_per_sample"])
        m.log_q_z_given_x = iw_mean(o["log_q_z_given_x_per_sample"])
        m.kl = iw_mean(o["kl_per_sample"])
        m.mse_per_sample = ((tiled - o["canvas"]) ** 2).mean((0, 2, 3))
        m.mse = iw_mean(m.mse_per_sample)
        m.raw_mse = m.mse_per_sample.mean()
        m.num_steps = iw_mean(o["num_steps_per_sample"])
        m.num_disc_steps = iw_mean(o["num_disc_steps_per_sample"])
        m.num_prop_steps = iw_mean(o["num_prop_steps_per_sample"])
        if num is not None:
            gt = torch.as_tensor(np.asarray(num), dtype=self.dtype).sum(-1)  # [T,B]
            ns = o["num_steps_per_sample"].reshape(-1, B, K)
            acc = (gt[..., None] == ns).to(self.dtype)
            m.raw_num_step_accuracy = acc.mean()
            m.num_step_accuracy = iw_mean(acc)
        m.n_timesteps = T
        return m

    def make_target(self, m, l2_reg=0.0, vi_target="vimco"):
        """Model.make_target (model.py:150-168): VIMCO / T (+ l2); `reinforce` = the other entry of Model.VI_TARGETS (model.py:36)."""
        log_probs = m.discrete_log_prob.sum(0)
        fn = {"vimco": vimco, "reinforce": reinforce}[vi_target]
        target = fn(m.log_weights, log_probs, m.elbo_iwae_per_example) / float(m.n_timesteps)
        if l2_reg != 0.0:
            target = target + l2_reg * sum(0.5 * (p ** 2).sum() for p in self.P.values())
        return target
