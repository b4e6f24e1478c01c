"""-m gpu: unit parity of the individual HIP kernels against the oracle, through the C-ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sqair_oracle as O
from sqair_amd import _capi
from sqair_amd.flags import make_flags
from sqair_amd.model import make_config
from tests.hip_util import dev, rel_err, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    lib = _capi.lib()
    F = make_flags(k_particles=3, n_steps_per_image=4)
    cfg = make_config(F, (50, 50))
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    yield lib, h, F
    lib.sqair_destroy(h)


@pytest.mark.parametrize("M,K,N,act", [(160, 256, 256, 1), (37, 54, 109, 0), (640, 400, 256, 1), (5, 2500, 256, 1),
                                       (160, 311, 256, 2), (160, 256, 8, 0), (33, 128, 400, 3), (160, 256, 100, 4),
                                       (16, 16, 16, 0), (1, 3, 1, 0),
                                       # thousands of rows on the split-K kernel (below 6000 rows since round 5: ragged last row
                                       # tile, column tiles past N, K that is not a multiple of 16, more than two blocks of chunks)
                                       (2048, 256, 256, 1), (5120, 400, 256, 1), (2100, 311, 109, 2), (2049, 72, 8, 0),
                                       (3000, 1100, 200, 4), (2048, 40, 64, 3),
                                       # from 6000 rows: the throughput kernels -- k_linear_big with its four tile shapes
                                       # (128 x 64, 64 x 64, 128 x 64 deep K, 96 x 64), one column tile (macro-tile kernel), K <= 64
                                       # (k_linear_rows)
                                       (6400, 256, 256, 1), (6100, 311, 109, 2), (7000, 1100, 200, 4), (6400, 256, 400, 0),
                                       (6001, 72, 8, 0), (6400, 40, 64, 3),
                                       # the wide layers from 1792 rows: a few more 128 x 64 tiles than CUs (270) -> 96 x 96 tiles
                                       # (cfg-4's widest once-per-frame layer; ragged last row block and column block), 96 x 64
                                       (1920, 362, 1152, 0), (1900, 311, 1100, 1), (1920, 312, 768, 0)])
def test_linear_mfma_matches_fp64(handle, M, K, N, act):
    lib, h, _ = handle
    rng = np.random.default_rng(M * 7 + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y = torch.zeros(M, N, device="cuda")
    scratch = torch.empty(4 * ((K + 15) // 16) * ((N + 15) // 16) * 256 + 8192 + M * (K + 4), dtype=torch.float32, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(b)  # keep alive: a temporary's block is recycled by the caching allocator
    rc = lib.sqair_linear_test(h, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), M, K, N, act,
                               scratch.data_ptr(), scratch.numel() * 4, stream())
    assert rc == 0, lib.sqair_last_error(h)
    ref = torch.tensor(x, dtype=torch.float64) @ torch.tensor(w, dtype=torch.float64) + torch.tensor(b, dtype=torch.float64)
    ref = [lambda v: v, O.elu, torch.tanh, torch.sigmoid, lambda v: O.softplus(v) + 1e-2][act](ref)
    # fp32 fma chain of length K: ~1e-7 * sum|a b| (guide: 0.75-1.5e-7 at K <= 1024)
    assert np.abs(y.cpu().numpy() - ref.numpy()).max() < 2e-5 * max(1.0, np.sqrt(K / 256.0))


def test_linear_identity_asymmetric(handle):
    """A = I against an asymmetric B catches a row<->col swap in the C-write (guide section 3)."""
    lib, h, _ = handle
    n = 48
    w = (np.arange(n * n, dtype=np.float32).reshape(n, n) * 0.01)
    y = torch.zeros(n, n, device="cuda")
    scratch = torch.empty(1 << 18, dtype=torch.float32, device="cuda")
    de, dw = dev(np.eye(n, dtype=np.float32)), dev(w)
    assert lib.sqair_linear_test(h, de.data_ptr(), dw.data_ptr(), None, y.data_ptr(),
                                 n, n, n, 0, scratch.data_ptr(), scratch.numel() * 4, stream()) == 0
    assert np.array_equal(y.cpu().numpy(), w)


@pytest.mark.parametrize("M", [16, 6400])  # split-K kernel / row-slab kernel of the K <= 64 layers (rolled epilogue)
def test_activations_saturate_cleanly(handle, M):
    """Large finite pre-activations must saturate (tanh -> +-1, sigmoid -> 0 / 1, softplus -> x / 0.01, ELU -> x / -1) and never
    produce NaN: the hardware-exp2 based exponential overflowed to inf * r = NaN beyond |x| ~ 88 (round-2 advisor finding)."""
    lib, h, _ = handle
    vals = np.array([0.0, 1.0, -1.0, 44.0, -44.0, 50.0, -50.0, 88.7, -88.7, 89.0, -89.0, 100.0, -100.0, 1e4, -1e4, 3e38, -3e38],
                    dtype=np.float32)
    N = 32
    pre = np.resize(vals, N).astype(np.float32)
    x = np.zeros((M, 16), np.float32)
    w = np.zeros((16, N), np.float32)
    scratch = torch.empty(1 << 20, dtype=torch.float32, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(pre)
    with np.errstate(over="ignore"):
        p64 = pre.astype(np.float64)
        want = {1: np.where(p64 > 0, p64, np.expm1(np.minimum(p64, 0))), 2: np.tanh(p64), 3: 1.0 / (1.0 + np.exp(-p64)),
                4: np.maximum(p64, 0) + np.log1p(np.exp(-np.abs(p64))) + 1e-2}
    for act in (1, 2, 3, 4):
        y = torch.full((M, N), float("nan"), device="cuda")
        rc = lib.sqair_linear_test(h, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), M, 16, N, act,
                                   scratch.data_ptr(), scratch.numel() * 4, stream())
        assert rc == 0, lib.sqair_last_error(h)
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), (act, pre[~np.isfinite(got[0])])
        err = np.abs(got.astype(np.float64) - want[act][None, :]) / np.maximum(1.0, np.abs(want[act][None, :]))
        assert err.max() < 5e-7, (act, pre[err[0].argmax()], got[0, err[0].argmax()])
        if act == 2:
            assert np.array_equal(got[0, np.abs(pre) >= 44.0], np.sign(pre[np.abs(pre) >= 44.0]))
        if act == 3:
            assert np.array_equal(got[0, pre >= 50.0], np.ones((pre >= 50.0).sum(), np.float32))
            assert (got[0, pre <= -89.0] <= 1e-38).all()


@pytest.mark.parametrize("M,Kx", [(160, 360), (640, 54), (7, 17), (2304, 360), (2050, 54), (6400, 360), (6050, 54)])
def test_gru_step(handle, M, Kx):
    lib, h, _ = handle
    nh = 256
    rng = np.random.default_rng(Kx)
    P = {}
    flat = []
    for g in "zrh":
        P["g.w" + g] = (rng.standard_normal((Kx, nh)) / np.sqrt(Kx)).astype(np.float32)
        P["g.u" + g] = (rng.standard_normal((nh, nh)) / np.sqrt(nh)).astype(np.float32)
        P["g.b" + g] = rng.standard_normal(nh).astype(np.float32) * 0.1
        flat += [P["g.w" + g].ravel(), P["g.u" + g].ravel(), P["g.b" + g].ravel()]
    x = rng.standard_normal((M, Kx)).astype(np.float32)
    hs = rng.standard_normal((M, nh)).astype(np.float32)
    out = torch.zeros(M, nh, device="cuda")
    scratch = torch.empty(1 << 24, dtype=torch.float32, device="cuda")
    dx, dh, df = dev(x), dev(hs), dev(np.concatenate(flat))
    rc = lib.sqair_gru_test(h, dx.data_ptr(), dh.data_ptr(), df.data_ptr(),
                            out.data_ptr(), M, Kx, scratch.data_ptr(), scratch.numel() * 4, stream())
    assert rc == 0, lib.sqair_last_error(h)
    P64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in P.items()}
    ref = O.gru(P64, "g", torch.tensor(x, dtype=torch.float64), torch.tensor(hs, dtype=torch.float64))
    assert np.abs(out.cpu().numpy() - ref.numpy()).max() < 2e-5


@pytest.mark.parametrize("M,Kx", [(160, 360), (640, 54), (7, 17)])
def test_lstm_step_and_cell_adjoint(handle, M, Kx):
    """snt.LSTM step (gate GEMM over [x | h] + k_lstm_cell) against the oracle's restatement, and the cell adjoint against
    autograd through it."""
    lib, h, _ = handle
    nh = 256
    rng = np.random.default_rng(Kx + 1)
    P = {"l.w": (rng.standard_normal((Kx + nh, 4 * nh)) / np.sqrt(Kx + nh)).astype(np.float32),
         "l.b": (rng.standard_normal(4 * nh) * 0.1).astype(np.float32)}
    x = rng.standard_normal((M, Kx)).astype(np.float32)
    hs = rng.standard_normal((M, nh)).astype(np.float32)
    cs = rng.standard_normal((M, nh)).astype(np.float32)
    out = torch.zeros(M, 2 * nh, device="cuda")
    scratch = torch.empty(1 << 23, dtype=torch.float32, device="cuda")
    ts = [dev(x), dev(hs), dev(cs), dev(np.concatenate([P["l.w"].ravel(), P["l.b"]]))]
    rc = lib.sqair_lstm_test(h, *[t.data_ptr() for t in ts], out.data_ptr(), M, Kx, scratch.data_ptr(), scratch.numel() * 4, stream())
    assert rc == 0, lib.sqair_last_error(h)
    P64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in P.items()}
    h2, c2 = O.lstm(P64, "l", torch.tensor(x, dtype=torch.float64), torch.tensor(hs, dtype=torch.float64),
                    torch.tensor(cs, dtype=torch.float64))
    got = out.cpu().numpy()
    assert np.abs(got[:, :nh] - h2.numpy()).max() < 2e-5
    assert np.abs(got[:, nh:] - c2.numpy()).max() < 2e-5
    # cell adjoint
    gates = torch.tensor(rng.standard_normal((M, 4 * nh)), dtype=torch.float64, requires_grad=True)
    cp = torch.tensor(cs, dtype=torch.float64, requires_grad=True)
    i, j, f, o = torch.chunk(gates, 4, -1)
    cn = torch.sigmoid(f + 1.0) * cp + torch.sigmoid(i) * torch.tanh(j)
    hn = torch.tanh(cn) * torch.sigmoid(o)
    dh = rng.standard_normal((M, nh)).astype(np.float32)
    dc = rng.standard_normal((M, nh)).astype(np.float32)
    ((hn * torch.tensor(dh, dtype=torch.float64)).sum() + (cn * torch.tensor(dc, dtype=torch.float64)).sum()).backward()
    d_g = torch.zeros(M, 4 * nh, device="cuda")
    d_cp = torch.zeros(M, nh, device="cuda")
    ts = [dev(gates.detach().numpy().astype(np.float32)), dev(cs), dev(dh), dev(dc)]
    rc = lib.sqair_lstm_cell_bwd_test(h, *[t.data_ptr() for t in ts], d_g.data_ptr(), d_cp.data_ptr(), M, stream())
    assert rc == 0, lib.sqair_last_error(h)
    assert np.abs(d_g.cpu().numpy() - gates.grad.numpy()).max() < 1e-5
    assert np.abs(d_cp.cpu().numpy() - cp.grad.numpy()).max() < 1e-5


@pytest.mark.parametrize("hw", [(50, 50), (128, 128), (37, 60)])
@pytest.mark.parametrize("masked", [False, True])
def test_st_crop(hw, masked):
    lib = _capi.lib()
    F = make_flags(k_particles=3, n_steps_per_image=4)
    cfg = make_config(F, hw)
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    try:
        B, K, G = 5, 3, 20
        rng = np.random.default_rng(hw[0])
        img = rng.uniform(size=(B,) + hw).astype(np.float32)
        where = (rng.standard_normal((B * K, 4)) * 1.5).astype(np.float32)
        where[0] = [30.0, 30.0, 0.0, 0.0]      # full-frame glimpse
        where[1] = [-30.0, -30.0, 3.0, -3.0]   # scale clipped at 1e-4, far corner
        where[2] = [0.0, 0.0, 5.0, 5.0]        # mostly outside the frame
        mask = rng.uniform(size=(B * K, G * G)).astype(np.float32) if masked else None
        out = torch.zeros(B * K, G * G, device="cuda")
        di, dw, dm = dev(img), dev(where), (dev(mask) if masked else None)
        rc = lib.sqair_st_crop(h, di.data_ptr(), dw.data_ptr(), dm.data_ptr() if masked else None,
                               out.data_ptr(), B, stream())
        assert rc == 0
        torch.cuda.synchronize()
        ref = O.st_crop(torch.tensor(np.repeat(img, K, 0), dtype=torch.float64), torch.tensor(where, dtype=torch.float64), G)
        ref = ref.reshape(B * K, -1)
        if masked:
            ref = ref * torch.tensor(mask, dtype=torch.float64)
        assert np.abs(out.cpu().numpy() - ref.numpy()).max() < 1e-5   # fwd 1e-6-class (SURVEY build plan step 3)
    finally:
        lib.sqair_destroy(h)


@pytest.mark.parametrize("hw,n_slots,G", [((50, 50), 4, 20), ((128, 128), 4, 20), ((77, 130), 4, 20), ((40, 200), 3, 20), ((128, 128), 7, 20),
                                          ((30, 250), 8, 20), ((24, 300), 4, 20), ((90, 65), 1, 20), ((600, 100), 3, 20),
                                          ((128, 128), 4, 5), ((100, 128), 3, 12), ((128, 128), 2, 2)])
def test_st_insert_loglik(hw, n_slots, G):
    # 65 .. 256 columns and up to 8 slots: the row-wave kernel (sqair_canvas.h); the others the band kernel
    lib = _capi.lib()
    F = make_flags(k_particles=2, n_steps_per_image=n_slots, glimpse_size=G)
    cfg = make_config(F, hw)
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    try:
        B, K, N = 3, 2, n_slots
        R = B * K
        H, W = hw
        rng = np.random.default_rng(3)
        gl = (rng.standard_normal((R, N, G * G)) * 0.3).astype(np.float32)
        where = (rng.standard_normal((R, N, 4))).astype(np.float32)
        pres = (rng.uniform(size=(R, N)) > 0.4).astype(np.float32)
        pres[0] = 0.0
        img = rng.uniform(size=(B, H, W)).astype(np.float32)
        mean_img = rng.uniform(size=(H, W)).astype(np.float32)
        canvas = torch.zeros(R, H, W, device="cuda")
        dll = torch.zeros(R, device="cuda")
        dg, dw, dp, di, dm = dev(gl), dev(where), dev(pres), dev(img), dev(mean_img)
        rc = lib.sqair_st_insert_loglik(h, dg.data_ptr(), dw.data_ptr(), dp.data_ptr(),
                                        di.data_ptr(), dm.data_ptr(), canvas.data_ptr(),
                                        dll.data_ptr(), B, stream())
        assert rc == 0
        D = torch.float64
        ocfg = O.make_cfg(F, hw)
        g64 = torch.tensor(gl, dtype=D).reshape(R * N, G, G)
        w64 = torch.tensor(where, dtype=D).reshape(R * N, 4)
        p64 = torch.tensor(pres, dtype=D)
        inv = O.st_insert(g64, w64, H, W).reshape(R, N, H, W) * p64[..., None, None]
        nz = (O.st_insert(torch.ones_like(g64), w64, H, W).reshape(R, N, H, W) * p64[..., None, None]).sum(1)
        nz = torch.sigmoid(-10.0 + 20.0 * nz)
        cv = inv.sum(1) + torch.tensor(mean_img, dtype=D)[None] * nz
        std = nz * ocfg.output_std + (1 - nz) * ocfg.background_std
        ll = O.normal_log_prob(torch.tensor(np.repeat(img, K, 0), dtype=D), cv, std).sum((1, 2))
        assert np.abs(canvas.cpu().numpy() - cv.numpy()).max() < 2e-5
        assert rel_err(dll.cpu().numpy(), ll.numpy()) < 1e-5
    finally:
        lib.sqair_destroy(h)


@pytest.mark.parametrize("B,K,T", [(32, 5, 10), (3, 2, 4), (7, 64, 3), (256, 5, 10)])
def test_elbo_iwae_vimco(B, K, T):
    lib = _capi.lib()
    F = make_flags(k_particles=K)
    cfg = make_config(F, (50, 50))
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    try:
        rng = np.random.default_rng(B + K)
        lw_t = (rng.standard_normal((T, B * K)) * 30.0 + 500.0).astype(np.float32)
        dl_t = (rng.standard_normal((T, B * K)) * 2.0 - 3.0).astype(np.float32)
        x_t = rng.standard_normal((T, B * K)).astype(np.float32)
        d_lw, d_dl, d_x = dev(lw_t), dev(dl_t), dev(x_t)
        lw = torch.zeros(B, K, device="cuda"); el = torch.zeros(B, device="cuda"); iw = torch.zeros(B, K, device="cuda")
        sig = torch.zeros(B, K, device="cuda"); sc = torch.zeros(16, device="cuda"); mo = torch.zeros(8, device="cuda")
        means = (C.c_void_p * 8)(d_x.data_ptr(), *([None] * 7))
        rc = lib.sqair_elbo(h, d_lw.data_ptr(), d_dl.data_ptr(), T, B, lw.data_ptr(), el.data_ptr(), iw.data_ptr(),
                            sig.data_ptr(), sc.data_ptr(), means, 1, mo.data_ptr(), stream())
        assert rc == 0
        D = torch.float64
        LW = torch.tensor(lw_t, dtype=D).sum(0).reshape(B, K)
        DL = torch.tensor(dl_t, dtype=D).sum(0).reshape(B, K)
        el_ref = O.iwae(LW)
        tgt = O.vimco(LW, DL, el_ref) / T
        w_ref = torch.softmax(LW, -1)
        assert rel_err(lw.cpu().numpy(), LW.numpy()) < 1e-6
        assert rel_err(el.cpu().numpy(), el_ref.numpy()) < 1e-6
        assert np.abs(iw.cpu().numpy() - w_ref.numpy()).max() < 5e-3  # fp32 sums of ~5000-nat weights: ulp 5e-4
        s = sc.cpu().numpy()
        assert abs(s[0] - float(LW.mean())) < 1e-6 * abs(float(LW.mean()))
        assert abs(s[1] - float(el_ref.mean())) < 1e-6 * abs(float(el_ref.mean()))
        assert abs(s[2] - float(tgt)) < 2e-4 * abs(float(tgt))
        assert abs(s[3] - float(O.ess(w_ref))) < 1e-3
        xm = torch.tensor(x_t, dtype=D).reshape(T, B, K).mean(0)
        assert abs(mo.cpu().numpy()[0] - float((w_ref * xm * K).mean())) < 2e-3
        sig_ref = LW - O.vimco_control_variate(LW)
        assert np.abs(sig.cpu().numpy() - sig_ref.numpy()).max() < 5e-3
    finally:
        lib.sqair_destroy(h)


@pytest.mark.gpu
def test_device_noise_is_standard_and_shard_invariant():
    """sqair_fill_noise: N(0,1) / U[0,1) moments, determinism in (seed, step), and the data-parallel property: a rank that
    owns sequences [b0, b0 + B) of the global batch draws exactly the rows of the one-GPU draw."""
    from sqair_amd.flags import make_flags
    from sqair_amd.model import SqairCore
    F = make_flags(k_particles=5, n_steps_per_image=4)
    core = SqairCore(F, (50, 50))
    core.bind(10, 32, "minimal")
    core.draw_noise(seed=7, step=3)
    torch.cuda.synchronize()
    full = core.noise.clone()
    eps, u = full[..., :-1].double(), full[..., -1].double()
    assert abs(float(eps.mean())) < 5e-3 and abs(float(eps.var()) - 1.0) < 1e-2
    assert abs(float((eps ** 4).mean()) - 3.0) < 0.1                       # kurtosis of a Normal
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 5e-3
    assert abs(float(u.var()) - 1.0 / 12.0) < 2e-3
    core.draw_noise(seed=7, step=3)
    torch.cuda.synchronize()
    assert torch.equal(core.noise, full)
    core.draw_noise(seed=7, step=4)
    torch.cuda.synchronize()
    assert not torch.equal(core.noise, full)
    assert abs(float((core.noise[..., :-1] * full[..., :-1]).mean())) < 5e-3  # consecutive steps are uncorrelated
    shard = SqairCore(F, (50, 50))
    shard.bind(10, 8, "minimal")
    shard.draw_noise(seed=7, step=3, global_batch=32, b0=16)
    torch.cuda.synchronize()
    assert torch.equal(shard.noise, full[:, 16 * 5:24 * 5])


def test_dense_launch_census_and_graph_node_time():
    """The measurement helpers behind profiles/r06_dense_b2b.json (tools/dense_graph_time.py): the host-side launch log reports the
    746 dense launches bench.py's roofline divides the step's FLOPs by (60 of them the VanillaRNN layer with the slot tail in
    front), and a HIP graph of one dense shape times a node in the low microseconds, a dependent chain not faster than a repeated one
    by more than noise."""
    from sqair_amd.data import config_inputs
    from sqair_amd.model import Model, SqairCore
    from tests.hip_util import params32
    ov, obs, nums, _ = config_inputs(2)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    core = SqairCore(F, hw)
    core.set_params(params32(F, hw, 0, 0.02, obs.mean((0, 1))))
    m = Model(obs, None, core, int(F.k_particles), presence=nums, outputs="minimal")
    lib, h = core.lib, core.handle
    m.run(use_graph=False)
    assert lib.sqair_debug_dense_log(h, 1) == 0
    m.run(use_graph=False)
    n = lib.sqair_debug_dense_log(h, 0)
    e = (C.c_int * 4)()
    fused = 0
    for i in range(n):
        assert lib.sqair_debug_dense_log_entry(h, i, e) == 0
        fused += e[0] < 0
        assert e[1] >= 160 and e[2] % 16 == 0 and e[3] >= 4
    assert (n, fused) == (746, 60)
    m.run(use_graph=False)
    assert lib.sqair_debug_dense_log(h, 0) == 746, "the log is off: a further pass adds nothing"
    torch.cuda.set_stream(core.stream)
    s = C.c_void_p(core.stream.cuda_stream)
    M, K, N = 160, 256, 256
    x, w, b = torch.randn(M, K, device="cuda") * 0.1, torch.randn(K, N, device="cuda") / 16, torch.zeros(N, device="cuda")
    y = torch.zeros(M, N, device="cuda")
    scratch = torch.zeros(2 * 16 * 16 * 256 + 2 * 16 * 16 + 256 + 2 * M * (K + 4) + 128, device="cuda")
    us = {}
    for dep in (0, 1):
        out = C.c_float()
        assert lib.sqair_debug_linear_graph_time(h, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, 2, scratch.data_ptr(),
                                                 scratch.numel() * 4, 200, 5, dep, C.byref(out), s) == 0, lib.sqair_last_error(h)
        us[dep] = float(out.value)
    torch.cuda.set_stream(torch.cuda.default_stream())
    assert 1.0 < us[0] < 12.0 and 1.0 < us[1] < 12.0 and us[1] > us[0] - 0.5, us   # (loose: a timing, not a benchmark)
