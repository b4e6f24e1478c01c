"""-m gpu: the in-launch slot chain (sqair_set_option "slot_chain", sqair_amd/csrc/sqair_chain.h) against the launch-per-op path.

The chain restates the slot kernels with the same arithmetic in the same order, so the bar is BIT-identity of every output, of
the graph replay, and of every gradient -- the oracle comparisons of the other test files then carry over unchanged."""
import numpy as np
import pytest
import torch

from sqair_amd.data import make_sequences, to_float
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from tests.hip_util import draw_noise, params32

pytestmark = pytest.mark.gpu


def _inputs(B, K, N, T, hw, seed=3, obj_size=None, **flags):
    F = make_flags(k_particles=K, n_steps_per_image=N, **flags)
    # (obj_size: tests/fuzz_forward.py draws canvases down to 8 x 8 -- the generator, like the reference's, re-draws a sample until
    #  its objects fit without overlap, i.e. for ever when they cannot)
    d = make_sequences(B, T=T, canvas=hw, seed=seed, **({"obj_size": obj_size} if obj_size else {}))
    obs = to_float(d["imgs"])
    P = params32(F, hw, 0, 0.05, obs.mean((0, 1)))
    noise = draw_noise(np.random.default_rng(seed), T, B * K, N, 4 + int(F.n_what) + 1)
    return F, d, obs, P, noise


def _run(F, hw, d, obs, P, noise, K, chain, use_graph):
    core = SqairCore(F, hw, options={"slot_chain": 1} if chain else None)
    core.set_params(P)
    m = Model(obs, None, core, K, presence=d["nums"], debug=chain)   # (debug: also asserts every chain launch completed)
    m.run(noise=noise, use_graph=use_graph)
    torch.cuda.synchronize()
    out = {k: v.detach().cpu().numpy().copy() for k, v in core.out.items()}
    out["log_weights"] = core.log_weights.cpu().numpy().copy()
    out["scalars"] = core.scalars.cpu().numpy().copy()
    return core, m, out


@pytest.mark.parametrize("B,K,N,T,hw", [(4, 2, 3, 3, (50, 50)),      # one row tile
                                        (32, 5, 4, 3, (50, 50)),     # BASELINE configs[1] rows: ten row tiles over eight XCDs
                                        (7, 3, 5, 2, (37, 41)),      # ragged rows, a frame that is not a multiple of 4 floats
                                        (5, 5, 2, 2, (72, 64)),      # a frame beyond the LDS-staged crop
                                        (64, 5, 6, 2, (50, 50)),     # configs[3] rows: twenty row tiles, several per XCD
                                        (1, 1, 1, 2, (50, 50)),      # one row, one slot: a chain of a single slot-step
                                        (17, 2, 1, 2, (50, 50))])    # 34 rows: a last row tile of 2 rows, one slot
def test_chain_is_bit_identical_to_the_launches(B, K, N, T, hw):
    F, d, obs, P, noise = _inputs(B, K, N, T, hw)
    _, _, ref = _run(F, hw, d, obs, P, noise, K, chain=False, use_graph=False)
    for use_graph in (False, True):
        core, m, got = _run(F, hw, d, obs, P, noise, K, chain=True, use_graph=use_graph)
        for k, v in ref.items():
            assert np.array_equal(v, got[k], equal_nan=True), (k, use_graph)
        if use_graph:   # the chain replaces the slot loop's launches: two launches per frame instead of ~7 per slot and phase
            plain = SqairCore(F, hw)
            plain.set_params(P)
            Model(obs, None, plain, K, presence=d["nums"]).run(noise=noise, use_graph=True)
            n_chain, n_plain = core.lib.sqair_graph_nodes(core.handle), plain.lib.sqair_graph_nodes(plain.handle)
            assert n_chain < (n_plain // 2 if N > 1 else n_plain), (n_chain, n_plain)   # (one slot: little to replace)


def test_chain_passes_with_fresh_noise_and_frames_on_one_workspace():
    """Consecutive passes on ONE workspace and ONE captured graph with different noise and frames: a hand-off buffer missing from
    the pass's sentinel fill would let a consumer read the previous pass's (different) value instead of waiting -- with the same
    inputs every pass (the other tests) a stale value equals the right one."""
    B, K, N, T, hw = 32, 5, 4, 2, (50, 50)
    F, d, obs, P, noise = _inputs(B, K, N, T, hw)
    rng = np.random.default_rng(11)
    cores = []
    for chain in (False, True):
        core = SqairCore(F, hw, options={"slot_chain": 1} if chain else None)
        core.set_params(P)
        cores.append((core, Model(obs, None, core, K, presence=d["nums"], debug=chain)))
    prev = None
    for it in range(4):
        nz = draw_noise(rng, T, B * K, N, 4 + int(F.n_what) + 1)
        ob = np.roll(obs, it, axis=1) if it else obs
        outs = []
        for core, m in cores:
            with core.on_stream():
                core.obs.copy_(torch.as_tensor(ob).reshape(core.obs.shape))
            m.run(noise=nz, use_graph=it > 0)
            torch.cuda.synchronize()
            o = {k: v.detach().cpu().numpy().copy() for k, v in core.out.items()}
            o["log_weights"] = core.log_weights.cpu().numpy().copy()
            outs.append(o)
        for k, v in outs[0].items():
            assert np.array_equal(v, outs[1][k], equal_nan=True), (k, it)
        if prev is not None:
            assert not np.array_equal(outs[0]["log_weights"], prev), "the passes were meant to differ"
        prev = outs[0]["log_weights"]


def test_chain_gradients_match_the_launches():
    """The backward pass consumes the tape the chain's forward pass wrote -- bit-identical to the launches' tape -- so the gradients
    agree to the run-to-run reproducibility of the backward pass itself (its weight-gradient launches accumulate with float
    atomics: the same tolerance is applied between two evaluations of the launch path)."""
    B, K, N, T, hw = 8, 5, 3, 3, (50, 50)
    F, d, obs, P, noise = _inputs(B, K, N, T, hw)
    grads = []
    for chain in (False, True):
        core = SqairCore(F, hw, options={"slot_chain": 1} if chain else None)
        core.set_params(P)
        m = Model(obs, None, core, K, presence=d["nums"])
        with core.on_stream():
            core.noise.copy_(torch.as_tensor(noise).reshape(core.noise.shape))
            for use_graph in (False, True):
                g = core.grad_step(use_graph=use_graph).clone()
                core.stream.synchronize()
                grads.append(g.cpu().numpy())
        if chain:
            core.check_chain(train=True)
    assert np.isfinite(grads[0]).all() and np.abs(grads[0]).max() > 0
    gmax = float(np.abs(grads[0]).max())
    for g in grads[1:]:
        assert float(np.abs(grads[0] - g).max()) <= 2e-6 * gmax


def test_chain_keeps_out_of_configurations_it_does_not_serve():
    """LSTM cells, more than 320 particle rows: the option is accepted, the pass runs one launch per op (and stays correct)."""
    B, K, N, T, hw = 3, 2, 2, 2, (50, 50)
    F, d, obs, P, noise = _inputs(B, K, N, T, hw, time_transition="LSTM")
    _, _, ref = _run(F, hw, d, obs, P, noise, K, chain=False, use_graph=True)
    core, _, got = _run(F, hw, d, obs, P, noise, K, chain=True, use_graph=True)
    for k, v in ref.items():
        assert np.array_equal(v, got[k], equal_nan=True), k
    plain = SqairCore(F, hw)
    plain.set_params(P)
    Model(obs, None, plain, K, presence=d["nums"]).run(noise=noise, use_graph=True)
    assert core.lib.sqair_graph_nodes(core.handle) == plain.lib.sqair_graph_nodes(plain.handle)


def test_chain_tables_survive_fresh_operand_buffers_and_a_capture():
    """A chain launch reads a table holding operand addresses; tables are cached in device arenas of the handle.  A caller that
    streams batches through FRESH frame / noise buffers makes new tables every pass: with a small arena the passes below fill it
    many times over (recycled after a synchronise), a capture in between pins the arena its graph refers to (the later passes
    open another one), and every pass -- eager on fresh buffers, and the replay of the pinned graph at the end -- stays
    bit-identical to the launch-per-op path."""
    B, K, N, T, hw = 4, 2, 3, 3, (50, 50)
    F, d, obs, P, noise = _inputs(B, K, N, T, hw)
    _, _, ref = _run(F, hw, d, obs, P, noise, K, chain=False, use_graph=False)
    core = SqairCore(F, hw, options={"slot_chain": 1, "slot_chain_arena_kb": 256})
    core.set_params(P)
    m = Model(obs, None, core, K, presence=d["nums"], debug=True)
    keep = []   # old buffers stay alive: every pass gets addresses no earlier pass had

    def fresh():
        keep.append((core.obs, core.noise))
        core.obs, core.noise = core.obs.clone(), torch.empty_like(core.noise)

    def check(use_graph):
        m.run(noise=noise, use_graph=use_graph)
        torch.cuda.synchronize()
        for k, v in core.out.items():
            assert np.array_equal(ref[k], v.detach().cpu().numpy(), equal_nan=True), (k, use_graph)
        assert np.array_equal(ref["log_weights"], core.log_weights.cpu().numpy())

    for _ in range(8):
        fresh()
        check(False)
    check(True)            # capture on the current buffers: their tables are pinned
    pinned = (core.obs, core.noise)
    for _ in range(8):
        fresh()
        check(False)
    core.obs, core.noise = pinned
    check(True)            # the graph's tables are still where it expects them
    assert core.lib.sqair_set_option(core.handle, b"slot_chain_arena_kb", 512) == -2   # only before the first pass
