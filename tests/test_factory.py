"""The drop-in factory itself: `sqair_amd.model.load(img, coords, num, mean_img, debug)` reading the GLOBAL flags, exactly as the
reference's driver reaches its model (reference: sqair/experiment_tools.py:147-157 loads configs/mlp_mnist_model.py and calls
`load(img, coords, num, mean_img, debug)` (:74-150) with the flags parsed at import time; sqair/scripts/experiment.py:118-147
then does `model.make_target(opt)` and `opt.apply_gradients(gvs)`).  Every other test builds `Model(obs, None, core, K, ...)`
by hand; this one goes through the front door."""
import copy

import numpy as np
import pytest
import torch

from oracle import sqair_oracle as O
from sqair_amd import flags as FL
from sqair_amd import model as MD
from sqair_amd.data import make_sequences, to_float
from sqair_amd.params import init_params
from sqair_amd.train import Optimizer, learning_rate, rmsprop_reference
from tests.hip_util import stable_noise


@pytest.fixture
def global_flags():
    """The reference's flags are process-global (tf.flags); tests restore them."""
    saved = copy.deepcopy(FL.FLAGS.__dict__)
    yield FL.FLAGS
    FL.FLAGS.__dict__.clear()
    FL.FLAGS.__dict__.update(saved)


def test_load_error_paths_mirror_the_reference(global_flags):
    """Invalid flag values fail in the factory with the reference's exceptions, before any device work
    (reference: sqair/propagate.py:42-43 `raise ValueError('Invalid prior type...')`, sqair/sqair_modules.py:224)."""
    img = np.zeros((2, 2, 50, 50, 1), np.float32)
    global_flags.update(prop_prior_type="bogus")
    with pytest.raises(ValueError, match="Invalid prior type"):
        MD.make_config(global_flags, (50, 50))
    global_flags.update(prop_prior_type="rnn", disc_prior_type="poisson")
    with pytest.raises(ValueError, match="Invalid prior type"):
        MD.make_config(global_flags, (50, 50))
    global_flags.update(disc_prior_type="cat", scale_prior="1,2,3")
    with pytest.raises(ValueError, match="Incorrect number of elements"):   # configs/mlp_mnist_model.py:67-68
        MD.make_config(global_flags, (50, 50))
    with pytest.raises(ValueError, match="unknown flag"):
        global_flags.update(no_such_flag=1)
    if torch.cuda.is_available():
        global_flags.update(scale_prior="-2", prop_prior_type="bogus")
        with pytest.raises(ValueError, match="Invalid prior type"):
            MD.load(img, None, None, None)


@pytest.mark.gpu
def test_load_through_global_flags_then_one_training_step_matches_the_oracle(global_flags):
    F = global_flags
    F.update(k_particles=3, n_steps_per_image=3, learning_rate=1e-3, train_itr=100, opt="rmsprop")
    T, B, hw = 3, 3, (50, 50)
    K, N = 3, 3
    d = make_sequences(B, T=T, canvas=hw, n_objects=(1, 2), obj_size=24, seed=21)
    img = to_float(d["imgs"])[..., None]                       # 5-D [T, B, H, W, 1] as the reference's pipeline delivers it
    mean_img = img.mean((0, 1))[..., 0]
    m = MD.load(img, d["coords"], d["nums"], mean_img, debug=True)          # F defaults to the global FLAGS
    assert m.core.F is FL.FLAGS and m.k_particles == K and m.core.N == N
    assert (m.n_timesteps, m.batch_size, m.tiled_batch_size) == (T, B, B * K) and m.img_size == list(hw)
    # the factory's own initial parameters (seed 0, the reference's initialisers with the data mean as dec.mean_img)
    P = {k: np.asarray(v, np.float32) for k, v in init_params(F, hw, seed=0, mean_img=mean_img).items()}
    got = m.core.get_params()
    assert all(np.array_equal(got[k], P[k].reshape(got[k].shape)) for k in P)
    obs = img[..., 0]
    noise, ref, orc, _ = stable_noise(F, hw, P, obs, T, B * K, N, seed0=300, nums=d["nums"], requires_grad=True)
    m.run(noise=noise)
    assert np.array_equal(m.presence.cpu().numpy(), ref.presence.detach().numpy())
    for k in ("elbo_iwae", "elbo_vae", "data_ll", "kl", "num_steps", "num_step_accuracy"):
        a, b = float(getattr(m, k)), float(getattr(ref, k).detach())
        assert abs(a - b) <= 1e-4 * max(abs(b), 1.0), (k, a, b)
    # experiment.py:140-147: opt = RMSProp(lr, momentum=.9); target, gvs = model.make_target(opt); opt.apply_gradients(gvs)
    opt = Optimizer(m.core, F.opt)
    target, gvs = m.make_target(opt)
    want_target = orc.make_target(ref)
    assert abs(float(target) - float(want_target.detach())) <= 1e-3 * abs(float(want_target.detach()))
    want_target.backward()
    assert len(gvs) == len(m.core.spec) and all(isinstance(n, str) for _, n in gvs)
    before = {k: v.copy() for k, v in m.core.get_params().items()}
    opt.apply_gradients(gvs)
    m.core.stream.synchronize()
    torch.cuda.synchronize()
    after = m.core.get_params()
    lr = learning_rate(F, 0)
    checked = 0
    for name in ("dec.l2.b", "disc.steps.l1.b", "enc.glimpse.l1.w", "prop.prior_linear.w", "seq.latent_enc.l0.w"):
        g = orc.P[name].grad
        assert g is not None, name
        g = g.numpy().reshape(before[name].shape)
        want, _, _ = rmsprop_reference(before[name].astype(np.float64), g, np.ones_like(g), np.zeros_like(g), lr)
        delta_w, delta_g = want - before[name], after[name].astype(np.float64) - before[name]
        assert np.abs(delta_w).max() > 0, name
        assert np.abs(delta_g - delta_w).max() <= 2e-3 * np.abs(delta_w).max(), (name, np.abs(delta_g - delta_w).max(), np.abs(delta_w).max())
        checked += 1
    assert checked == 5
    # the next pass runs on the updated (re-packed) parameters
    m.run(noise=noise)
    assert np.isfinite(float(m.elbo_iwae)) and float(m.elbo_iwae) != float(ref.elbo_iwae.detach())


@pytest.mark.gpu
def test_make_target_with_debug_checks_the_training_pass_and_takes_the_iwae_alias(global_flags):
    """`make_target(opt)` with debug=True and NO earlier `run()`: the debug checks must look at the workspace the gradient pass
    wrote (the tape), not at the never-written inference workspace (whose zero scales used to raise a false 'scale not positive').
    `vi_target='iwae'` is the reference's name for the VIMCO target (Model.VI_TARGETS): an alias, not a change of target -- it
    must neither drop the captured graphs on every call nor slip past the one-particle guard."""
    F = global_flags
    F.update(k_particles=2, n_steps_per_image=2, learning_rate=1e-4, train_itr=100, opt="rmsprop")
    T, B, hw = 2, 3, (50, 50)
    d = make_sequences(B, T=T, canvas=hw, n_objects=(1, 2), obj_size=24, seed=5)
    img = to_float(d["imgs"])
    for chain in (0, 1):
        if not chain:
            m = MD.load(img, d["coords"], d["nums"], img.mean((0, 1)), debug=True)
        else:   # (the in-launch slot chain on: its status words live in the training workspace too)
            core = MD.SqairCore(F, hw, options={"slot_chain": 1})
            core.set_params(init_params(F, hw, seed=0, mean_img=img.mean((0, 1))))
            m = MD.Model(img, d["coords"], core, 2, presence=d["nums"], debug=True)
        opt = Optimizer(m.core, F.opt)
        m._use_graph = True
        target, gvs = m.make_target(opt)            # first pass of this model: the training pass, debug checks on the tape
        assert np.isfinite(float(target)) and len(gvs) == len(m.core.spec)
        assert m.core._train_graph_ready
        t2, _ = m.make_target(opt, vi_target="iwae")
        assert getattr(m.core, "vi_target", "vimco") == "vimco" and m.core._train_graph_ready, "the alias must not drop the captured graph"
        assert np.isfinite(float(t2))
    with pytest.raises(ValueError, match="vi_target"):
        m.make_target(opt, vi_target="elbo")
    F.update(k_particles=1)
    m1 = MD.load(img, d["coords"], d["nums"], img.mean((0, 1)))
    with pytest.raises(ValueError, match="k_particles >= 2"):
        m1.make_target(None, vi_target="iwae")
