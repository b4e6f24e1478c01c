"""n_hidden values the row kernels are not written for (anything but a multiple of 128) run on layers PADDED with inert hidden
units (csrc/sqair_internal.h: SqairHandle): zero weights and biases in, zero weights out, zero initial states.  CPU checks of the
two things that rests on: the library's map from the caller's flat parameter buffer (reference shapes,
notebooks/play.ipynb:239-362) into the padded one puts every element where the padded model expects it, and the padded model IS
the original one -- the oracle evaluated on the padded parameters with the padded width reproduces the original outputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sqair_oracle as O
from sqair_amd import _capi
from sqair_amd.data import make_sequences, to_float
from sqair_amd.flags import make_flags
from sqair_amd.model import make_config
from sqair_amd.params import flatten_params, init_params, param_offsets, param_spec, unflatten_params
from tests.hip_util_cpu import draw_noise


@pytest.mark.parametrize("n_units,padded_units,cells", [
    (2, 4, {}), (5, 8, dict(time_transition="LSTM", prior_transition="LSTM", transition="LSTM")),
    (3, 4, dict(transition="GRU", time_transition="VanillaRNN", prior_transition="VanillaRNN")), (7, 8, dict(prior_transition="LSTM"))])
def test_padded_parameters_reproduce_the_unpadded_model(n_units, padded_units, cells):
    lib = _capi.lib()
    hw, T, B, K, N = (32, 40), 2, 2, 2, 3
    F = make_flags(k_particles=K, n_steps_per_image=N, n_units=n_units, **cells)
    Fp = make_flags(k_particles=K, n_steps_per_image=N, n_units=padded_units, **cells)
    cfg = make_config(F, hw)
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    try:
        spec, specp = param_spec(F, hw), param_spec(Fp, hw)
        total, totalp = param_offsets(spec)[1], param_offsets(specp)[1]
        assert lib.sqair_param_count(h) == total
        u2i = np.zeros(total, np.int32)
        # the padded inventory has exactly the shapes of the model with the padded width
        assert lib.sqair_debug_padded_count(h, u2i.ctypes.data_as(C.POINTER(C.c_int))) == totalp
        assert len(np.unique(u2i)) == total and u2i.min() >= 0 and u2i.max() < totalp
        back = _capi.SqairConfig()
        assert lib.sqair_get_config(h, C.byref(back)) == 0 and back.n_hidden == 32 * n_units   # the caller's configuration
    finally:
        lib.sqair_destroy(h)
    d = make_sequences(B, T=T, canvas=hw, n_objects=(1, 2), obj_size=20, seed=9)
    obs = to_float(d["imgs"])
    P = {k: np.asarray(v, np.float32) for k, v in init_params(F, hw, seed=5, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
    fp = np.zeros(totalp, np.float32)
    fp[u2i] = flatten_params(P, spec)
    Pp = unflatten_params(fp, specp)
    noise = draw_noise(np.random.default_rng(0), T, B * K, N, 55)
    with torch.no_grad():
        a = O.SqairOracle(P, O.make_cfg(F, hw), torch.float64).model(obs, noise)
        b = O.SqairOracle(Pp, O.make_cfg(Fp, hw), torch.float64).model(obs, noise)
    assert float(a.prop_pres.sum()) > 0
    for k, v in a.outputs.items():
        w = b.outputs[k].numpy()
        if k in ("_final_temporal_state", "_final_prior_state"):   # recurrent states: [hidden | cell] halves, each padded on its own
            nh, nhp = 32 * n_units, 32 * padded_units
            w = np.concatenate([w[..., i * nhp:i * nhp + nh] for i in range(w.shape[-1] // nhp)], -1)
        assert v.shape == w.shape, k
        assert np.abs(v.numpy() - w).max() <= 1e-11 * max(1.0, np.abs(w).max()), k
    assert abs(float(a.elbo_iwae) - float(b.elbo_iwae)) <= 1e-10 * abs(float(a.elbo_iwae))
