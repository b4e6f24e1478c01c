"""CPU: the oracle against the committed golden vectors (tests/golden/*.npz, generator tests/golden/make_golden.py).

The fixtures were produced BY the fp64 oracle (the reference cannot run here), so these tests pin the oracle
against drift — a change of the restatement that moves any of the 38 outputs, the objective or a gradient shows up here on the
CPU, before the GPU parity tests compare the HIP path with the same files — and guard the fixtures' inputs: the parameters are
regenerated from (seed, jitter) and must hash to the stored `params_sha256`."""
import os

import numpy as np
import pytest
import torch

from oracle import sqair_oracle as O
from sqair_amd.flags import make_flags
from tests.hip_util_cpu import GOLDEN, fixture_params


@pytest.mark.parametrize("name", ["cfg1_plumbing", "hw128_small"])
def test_oracle_reproduces_the_forward_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    T, B, K, N, H, W, _, _ = [int(v) for v in z["meta"]]
    F = make_flags(k_particles=K, n_steps_per_image=N)
    P = fixture_params(z, F, (H, W))
    orc = O.SqairOracle(P, O.make_cfg(F, (H, W)), torch.float64)
    with torch.no_grad():
        m = orc.model(z["obs"], z["noise"], num=z["nums"], resample_u=z["resample_u"])
    for k in [f for f in z.files if f.startswith("out_")]:
        name_o = k[4:]
        got = m.outputs["_" + name_o if name_o.startswith("final_") else name_o].numpy().reshape(z[k].shape)
        assert np.allclose(got, z[k], rtol=1e-12, atol=1e-12), k
    for k in ("elbo_iwae", "elbo_vae", "data_ll", "kl"):
        assert abs(float(getattr(m, k)) - float(z["model_" + k])) <= 1e-10 * max(1.0, abs(float(z["model_" + k]))), k
    # the fp32 mode of the same oracle (what bench.py's cpu_baseline times) agrees to the north-star tolerance
    o32 = O.SqairOracle(P, O.make_cfg(F, (H, W)), torch.float32)
    with torch.no_grad():
        m32 = o32.model(z["obs"], z["noise"], num=z["nums"], resample_u=z["resample_u"])
    if np.array_equal(m32.presence.numpy(), z["out_presence"]):
        assert abs(float(m32.elbo_iwae) - float(z["model_elbo_iwae"])) <= 1e-4 * abs(float(z["model_elbo_iwae"]))


def test_oracle_reproduces_the_gradient_fixture():
    """k5_iwae_vimco_grads.npz: per parameter, sampled elements + sum / |sum| / L2 / max of the gradient of the VIMCO target."""
    z = np.load(os.path.join(GOLDEN, "k5_iwae_vimco.npz"))
    g = np.load(os.path.join(GOLDEN, "k5_iwae_vimco_grads.npz"))
    assert str(g["params_sha256"]) == str(z["params_sha256"])
    T, B, K, N, H, W, _, _ = [int(v) for v in z["meta"]]
    F = make_flags(k_particles=K, n_steps_per_image=N)
    P = fixture_params(z, F, (H, W))
    orc = O.SqairOracle(P, O.make_cfg(F, (H, W)), torch.float64, requires_grad=True)
    m = orc.model(z["obs"], z["noise"], num=z["nums"], resample_u=z["resample_u"])
    target = orc.make_target(m)
    assert abs(float(target.detach()) - float(g["vimco_target"])) <= 1e-10 * abs(float(g["vimco_target"]))
    target.backward()
    names = [k[4:] for k in g.files if k.startswith("idx/")]
    assert len(names) == len(orc.P) and all(orc.P[n].grad is not None for n in names)   # the reference's own assert, model.py:163-166
    for n in names:
        got = orc.P[n].grad.numpy().reshape(-1)
        stat = g["stat/" + n]
        assert np.allclose(got[g["idx/" + n]], g["val/" + n], rtol=1e-9, atol=1e-12 * max(stat[3], 1e-30)), n
        assert np.allclose([got.sum(), np.abs(got).sum(), np.sqrt((got * got).sum()), np.abs(got).max()], stat, rtol=1e-9,
                           atol=1e-12 * max(stat[3], 1e-30)), n
