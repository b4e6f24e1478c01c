"""Helpers shared by CPU and GPU tests that need no HIP device (fixture loading)."""
import hashlib
import os

import numpy as np

from sqair_amd.params import flatten_params, init_params, param_spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_params(z, F, hw):
    """The float32 parameters of a golden fixture, regenerated from its (seed, jitter, mean image) and checked against the
    stored sha256 of the flat buffer: a change of the initialisers must fail here, not as a mysterious parity error."""
    pseed = int(z["meta"][6])
    P = init_params(F, hw, seed=pseed, mean_img=z["mean_img"], jitter=float(z["jitter"]))
    P = {k: np.asarray(v, dtype=np.float32) for k, v in P.items()}
    got = hashlib.sha256(flatten_params(P, param_spec(F, hw)).tobytes()).hexdigest()
    assert got == str(z["params_sha256"]), "fixture parameters changed: sha256 {} != stored {}".format(got, z["params_sha256"])
    return P


def draw_noise(rng, T, R, N, nzw):
    """eps ~ N(0, 1) for the Normals, u ~ U[0, 1) in the last entry of every slot: the noise tensor of one pass [T, R, 2, N, nzw]."""
    nz = rng.standard_normal((T, R, 2, N, nzw)).astype(np.float32)
    nz[..., -1] = rng.uniform(size=nz.shape[:-1]).astype(np.float32)
    return nz
