"""Helpers shared by the -m gpu parity tests (all compute goes through the C-ABI library)."""
import ctypes as C
import os

import numpy as np
import torch

from oracle import sqair_oracle as O
from sqair_amd import _capi
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def params32(F, hw, seed, jitter, mean_img=None):
    P = init_params(F, hw, seed=seed, mean_img=mean_img, jitter=jitter)
    return {k: np.asarray(v, dtype=np.float32) for k, v in P.items()}


def draw_noise(rng, T, R, N, nzw):
    nz = rng.standard_normal((T, R, 2, N, nzw)).astype(np.float32)
    nz[..., -1] = rng.uniform(size=nz.shape[:-1]).astype(np.float32)
    return nz


def run_hip(F, hw, P, obs, noise, nums=None, resample_u=None, use_graph=False, outputs="all"):
    core = SqairCore(F, hw)
    core.set_params(P)
    m = Model(obs, None, core, int(F.k_particles), presence=nums, outputs=outputs)
    m.run(noise=noise, resample_u=resample_u, use_graph=use_graph)
    torch.cuda.synchronize()
    return m


def run_oracle(F, hw, P, obs, noise, nums=None, resample_u=None, dtype=torch.float64):
    orc = O.SqairOracle(P, O.make_cfg(F, hw), dtype)
    with torch.no_grad():
        m = orc.model(obs, noise, num=nums, resample_u=resample_u)
        m.vimco_target = orc.make_target(m)
    return m


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
