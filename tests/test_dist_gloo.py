"""World-size-2 gloo test (CPU) of the data-parallel path: sharding by sequences, identical noise rows,
one scalar all-reduce for the metrics, one flat-gradient all-reduce for training.  The per-rank compute is
the CPU oracle here (the HIP library needs a GPU); what is under test is the N>1 logic of sqair_amd.dist."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sqair_oracle as O
from sqair_amd import dist as sqdist
from sqair_amd.flags import make_flags
from sqair_amd.params import flatten_params, init_params, param_spec


def _problem():
    F = make_flags(k_particles=2, n_steps_per_image=2)
    hw = (16, 16)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, jitter=0.05).items()}
    rng = np.random.default_rng(0)
    T, B = 3, 4
    obs = rng.uniform(size=(T, B) + hw).astype(np.float32)
    nz = rng.standard_normal((T, B * 2, 2, 2, 55)).astype(np.float32)
    nz[..., -1] = rng.uniform(size=nz.shape[:-1])
    return F, hw, P, obs, nz


def _flat_grad(orc, spec):
    g = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in orc.P.items()}
    return torch.tensor(flatten_params(g, spec).astype(np.float64))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    F, hw, P, obs, nz = _problem()
    spec = param_spec(F, hw)
    obs_r = sqdist.shard_batch(obs, rank, world)
    nz_r = sqdist.shard_noise(nz, 2, rank, world)
    orc = O.SqairOracle(P, O.make_cfg(F, hw), torch.float64, requires_grad=True)
    m = orc.model(obs_r, nz_r)
    orc.make_target(m).backward()
    g = _flat_grad(orc, spec)
    g = g * sqdist.allreduce_flat_grads(g)   # in-place sum; the returned 1 / world goes into the optimiser's grad_scale
    sc = sqdist.reduce_scalars(torch.tensor([float(m.elbo_iwae), float(m.elbo_vae)], dtype=torch.float64))
    if rank == 0:
        np.savez(out, grad=g.numpy(), scalars=sc.numpy(), lw=m.log_weights.detach().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process(tmp_path):
    out = str(tmp_path / "r0.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    F, hw, P, obs, nz = _problem()
    spec = param_spec(F, hw)
    orc = O.SqairOracle(P, O.make_cfg(F, hw), torch.float64, requires_grad=True)
    m = orc.model(obs, nz)
    orc.make_target(m).backward()
    g = _flat_grad(orc, spec).numpy()
    # per-row results are shard-invariant, metrics and gradients equal the global-batch ones
    assert np.allclose(z["lw"], m.log_weights.detach().numpy()[:2], rtol=0, atol=1e-9)
    assert np.allclose(z["scalars"], [float(m.elbo_iwae), float(m.elbo_vae)], rtol=1e-12)
    assert np.allclose(z["grad"], g, rtol=1e-5, atol=1e-6 * np.abs(g).max())  # the flat gradient buffer is fp32


def test_shard_helpers():
    x = np.arange(2 * 8 * 3).reshape(2, 8, 3)
    assert sqdist.shard_batch(x, 1, 4).shape == (2, 2, 3)
    assert np.array_equal(sqdist.shard_batch(x, 3, 4), x[:, 6:8])
    nz = np.arange(2 * 40).reshape(2, 40, 1, 1, 1)
    assert np.array_equal(sqdist.shard_noise(nz, 5, 1, 4)[0, :, 0, 0, 0], np.arange(10, 20))
    try:
        sqdist.shard_bounds(10, 0, 4)
        assert False
    except ValueError:
        pass
