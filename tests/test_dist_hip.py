"""N > 1 on the HIP path: two ranks, each running the library on its own shard of sequences, one
all-reduce of the flat gradient buffer, the fused optimiser with 1 / world folded in.  With >= 2 visible devices the
ranks sit on different GPUs and the collective is ``ncclAllReduce`` (RCCL C API) enqueued on the library's launch stream;
on the 1-GPU box both ranks share device 0 and the process group is gloo (RCCL refuses two ranks on one device) — the
per-rank compute is the HIP path either way.  Also: the native communicator on real hardware with one rank, and
``python bench.py --gpus 2`` without a launcher."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from sqair_amd.flags import make_flags

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from sqair_amd.data import make_sequences, to_float
from sqair_amd.dist import shard_batch
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params
from sqair_amd.train import Trainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n_dev = torch.cuda.device_count()
backend = "nccl" if n_dev >= world else "gloo"
dev = "cuda:%d" % (rank % n_dev)
torch.cuda.set_device(rank % n_dev)
dist.init_process_group(backend, **(dict(device_id=torch.device(dev)) if backend == "nccl" else {{}}))
comm = None
if backend == "nccl":
    from sqair_amd.rccl import RcclComm
    comm = RcclComm.from_process_group(dev)
K, N, T, B, hw = 3, 3, 3, 4, (50, 50)
F = make_flags(k_particles=K, n_steps_per_image=N, learning_rate=1e-3)
obs = to_float(make_sequences(B, T=T, canvas=hw, seed=11)["imgs"])
P = {{k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=2, mean_img=obs.mean((0, 1)), jitter=0.05).items()}}
core = SqairCore(F, hw, device=dev)
core.set_params(P)
mine = shard_batch(obs, rank, world)
tr = Trainer(Model(mine, None, core, K, outputs="minimal"), F, comm=comm)
per = B // world
grads = []
for it in range(2):
    g = tr.step(seed=77, global_batch=B, b0=rank * per)
    core.stream.synchronize()
    grads.append(g.cpu().numpy().copy())       # the all-reduced SUM of the shard gradients
if rank == 0:
    np.savez({out!r}, g0=grads[0], g1=grads[1], flat=core.flat.cpu().numpy(), backend=backend,
             rccl_ranks=(comm.n_ranks if comm is not None else 0))
dist.barrier()
if comm is not None:
    comm.destroy()
dist.destroy_process_group()
"""


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_rank_hip_training_steps_equal_one_rank_on_the_global_batch(tmp_path):
    from sqair_amd.data import make_sequences, to_float
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.params import init_params
    from sqair_amd.train import Trainer
    out = str(tmp_path / "r0.npz")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=env, timeout=600)
    assert rc == 0
    z = np.load(out)
    if torch.cuda.device_count() >= 2:
        assert str(z["backend"]) == "nccl" and int(z["rccl_ranks"]) == 2
    # the same two steps in ONE process on the global batch (Philox noise is keyed by the position in the global batch)
    K, N, T, B, hw = 3, 3, 3, 4, (50, 50)
    F = make_flags(k_particles=K, n_steps_per_image=N, learning_rate=1e-3)
    obs = to_float(make_sequences(B, T=T, canvas=hw, seed=11)["imgs"])
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=2, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
    core = SqairCore(F, hw)
    core.set_params(P)
    tr = Trainer(Model(obs, None, core, K, outputs="minimal"), F)
    for it in range(2):
        g = tr.step(seed=77, global_batch=B, b0=0)
        core.stream.synchronize()
        want = g.cpu().numpy()
        got = 0.5 * z["g%d" % it]                # sum over 2 ranks of shard means / world = global mean
        scale = np.abs(want).max()
        assert np.isfinite(got).all() and scale > 0
        assert np.abs(got - want).max() <= 2e-4 * scale, (it, np.abs(got - want).max() / scale)
    delta = np.abs(core.flat.cpu().numpy() - np.concatenate([np.asarray(v).reshape(-1) for v in P.values()])).max()
    assert delta > 0
    assert np.abs(z["flat"] - core.flat.cpu().numpy()).max() <= 2e-3 * delta   # identical updates on every rank


def test_two_ranks_on_two_devices_all_reduce_over_rccl(tmp_path):
    """The path the driver's 8-GPU run takes, at world size 2: one rank per DEVICE, `ncclCommInitRank` with nranks > 1, the
    gradient all-reduce enqueued on the library's launch stream.  Needs two visible devices; the 1-GPU boxes of this
    environment cannot run it (RCCL refuses two ranks on one device), which is reported as an expected failure, not a pass."""
    if torch.cuda.device_count() < 2:
        pytest.xfail("needs >= 2 visible HIP devices for a 2-rank RCCL communicator (this box has {}): ncclCommInitRank with "
                     "nranks > 1 stays unexecuted here; the 2-rank step itself is covered with gloo above".format(
                         torch.cuda.device_count()))
    out = str(tmp_path / "r0.npz")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=env, timeout=600)
    assert rc == 0
    z = np.load(out)
    assert str(z["backend"]) == "nccl" and int(z["rccl_ranks"]) == 2
    assert np.isfinite(z["g0"]).all() and np.isfinite(z["flat"]).all()


def _bench_with_broken_rccl(extra):
    code = ("import sys, runpy; sys.argv = ['bench.py', '--force-dist', '--steps', '1', '--warmup', '1', '--train-steps', '1', "
            "'--no-cpu-baseline', '--no-timeline', '--streams', '0', '--cfg', '1'] + {extra!r}\n"
            "sys.path.insert(0, {root!r})\n"
            "import sqair_amd.rccl as R\n"
            "def broken(*a, **k): raise OSError('librccl.so deliberately unavailable')\n"
            "R.RcclComm.from_process_group = classmethod(broken)\n"
            "runpy.run_path({bench!r}, run_name='__main__')\n").format(root=ROOT, bench=os.path.join(ROOT, "bench.py"), extra=extra)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)


def test_bench_never_falls_back_silently_when_the_native_communicator_fails(tmp_path):
    """With one device per rank the native communicator (RCCL's C API on the launch stream) is what the step's all-reduce uses.
    If it cannot be built the run says so -- on stderr and in the JSON line -- and goes through torch.distributed's nccl group;
    with `--require-native-comm` it exits with status 3 and prints no bench line.  Simulated at one rank with `--force-dist`
    and an unloadable RCCL (the ctypes loader is broken through the module attribute, not through the environment)."""
    p = _bench_with_broken_rccl(["--require-native-comm"])
    assert p.returncode == 3, (p.returncode, p.stderr[-1500:])
    assert "refusing to fall back" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    p = _bench_with_broken_rccl([])
    assert p.returncode == 0, (p.returncode, p.stderr[-1500:])
    assert "could not be built" in p.stderr and "torch.distributed" in p.stderr
    import json
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert "deliberately unavailable" in line["native_comm_error"]
    assert line["train"]["native_comm_error"] == line["native_comm_error"] and line["dist_backend"] == "nccl"
    # which communicator carried the all-reduce is said in one field (and rccl_ranks counts THAT communicator's ranks)
    assert line["collective_path"] == "torch-nccl" and line["train"]["collective_path"] == "torch-nccl" and line["rccl_ranks"] == 1
    # ... and the healthy branch: the native communicator on the launch stream
    code_ok = ("import sys, runpy; sys.argv = ['bench.py', '--force-dist', '--steps', '1', '--warmup', '1', '--train-steps', '1', "
               "'--no-cpu-baseline', '--no-timeline', '--streams', '0', '--cfg', '1']\n"
               "runpy.run_path({bench!r}, run_name='__main__')\n").format(bench=os.path.join(ROOT, "bench.py"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code_ok], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr[-1500:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["collective_path"] == "native" and line["train"]["collective_path"] == "native" and line["native_comm_error"] is None
    assert line["rccl_ranks"] == 1


def test_native_rccl_all_reduce_on_the_launch_stream_single_rank():
    """ncclAllReduce through the C API, enqueued on the core's own stream between a gradient-graph replay and the fused
    optimiser (one rank here: the sum is the identity; what is checked is that the communicator comes up on the hardware
    and composes with the one-stream discipline)."""
    from sqair_amd.data import make_sequences, to_float
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.rccl import RcclComm
    from sqair_amd.train import Trainer
    from tests.hip_util import params32
    K, N, T, B, hw = 2, 2, 2, 2, (50, 50)
    F = make_flags(k_particles=K, n_steps_per_image=N, learning_rate=1e-3)
    obs = to_float(make_sequences(B, T=T, canvas=hw, seed=2)["imgs"])
    core = SqairCore(F, hw)
    core.set_params(params32(F, hw, 4, 0.05, obs.mean((0, 1))))
    comm = RcclComm.from_process_group("cuda:0")
    assert comm.n_ranks == 1 and comm.version() > 20000
    try:
        with core.on_stream():
            Model(obs, None, core, K, outputs="minimal")
            core.draw_noise(seed=3, step=0)
            before = core.grad_step(use_graph=True).clone()
            comm.all_reduce_sum_(core.flat_grad, core.stream)
            after = core.flat_grad.clone()
        core.stream.synchronize()
        assert torch.equal(after, before)
        tr = Trainer(Model(obs, None, core, K, outputs="minimal"), F, comm=comm)
        p0 = core.flat.clone()
        tr.step(seed=3)
        core.stream.synchronize()
        assert torch.isfinite(core.flat).all() and float((core.flat - p0).abs().max()) > 0
    finally:
        comm.destroy()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment: bench.py re-executes itself under
    torch.distributed.run, rank 0 prints the JSON line last."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--train-steps", "2", "--no-cpu-baseline", "--cfg", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and np.isfinite(line["value"]) and line["value"] > 0
    assert line["config"]["parallelism"] == "dp2" and np.isfinite(line["train"]["value"])
    sg = line["single_gpu_at_global_batch"]   # the strong-scaling reference: rank 0 alone on the global batch, outside `value`
    assert sg["sequences"] == line["config"]["global_batch"] and sg["forward_value"] > 0 and sg["train_value"] > 0
    assert line["train"]["allreduce_ms"] > 0
    if torch.cuda.device_count() >= 2:
        assert line["rccl_ranks"] == 2 and line["train"]["rccl_ranks"] == 2 and line["train"]["allreduce_on_launch_stream"] is True
        assert line["collective_path"] == "native"
    else:
        assert line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0 and line["train"]["allreduce_on_launch_stream"] is False
        assert line["collective_path"] == "gloo" and line["train"]["collective_path"] == "gloo"


def test_bench_eight_ranks_functional_check_on_one_node():
    """The driver's 8-GPU command line, runnable on whatever the box has: `bench.py --gpus 8` launching its own eight ranks.  On a
    1-GPU box the ranks share the device and the process group is gloo (RCCL refuses two ranks on one device) -- a FUNCTIONAL
    check of everything around the collective that world size 2 cannot reach: rank-indexed data seeding and Philox offsets for
    ranks 2..7, eight rendezvous clients on one port, the barrier order of the extra legs (training, expected-step, one GPU at the
    global batch), eight contexts' worth of memory.  No scaling number is taken from it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # (cfg-2's model at 3 sequences per rank: cfg-1 has K = 1, whose VIMCO target is NaN by the reference's own formula)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--cfg", "2", "--batch", "3", "--steps", "2", "--warmup", "1",
           "--train-steps", "2", "--no-cpu-baseline", "--no-timeline"]
    if torch.cuda.device_count() < 8:
        cmd += ["--dist-backend", "gloo"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak"
    B = line["config"]["global_batch"] // 8
    assert line["config"]["global_batch"] == 8 * B and B >= 1
    assert np.isfinite(line["value"]) and line["value"] > 0 and np.isfinite(line["elbo_iwae_nats_per_seq"])
    tr = line["train"]
    assert np.isfinite(tr["value"]) and tr["value"] > 0 and "finite=True" in tr["what"]
    assert tr["allreduce_ms"] > 0 and tr["single_rank_ms_per_step_no_collective"] > 0
    assert abs(tr["expected_ms"] - (tr["single_rank_ms_per_step_no_collective"] + tr["allreduce_ms"])) < 1e-9
    assert tr["measured_over_expected"] > 0
    sg = line["single_gpu_at_global_batch"]
    assert sg["sequences"] == 8 * B and sg["forward_value"] > 0 and sg["train_value"] > 0
    assert line["forward_all_outputs_ms"] > 0
    if torch.cuda.device_count() >= 8:
        assert line["rccl_ranks"] == 8 and tr["allreduce_on_launch_stream"] is True and line["collective_path"] == "native"
    else:
        assert line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0 and tr["allreduce_on_launch_stream"] is False
        assert line["collective_path"] == "gloo"
