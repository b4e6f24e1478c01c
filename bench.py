#!/usr/bin/env python
"""bench.py — frames/sec of the SQAIR Discover/Propagate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input: noise draw on device, the
T-frame forward unroll (HIP-graph replay of the launch sequence), IWAE / VIMCO reductions.  Inputs
(frames, parameters) are resident in HBM before the timed region starts.  Workload at every N:
BASELINE.json configs[1] per GPU — multi-MNIST-like 2-glyph sequences, seq_len 10, 50x50, batch 32,
K=5 IWAE particles, 4 object slots ("cfg2"); with N > 1 every rank processes its own 32-sequence
shard (weak scaling, the sharding of configs[2]; no collective on the data path of the forward pass).
frames/step = B * T per rank (all K particles of a frame count as one frame).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — dominant kernel family k_linear (fp32 MFMA dense layers): achieved = algorithmic FLOPs per
                 launch (135.4 MFLOP/frame at cfg-2, as the reference graph computes
                 them, / dense launches per step) / average launch duration, measured live with one
                 HIP-event pair around replays of a graph of the pass's dense launches (boundary
                 included); frac_device_clock (in-kernel only) and frac_rocprof (profiles/, recomputable
                 with tools/roofline_from_rocprof.py) are reported beside it; peak = 157.3 TFLOP/s.
  cpu_baseline — the oracle (PyTorch-CPU fp32 restatement at the reference's op granularity,
                 kind "port": the TF1 reference cannot run here) timed on the host cores on a bounded
                 sample of the same workload; only this leg imports oracle/.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense f32-in matrix peak (= vector peak)
FLOP_PER_FRAME_CFG2 = 2 * 5 * 13543744  # 2 * K * MACs per frame-particle (SURVEY.md 8(d), Appendix D, N=4)


def cpu_baseline(F, hw, P, obs, noise, hip_ref, budget_s=25.0):
    """Times the oracle (fp32 PyTorch-CPU at the reference's op granularity) on the SAME batch, parameters and noise.
    The ops are tiny (M = B*K = 160-row GEMMs), so more threads is not faster: the thread count is calibrated on a
    2-frame slice among a few candidates and `cores` reports the one actually used."""
    from oracle import sqair_oracle as O
    T, B = obs.shape[:2]
    ncpu = os.cpu_count() or 1
    orc = O.SqairOracle(P, O.make_cfg(F, hw), torch.float32)
    best_thr, best_t = 1, float("inf")
    with torch.no_grad():
        for thr in [t for t in (1, 4, 8, 16, 32, 64) if t <= ncpu]:
            torch.set_num_threads(thr)
            orc.model(obs[:2], noise[:2])
            t0 = time.perf_counter()
            orc.model(obs[:2], noise[:2])
            dt = time.perf_counter() - t0
            if dt < best_t:
                best_thr, best_t = thr, dt
        torch.set_num_threads(best_thr)
        times, m = [], None
        t_start = time.perf_counter()
        for it in range(1 + 7):
            t0 = time.perf_counter()
            m = orc.model(obs, noise)
            dt = time.perf_counter() - t0
            if it > 0:
                times.append(dt)
            if time.perf_counter() - t_start > budget_s and len(times) >= 1:
                break
    # the same workload as a training step: forward + VIMCO target + autograd backward (no optimiser update)
    orc_t = O.SqairOracle(P, O.make_cfg(F, hw), torch.float32, requires_grad=True)
    ttimes = []
    for it in range(3):
        t0 = time.perf_counter()
        mt = orc_t.model(obs, noise)
        orc_t.make_target(mt).backward()
        if it > 0:
            ttimes.append(time.perf_counter() - t0)
        for v in orc_t.P.values():
            v.grad = None
    tmed = float(np.median(ttimes))
    med = float(np.median(times))
    lw_cpu = m.log_weights.numpy().astype(np.float64)
    pres_cpu = m.presence.numpy()
    same_rows = (pres_cpu == hip_ref["presence"]).all((0, 2)).reshape(lw_cpu.shape)
    rel = float(np.abs(lw_cpu - hip_ref["log_weights"])[same_rows].max() / np.abs(lw_cpu).max())
    cpu_name = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=B * T / med, unit="frames/s", cores=best_thr, kind="port",
                sample="the full step workload ({} sequences x T={} x K={}, same frames / parameters / noise as one GPU step), "
                       "fp32 PyTorch-CPU oracle at reference op granularity, median of {} passes after 1 warm-up, "
                       "{} threads (best of 1/4/8/16/32/64 on a 2-frame slice) of {} logical CPUs; host CPU: {}".format(
                           B, T, int(F.k_particles), len(times), best_thr, ncpu, cpu_name),
                ms_per_pass=med * 1e3, train_value=B * T / tmed, train_ms_per_pass=tmed * 1e3,
                parity=dict(rows_with_identical_presence=float(same_rows.mean()), log_weights_max_rel_err=rel,
                            elbo_iwae_oracle=float(m.elbo_iwae), elbo_iwae_hip=hip_ref["elbo_iwae"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--cfg", type=int, default=2, help="BASELINE.json config id used as the workload (default 2)")
    ap.add_argument("--train-steps", type=int, default=-1,
                    help="steps of the extra training-step leg (default min(steps, 20); 0 = skip)")
    ap.add_argument("--streams", type=int, default=3,
                    help="extra object \"streams\": forward passes of this many INDEPENDENT batches in flight on their own "
                         "streams (0 = skip; single GPU only; never part of `value`)")
    ap.add_argument("--batch", type=int, default=0,
                    help="sequences per GPU (default: the configuration's own; other values are batch-scaling experiments, "
                         "not BASELINE's metric)")
    ap.add_argument("--time-transition", default="GRU", choices=["GRU", "LSTM"],
                    help="propagation temporal cell (the shipped config and BASELINE's metric use GRU)")
    ap.add_argument("--prior-transition", default="GRU", choices=["GRU", "LSTM"], help="propagation prior cell (shipped: GRU)")
    ap.add_argument("--transition", default="VanillaRNN", choices=["VanillaRNN", "GRU", "LSTM"],
                    help="slot RNN of both cores (shipped: VanillaRNN)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run, exactly the
        # command line the driver uses; the ranks' stdout (rank 0's JSON line last) passes through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus {} but WORLD_SIZE {} (launch one rank per GPU, or run without WORLD_SIZE)".format(args.gpus, world))
    n_dev = torch.cuda.device_count()
    # RCCL wants one device per rank.  With fewer visible devices than ranks (the 1-GPU test box) the ranks share devices
    # and the process group falls back to gloo — a functional check of the N > 1 code path, reported as such
    # ("rccl_ranks": 0); SQAIR_DIST_BACKEND overrides.
    backend = os.environ.get("SQAIR_DIST_BACKEND", "nccl" if n_dev >= world else "gloo")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("SQAIR_FORCE_DIST") == "1":  # (the env knob: single-rank RCCL group, to measure what
        import torch.distributed as dist                            #  a communicator in the process costs the graph replays)
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = dict(device_id=torch.device("cuda", local_rank)) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, **kw)  # "nccl" = RCCL on ROCm
    # the training step's gradient all-reduce goes through RCCL's C API on the library's own launch stream
    comm, comm_error = None, None
    if dist is not None and backend == "nccl":
        from sqair_amd.rccl import RcclComm
        try:
            comm = RcclComm.from_process_group("cuda:{}".format(local_rank))
        except Exception as e:  # e.g. the bundled librccl.so cannot be opened a second time through ctypes
            comm_error = "{}: {}".format(type(e).__name__, e)
        # all ranks or none: a rank without the native communicator sends everybody to the process group's all-reduce
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda:{}".format(local_rank))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.destroy()
            comm = None
        if comm is None and rank == 0:
            print("bench: native RCCL communicator unavailable ({}); gradient all-reduce through torch.distributed".format(
                comm_error or "failed on another rank"), file=sys.stderr)

    from sqair_amd.data import config_inputs
    from sqair_amd.flags import make_flags
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.params import init_params

    # every rank synthesises its own shard of the global batch (seeded by rank), weights are replicated
    ov, obs, nums, _ = config_inputs(args.cfg, B=args.batch or None)
    if rank > 0:
        from sqair_amd.data import make_sequences, to_float
        d = make_sequences(obs.shape[1], T=obs.shape[0], canvas=obs.shape[2:], n_objects=(0, nums.shape[-1] - 1),
                           obj_size=28 if obs.shape[2] <= 64 else 72, seed=1234 + args.cfg + 1000 * rank)
        obs, nums = to_float(d["imgs"]), d["nums"]
    ov.update(time_transition=args.time_transition, prior_transition=args.prior_transition, transition=args.transition)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    T, B = int(obs.shape[0]), int(obs.shape[1])
    K, N = int(F.k_particles), int(F.n_steps_per_image)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in
         init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    device = "cuda:{}".format(local_rank)
    core = SqairCore(F, hw, device=device)
    # one stream for everything (noise draw, graph replays, optimiser, reductions): see SqairCore.on_stream
    torch.cuda.set_stream(core.stream)
    core.set_params(P)
    model = Model(obs, None, core, K, presence=nums, outputs="minimal")
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + rank)
    use_graph = not args.no_graph

    step_no = [0]

    def step():
        # device-side Philox noise keyed by (seed, step, position in the GLOBAL batch): ranks draw what one GPU would draw
        model.core.draw_noise(seed=1000, step=step_no[0], global_batch=B * world, b0=rank * B)
        step_no[0] += 1
        model.core.forward(use_graph=use_graph)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        # RCCL writes its start-up banner (NCCL_DEBUG=VERSION on the GPU boxes) to the C stdout buffer at communicator
        # creation: push it out now, on every rank, so that rank 0's JSON line is the LAST thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_step = B * T * world
    value = frames_per_step / (elapsed / args.steps)

    model._collect()
    elbo = float(model.elbo_iwae)
    elbo_vae = float(model.elbo_vae)
    if dist is not None:
        e = torch.tensor([elbo, elbo_vae], dtype=torch.float64, device=device)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)  # the only collective: scalar metrics, outside the timed region
        elbo, elbo_vae = float(e[0] / world), float(e[1] / world)

    # ---- training-step leg (extra object "train"): noise draw + gradient evaluation (forward with tape, VIMCO target,
    # backward; ONE HIP-graph replay) + the step's single collective (all-reduce of the flat fp32 gradient buffer over
    # RCCL) + fused RMSProp + weight re-pack.  Same workload, same barrier / max-over-ranks timing.
    train = None
    n_train = min(args.steps, 20) if args.train_steps < 0 else args.train_steps
    if n_train > 0:
        from sqair_amd.train import Trainer
        Ftr = make_flags(**dict(ov, learning_rate=1e-5, train_itr=1000000))
        trainer = Trainer(model, Ftr, use_graph=use_graph, comm=comm)
        for _ in range(max(2, min(args.warmup, 3))):
            trainer.step(seed=2000, global_batch=B * world, b0=rank * B)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_train):
            trainer.step(seed=2000, global_batch=B * world, b0=rank * B)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        train = dict(value=frames_per_step / (el / n_train), unit="frames/s", ms_per_step=el / n_train * 1e3, steps=n_train,
                     scaling="weak", graph_nodes=getattr(core, "train_graph_nodes", None),
                     collective="all-reduce(sum) of {} fp32 gradients ({:.1f} MB) per step, {}; 1/world folded into the fused "
                                "RMSProp kernel".format(core.n_params, core.n_params * 4 / 1e6,
                                                        "ncclAllReduce (RCCL C API) on the launch stream" if comm is not None
                                                        else "torch.distributed " + backend) if world > 1 else "none (1 rank)",
                     what="draw noise + forward(train) + VIMCO target + backward (one HIP-graph replay) + all-reduce + "
                          "RMSProp + re-pack; finite={}".format(bool(torch.isfinite(core.flat).all())))
        core.set_params(P)  # back to the benchmark parameters for the parity / roofline legs below

    # ---- independent batches in flight (extra object "streams"): handles share nothing, so n of them replaying their graphs on
    # n streams overlap their dependent-launch chains.  Throughput of evaluation / serving, NOT the metric's one-pass-at-a-time rate.
    streams = None
    if world == 1 and use_graph and args.streams > 1:
        try:
            from sqair_amd.data import make_sequences, to_float
            cores = [core]
            for i in range(1, args.streams):
                d_i = make_sequences(B, T=T, canvas=hw, n_objects=(0, nums.shape[-1] - 1), obj_size=28 if hw[0] <= 64 else 72,
                                     seed=4321 + i)
                c_i = SqairCore(F, hw, device=device)
                with c_i.on_stream():
                    c_i.set_params(P)
                    Model(to_float(d_i["imgs"]), None, c_i, K, presence=d_i["nums"], outputs="minimal")
                cores.append(c_i)

            def round_(k):
                for i, c_i in enumerate(cores):
                    with c_i.on_stream():
                        c_i.draw_noise(seed=3000 + i, step=k, global_batch=B, b0=0)
                        c_i.forward(use_graph=True)
            for k in range(3):
                round_(k)
            torch.cuda.synchronize()
            n_rounds = max(5, args.steps // 2)
            t0 = time.perf_counter()
            for k in range(n_rounds):
                round_(3 + k)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            streams = dict(passes_in_flight=len(cores), value=len(cores) * B * T * n_rounds / el, unit="frames/s",
                           ms_per_round=el / n_rounds * 1e3, rounds=n_rounds,
                           what="{} independent batches of {} sequences (own handle, stream, graph, data, noise) per round; every "
                                "pass = noise draw + graph replay + ELBO".format(len(cores), B))
            del cores[1:]
            torch.cuda.set_stream(core.stream)
        except Exception as e:  # never take the bench line down
            streams = dict(error="{}: {}".format(type(e).__name__, e))

    if rank != 0:
        if dist is not None:
            dist.barrier()
            if comm is not None:
                comm.destroy()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (the fp32-MFMA dense layers) ----
    # Three measurements of "average dense-launch duration", all reported; `frac` uses (1):
    #  (1) HIP events, live: a graph holding ONLY the dense launches of the pass (same kernels / arguments / order) replayed
    #      between one event pair on the launch stream -> duration per launch INCLUDING the dependent kernel boundary, i.e. what
    #      a per-dispatch profiler sees, without a profiler attached;
    #  (2) device clock, live: first-workgroup-start -> last-workgroup-end per launch (s_memrealtime, stamped by the kernels in
    #      one eager pass) -> in-kernel time only, no boundary;
    #  (3) rocprofv3 --kernel-trace --stats of this command, committed under profiles/ for THIS build (build_id must match):
    #      per-dispatch duration under the profiler (its floor for a trivial dependent kernel is ~4.2-4.5 us on this stack).
    from sqair_amd._capi import build_id
    bid = build_id()
    prof = None
    for _ in range(3):
        prof = core.profile_linear()
    torch.cuda.synchronize()
    core.forward(use_graph=use_graph)          # a real pass: finite activations in the workspace for the dense-only replay
    if use_graph:
        lg = core.profile_linear_graph(replays=20)
    else:  # (--no-graph is what the PMC passes use: rocprofv3's counter collection crashes on graph launches)
        lg = dict(ms_per_replay=float("nan"), launches=prof["launches"], avg_launch_us=float("nan"))
    torch.cuda.synchronize()
    lin_ms = prof["linear_ms"]
    nh_in = 256 + 4 + 2 * int(F.n_what)
    algo_flops_step = float(B * T) * (FLOP_PER_FRAME_CFG2 if (args.cfg in (2, 3)) else
                                     2.0 * K * {1: 10166288, 4: 20298656, 5: 27760960}[args.cfg])
    if args.time_transition == "LSTM":  # a 4th gate over the same [x 360 | h 256] -> 256 input: + (360 + 256) * 256 MACs / slot
        algo_flops_step += float(B * T) * 2.0 * K * N * (nh_in + 256) * 256
    if args.prior_transition == "LSTM":  # likewise over [what, where 54 | h 256]
        algo_flops_step += float(B * T) * 2.0 * K * N * (int(F.n_what) + 4 + 256) * 256
    gates = {"VanillaRNN": 1, "GRU": 3, "LSTM": 4}[args.transition] - 1   # extra gate blocks of the two slot RNNs
    algo_flops_step += float(B * T) * 2.0 * K * N * gates * ((416 + 256) + (256 + 256 + int(F.n_what) + 5 + 256)) * 256
    n_launch = prof["launches"]
    algo_per_launch = algo_flops_step / n_launch
    exec_per_launch = prof["executed_flops"] / n_launch
    us_events = lg["avg_launch_us"]
    us_clock = lin_ms * 1e3 / n_launch

    def tf(flop, us):
        return flop / (us * 1e-6) / 1e12

    rocprof = None
    stats_path = os.path.join(ROOT, "profiles", "r02_kernel_stats.csv")
    meta_path = os.path.join(ROOT, "profiles", "r02_profile_meta.json")
    traffic = traffic_note = None
    if os.path.exists(meta_path):
        try:
            meta = json.load(open(meta_path))
            same_build = meta.get("build_id") == bid
            if os.path.exists(stats_path) and args.cfg == 2 and not args.batch:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from roofline_from_rocprof import dense_average_ns
                avg_ns, per = dense_average_ns(stats_path)
                dom = max(per, key=lambda k: per[k][0])
                rocprof = dict(avg_launch_us=avg_ns / 1e3, dominant=dom, dominant_avg_launch_us=per[dom][1] / 1e3,
                               frac=tf(algo_per_launch, avg_ns / 1e3) / PEAK_FP32_MFMA_TFLOPS,
                               frac_dominant=tf(algo_per_launch, per[dom][1] / 1e3) / PEAK_FP32_MFMA_TFLOPS,
                               same_build_as_this_run=same_build, profile_build_id=meta.get("build_id"),
                               recompute="python tools/roofline_from_rocprof.py profiles/r02_kernel_stats.csv")
            tpath = os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("build_id") == bid:
                    traffic = tj.get("dominant_bytes_per_launch")
                    traffic_note = "PMC FETCH_SIZE / WRITE_SIZE of {} on this build (profiles/r02_hbm_traffic.json, two separate --pmc passes)".format(tj.get("dominant"))
                else:
                    traffic_note = "profiles/r02_hbm_traffic.json was measured on build {} != this build {}: not quoted".format(tj.get("build_id"), bid)
        except Exception as e:  # a broken profile file must not take the bench line down
            traffic_note = "profiles unreadable: {}".format(e)
    roofline = dict(
        kernel="k_linear family (fp32 MFMA 16x16x4 dense layers, {} launches/step)".format(n_launch),
        bound="mfma", achieved=tf(algo_per_launch, us_events), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
        frac=tf(algo_per_launch, us_events) / PEAK_FP32_MFMA_TFLOPS, traffic=traffic, traffic_note=traffic_note,
        frac_is="frac_hip_events: algorithmic FLOPs per launch / average launch duration incl. the dependent kernel boundary, one "
                "HIP-event pair around 20 replays of a graph of the pass's dense launches only",
        frac_hip_events=tf(algo_per_launch, us_events) / PEAK_FP32_MFMA_TFLOPS,
        frac_device_clock=tf(algo_per_launch, us_clock) / PEAK_FP32_MFMA_TFLOPS,
        frac_rocprof=(rocprof or {}).get("frac"),
        frac_executed_hip_events=tf(exec_per_launch, us_events) / PEAK_FP32_MFMA_TFLOPS,
        frac_executed_device_clock=tf(exec_per_launch, us_clock) / PEAK_FP32_MFMA_TFLOPS,
        avg_launch_us=us_events, avg_launch_us_device_clock=us_clock, rocprof=rocprof,
        algorithmic_flops_per_launch=algo_per_launch, executed_flops_per_launch=exec_per_launch,
        dense_only_graph_ms=lg["ms_per_replay"], dense_share_of_step=lg["ms_per_replay"] / ms_per_step,
        step_ms_hip_events_eager=prof["forward_ms_events"], build_id=bid,
        note="algorithmic = as-reference FLOPs of the step (SURVEY.md 8(d): input encoder counted N times, mask MLP twice, as the "
             "reference graph computes them); executed = what the hoisted launch sequence runs.  Per-launch HIP events cannot "
             "resolve 2-5 us kernels (an empty pair costs ~8 us here), hence the dense-only graph.",
    )

    cpu = None
    if not args.no_cpu_baseline:
        # parity + timing on the same frames / parameters / noise as one GPU step
        full = Model(obs, None, core, K, presence=nums, outputs=["log_weights_per_timestep", "discrete_log_prob", "presence"])
        gen.manual_seed(4242)
        full.core.draw_noise(gen)
        noise = full.core.noise.cpu().numpy().copy()
        full.core.forward(use_graph=False)
        torch.cuda.synchronize()
        hip_ref = dict(presence=full.core.out["presence"].cpu().numpy(),
                       log_weights=full.core.log_weights.cpu().numpy().astype(np.float64),
                       elbo_iwae=float(full.core.scalars[1]))
        cpu = cpu_baseline(F, hw, P, obs, noise, hip_ref)

    line = {
        "metric": "frames/sec (forward IWAE ELBO, 10-step 50x50 moving glyphs, K=5)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "rccl_ranks": comm.n_ranks if comm is not None else 0, "dist_backend": (backend if dist is not None else None),
        "config": {"workload": "cfg{}: T={} HxW={}x{} B={}/GPU K={} N={} cells {}/{}/{} forward (elbo_iwae), HIP-graph replay={}".format(
            args.cfg, T, hw[0], hw[1], B, K, N, args.transition, args.time_transition, args.prior_transition, use_graph), "global_batch": B * world, "seq_len": T,
            "parallelism": "dp{}".format(world), "graph_nodes": core.graph_nodes()},
        "elbo_iwae_nats_per_seq": elbo, "elbo_vae_nats_per_seq": elbo_vae,
        "roofline": roofline, "cpu_baseline": cpu, "train": train, "streams": streams,
    }
    if cpu is not None:
        line["speedup_vs_cpu_baseline"] = value / world / cpu["value"]
        if train is not None:
            train["speedup_vs_cpu_baseline"] = train["value"] / world / cpu["train_value"]
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
