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

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline     — dominant kernel family (the fp32-MFMA dense layers k_linear* / k_rnn_tail) from a PER-DISPATCH
                 TIMELINE of one step measured live, without a profiler: the same step replayed on
                 libsqair_hip_timeline.so (the library compiled with -DSQAIR_TIMELINE: every wave stamps its start
                 and end on the device wall clock, sqair_amd/timeline.py).  achieved = algorithmic FLOPs per launch
                 (135.4 MFLOP/frame at cfg-2, as the reference graph computes them, / dense
                 launches per step) / average SLOT of a dense launch (its busy time + the dependent-launch gap up to
                 the next dispatch: the slots of all dispatches sum to the step); frac_busy_only and
                 frac_whole_step beside it; peak = 157.3 TFLOP/s.  The committed profiles/r05_timeline_*.csv
                 are the same measurement (tools/timeline.py; `--recompute` recomputes the fractions from them).
  roofline_hbm — the gather / scatter / reduce class (k_crop_row, k_insert_loglik, k_compact, k_logprob): busy time
                 from the same timeline, algorithmic bytes from the shapes (sqair_amd/timeline.py), PMC traffic
                 from profiles/ when it was measured on this build; peak = 8 TB/s.
  cpu_baseline — the oracle (PyTorch-CPU fp32 restatement at the reference's op granularity,
                 kind "port": the TF1 reference cannot run here) timed on the host cores on a bounded
                 sample of the same workload; only this leg imports oracle/.
  train        — the full training step on the same workload (reported separately), with the
                 collective that ran (`rccl_ranks`, `allreduce_on_launch_stream`, its measured time).
  single_gpu_at_global_batch — N > 1 only: rank 0 alone on B * N sequences (outside `value`), the strong-scaling reference.
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

# multi-process GPU work on this stack needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the boxes export it already --
# this only covers a shell that does not.  Must be in the environment before the HIP runtime starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense f32-in matrix peak (= vector peak)
MACS_PER_FRAME_PARTICLE = {1: 10166288, 2: 13543744, 3: 13543744, 4: 20298656, 5: 27760960}
# ... and what the path EXECUTES per frame (the loop-invariant input encoder hoisted out of the N slot steps, the
# mask MLP evaluated once): MFLOP per frame, all K particles included.  `roofline.frac_executed` is quoted on this figure.
EXECUTED_MFLOP_PER_FRAME = {1: 17.5, 2: 114.3, 3: 114.3, 4: 167.7, 5: 149.8}
MIN_HBM_BYTES_PER_FRAME = {2: 101060, 3: 101060}   # minimum HBM bytes per frame (all K particles), cfg-2 shapes
PROFILE_TAG = "r06"                    # profiles/<tag>_*.json|csv: the committed files of this round (tools/profile_round.sh)


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


def timed_steps(step_fn, n, dist, device):
    """The contract's timed region: barrier + synchronize on both sides, EXACTLY n steps, MAX over ranks."""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step_fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el


def timeline_roofline(F, Ftr, hw, P, obs, nums, algo_flops_step, ms_fwd, ms_train, cfg_id, batch_override, bid):
    """Per-dispatch timeline of one forward and one training step on the timeline build of the library (sqair_amd/timeline.py)
    -> the `roofline` / `roofline_hbm` objects and a busy / gap summary of the training step."""
    from sqair_amd import timeline as TL
    from sqair_amd.train import Trainer
    T, B = int(obs.shape[0]), int(obs.shape[1])
    K, N = int(F.k_particles), int(F.n_steps_per_image)
    core_t, model_t = TL.make_model(F, hw, P, obs, nums, timeline=True, device="cuda:{}".format(torch.cuda.current_device()))
    tl = TL.Timeline(core_t, mbytes=768 * max(1, B // 32))
    n = [0]

    def fwd():
        core_t.draw_noise(seed=1000, step=n[0], global_batch=B, b0=0)
        n[0] += 1
        core_t.forward(use_graph=True)
    ms_t = TL.time_steps(core_t, fwd, steps=10)
    rows, ev_ms = tl.measure(fwd, warm=2)
    s, d = TL.summarise(rows, ev_ms), TL.dense_stats(rows)
    alg = TL.algorithmic_hbm_bytes(T, B, K, N, hw[0], hw[1], G=int(F.glimpse_size), nh=core_t.nh, nw=core_t.nw, snh=core_t.snh,
                                   psnh=core_t.psnh, masked=bool(F.masked_glimpse), train=False)
    # PMC traffic measured by rocprofv3 on this build, if committed (tools/profile_round.sh)
    traffic, traffic_note, fam_traffic, dense_fam_traffic = None, None, None, None
    tname = PROFILE_TAG + "_hbm_traffic" + ("" if cfg_id in (2, 3) else "_cfg{}".format(cfg_id)) + ".json"
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath) and not batch_override:
        try:
            tj = json.load(open(tpath))
            if tj.get("build_id") == bid:
                traffic = tj.get("dominant_bytes_per_launch")
                fam_traffic = tj.get("family_bytes_per_launch")
                dense_fam_traffic = tj.get("dense_family_bytes_per_launch")
                traffic_note = "PMC FETCH_SIZE / WRITE_SIZE of {} on this build (profiles/{}, two separate --pmc passes)".format(tj.get("dominant"), tname)
            else:
                traffic_note = "profiles/{} was measured on build {} != this build {}: not quoted".format(tname, tj.get("build_id"), bid)
        except Exception as e:  # a broken profile file must not take the bench line down
            traffic_note = "profiles unreadable: {}".format(e)
    per = algo_flops_step / max(d["launches"], 1)
    exec_ratio = EXECUTED_MFLOP_PER_FRAME[cfg_id] * 1e6 / (2.0 * K * MACS_PER_FRAME_PARTICLE[cfg_id])

    def frac(us):
        return per / (us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS
    committed = None
    cpath = os.path.join(ROOT, "profiles", PROFILE_TAG + "_timeline.json")
    if os.path.exists(cpath) and cfg_id == 2 and not batch_override:
        try:
            cj = json.load(open(cpath))
            committed = dict(file="profiles/{0}_timeline.json (+ {0}_timeline_fwd.csv, {0}_timeline_train.csv)".format(PROFILE_TAG),
                             same_build_as_this_run=cj.get("build_id") == bid, build_id=cj.get("build_id"),
                             frac_slot=cj["fwd"]["dense_frac"]["frac_slot"], frac_busy_only=cj["fwd"]["dense_frac"]["frac_busy_only"],
                             recompute="python tools/timeline.py --recompute profiles/{}_timeline_fwd.csv".format(PROFILE_TAG))
        except Exception as e:
            committed = dict(error=str(e))
    # PMC traffic against what the launches NEED: the dominant instantiation's unique operands (rows x K in, K x N weights, rows x N
    # out of the 160 x 256 x 256 slot layer), and the dense family's bytes per step against SURVEY 8(d)'s minimum for the whole
    # step (activations that must cross HBM + every weight once)
    traffic_ratio = None
    if traffic is not None:
        R_ = B * K
        uniq = 4.0 * (R_ * core_t.nh + core_t.nh * core_t.nh + R_ * core_t.nh)
        traffic_ratio = dict(dominant_over_unique_operands=traffic / uniq, dominant_unique_operand_bytes=uniq)
        if dense_fam_traffic is not None and cfg_id in MIN_HBM_BYTES_PER_FRAME:
            need = MIN_HBM_BYTES_PER_FRAME[cfg_id] * float(B * T) + 4.0 * core_t.n_params
            traffic_ratio.update(dense_family_bytes_per_launch=dense_fam_traffic, dense_family_bytes_per_step=dense_fam_traffic * d["launches"],
                                 algorithmic_min_bytes_per_step=need,
                                 dense_family_over_algorithmic_min=dense_fam_traffic * d["launches"] / need,
                                 why="each of the 8 XCD L2s pulls a layer's weights again on every launch (8 x 256 KB for the dominant one); "
                                     "not the bound: with the weight loads removed the step is 3 % faster (DESIGN.md section 3)")
    # third leg (profiles/<tag>_dense_b2b.json, tools/dense_graph_time.py): per-node time of 1000-node HIP graphs of every dense shape
    # of the step, PRODUCT library, plain HIP events -- a figure for the same fraction that the stamped build did not produce
    third = None
    bpath = os.path.join(ROOT, "profiles", PROFILE_TAG + "_dense_b2b.json")
    if os.path.exists(bpath) and cfg_id == 2 and not batch_override:
        try:
            bj = json.load(open(bpath))
            third = dict(file="profiles/{}_dense_b2b.json".format(PROFILE_TAG), same_build_as_this_run=bj.get("build_id") == bid,
                         build_id=bj.get("build_id"), frac_dense_family=bj.get("frac_dense_family"),
                         k_linear_avg_node_us=bj.get("k_linear_avg_node_us"),
                         graph_nodes_over_timeline_slots_k_linear=bj.get("graph_nodes_over_timeline_slots_k_linear"),
                         how="census of the step's dense launches x microseconds per node of a 1000-node HIP graph of each shape "
                             "(every launch reading what the previous one wrote), product library, HIP events")
        except Exception as e:
            third = dict(error=str(e))
    roofline = dict(
        kernel="fp32-MFMA dense layers (k_linear*, k_rnn_tail; {} launches/step)".format(d["launches"]),
        bound="mfma", achieved=per / (d["avg_slot_us"] * 1e-6) / 1e12, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
        frac=frac(d["avg_slot_us"]), traffic=traffic, traffic_note=traffic_note, traffic_over_algorithmic=traffic_ratio,
        third_leg=third,
        frac_is="frac_slot: algorithmic FLOPs per dense launch / average slot of a dense launch (first-wave start to the next "
                "dispatch's first-wave start = busy + dependent-launch gap), from the per-dispatch timeline of one step stamped "
                "by the kernels themselves (no profiler)",
        frac_slot=frac(d["avg_slot_us"]), frac_busy_only=frac(d["avg_busy_us"]),
        # the same slot fraction on the FLOPs the path executes (input encoder hoisted, mask MLP once): what the matrix cores
        # are really asked to do per launch; `frac` stays on the as-reference count, which is the contract
        frac_executed=frac(d["avg_slot_us"]) * exec_ratio, executed_over_algorithmic_flops=exec_ratio,
        frac_whole_step=algo_flops_step / (ms_fwd * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        avg_slot_us=d["avg_slot_us"], avg_busy_us=d["avg_busy_us"], algorithmic_flops_per_launch=per,
        timeline=dict(dispatches=s["dispatches"], span_us=s["span_us"], busy_us=s["busy_us"], gap_us=s["gap_us"],
                      overlap_us=s["overlap_us"], this_step_hip_event_ms=ev_ms, timeline_build_ms_per_step=ms_t,
                      product_ms_per_step=ms_fwd, stamp_overhead=ms_t / ms_fwd - 1.0,
                      span_over_product_step=s["span_us"] / (ms_fwd * 1e3),
                      dense_share_of_step=d["slot_us"] / max(s["slot_sum_us"], 1e-9),
                      families={k: dict(launches=v["launches"], slot_us=round(v["slot_us"], 1), busy_us=round(v["busy_us"], 1))
                                for k, v in sorted(s["families"].items(), key=lambda kv: -kv[1]["slot_us"])}),
        committed_profile=committed, build_id=bid,
        note="algorithmic = as-reference FLOPs of the step (SURVEY.md 8(d): input encoder counted N times, mask MLP twice, as the "
             "reference graph computes them) / dense launches of the step; slots of ALL dispatches sum to the step (span); the "
             "stamped build runs the step `stamp_overhead` slower than the product library.")
    roofline_hbm = TL.hbm_class(rows, alg, fam_traffic)
    train_tl = None
    if ms_train is not None:
        tr = Trainer(model_t, Ftr, use_graph=True, collective=False)
        step_t = lambda: tr.step(seed=2000, global_batch=B, b0=0)  # noqa: E731
        ms_tt = TL.time_steps(core_t, step_t, steps=5)
        rows_t, ev_t = tl.measure(step_t, warm=2)
        st = TL.summarise(rows_t, ev_t)
        algt = TL.algorithmic_hbm_bytes(T, B, K, N, hw[0], hw[1], G=int(F.glimpse_size), nh=core_t.nh, nw=core_t.nw,
                                        snh=core_t.snh, psnh=core_t.psnh, masked=bool(F.masked_glimpse), train=True)
        train_tl = dict(dispatches=st["dispatches"], span_us=st["span_us"], busy_us=st["busy_us"], gap_us=st["gap_us"],
                        timeline_build_ms_per_step=ms_tt, stamp_overhead=ms_tt / ms_train - 1.0,
                        families={k: dict(launches=v["launches"], slot_us=round(v["slot_us"], 1), busy_us=round(v["busy_us"], 1))
                                  for k, v in sorted(st["families"].items(), key=lambda kv: -kv[1]["slot_us"])[:12]},
                        hbm_class=[e for e in TL.hbm_class(rows_t, algt) if e["kernel"].endswith("_bwd")])
    tl.close()
    return roofline, roofline_hbm, train_tl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-timeline", action="store_true", help="skip the per-dispatch timeline (roofline objects become null)")
    ap.add_argument("--cfg", type=int, default=2, help="BASELINE.json config id used as the workload (default 2)")
    ap.add_argument("--train-steps", type=int, default=-1,
                    help="steps of the extra training-step leg (default min(steps, 20); 0 = skip)")
    ap.add_argument("--streams", type=int, default=3,
                    help="extra object \"streams\": forward passes of this many INDEPENDENT batches in flight on their own "
                         "streams (0 = skip; single GPU only; never part of `value`)")
    ap.add_argument("--batch", type=int, default=0,
                    help="sequences per GPU (default: the configuration's own; other values are batch-scaling experiments, "
                         "not BASELINE's metric)")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend (default: nccl = RCCL when every rank has its own device, gloo when ranks share "
                         "devices, e.g. a 2-rank functional check on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="initialise a 1-rank process group + RCCL communicator at N = 1")
    ap.add_argument("--time-transition", default="GRU", choices=["GRU", "LSTM"],
                    help="propagation temporal cell (the shipped config and BASELINE's metric use GRU)")
    ap.add_argument("--prior-transition", default="GRU", choices=["GRU", "LSTM"], help="propagation prior cell (shipped: GRU)")
    ap.add_argument("--require-native-comm", action="store_true",
                    help="with a device per rank: exit with status 3 if the native RCCL communicator (sqair_amd/rccl.py) cannot be "
                         "built, instead of sending the gradient all-reduce through torch.distributed's nccl group (reported)")
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
    # RCCL wants one device per rank.  With fewer visible devices than ranks (a 2-rank functional check on a 1-GPU box) the
    # ranks share devices and the process group is gloo — reported as such ("rccl_ranks": 0).  With a device per rank the
    # native RCCL communicator is REQUIRED: failing to build it is an error (non-zero exit), never a silent fallback.
    own_device = n_dev >= world
    backend = args.dist_backend or ("nccl" if own_device else "gloo")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = "cuda:{}".format(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = dict(device_id=torch.device("cuda", local_rank)) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, **kw)  # "nccl" = RCCL on ROCm
    # the training step's gradient all-reduce goes through RCCL's C API on the library's own launch stream
    comm = None
    native_comm_error = None
    if dist is not None and backend == "nccl":
        from sqair_amd.rccl import RcclComm
        comm_error = None
        try:
            comm = RcclComm.from_process_group(device)
        except Exception as e:
            comm_error = "{}: {}".format(type(e).__name__, e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            # Never silently: the failure goes to stderr on every rank and into the JSON line (`native_comm_error`,
            # `allreduce_on_launch_stream` false).  With --require-native-comm the run ends here (status 3); otherwise the step's
            # all-reduce goes through torch.distributed's "nccl" group -- RCCL over xGMI all the same, but issued from torch's
            # collective stream instead of the library's launch stream.
            native_comm_error = comm_error or "failed on another rank"
            print("bench: rank {}: the native RCCL communicator could not be built ({}); {}".format(
                rank, native_comm_error, "refusing to fall back" if args.require_native_comm else
                "the gradient all-reduce goes through torch.distributed (nccl = RCCL) instead"), file=sys.stderr, flush=True)
            if comm is not None:
                comm.destroy()
                comm = None
            if args.require_native_comm:
                dist.destroy_process_group()
                raise SystemExit(3)
        else:
            assert comm.n_ranks == world

    # which communicator carries the training step's gradient all-reduce -- said once, unambiguously
    if dist is None:
        collective_path = None            # one rank, no process group: no collective
    elif comm is not None:
        collective_path = "native"        # ncclAllReduce through RCCL's C API on the library's launch stream (sqair_amd/rccl.py)
    elif backend == "nccl":
        collective_path = "torch-nccl"    # torch.distributed's nccl group (= RCCL) on torch's collective stream
    else:
        collective_path = "gloo"          # ranks share devices: functional check only
    rccl_ranks = comm.n_ranks if collective_path == "native" else (world if collective_path == "torch-nccl" else 0)

    from sqair_amd._capi import build_id
    from sqair_amd.data import config_inputs, make_sequences, to_float
    from sqair_amd.flags import make_flags
    from sqair_amd.model import Model, SqairCore
    from sqair_amd.params import init_params
    from sqair_amd.train import Trainer

    # every rank synthesises its own shard of the global batch (seeded by rank), weights are replicated
    ov, obs, nums, _ = config_inputs(args.cfg, B=args.batch or None)

    def shard_data(n_seq, seed):
        d = make_sequences(n_seq, T=obs.shape[0], canvas=obs.shape[2:], n_objects=(0, nums.shape[-1] - 1),
                           obj_size=28 if obs.shape[2] <= 64 else 72, seed=seed)
        return to_float(d["imgs"]), d["nums"]
    if rank > 0:
        obs, nums = shard_data(obs.shape[1], 1234 + args.cfg + 1000 * rank)
    ov.update(time_transition=args.time_transition, prior_transition=args.prior_transition, transition=args.transition)
    F = make_flags(**ov)
    Ftr = make_flags(**dict(ov, learning_rate=1e-5, train_itr=1000000))
    hw = tuple(int(v) for v in obs.shape[2:])
    T, B = int(obs.shape[0]), int(obs.shape[1])
    K, N = int(F.k_particles), int(F.n_steps_per_image)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in
         init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    core = SqairCore(F, hw, device=device)
    # one stream for everything (noise draw, graph replays, optimiser, reductions): see SqairCore.on_stream
    torch.cuda.set_stream(core.stream)
    core.set_params(P)
    model = Model(obs, None, core, K, presence=nums, outputs="minimal")
    use_graph = not args.no_graph
    bid = build_id()

    step_no = [0]

    def step():
        # device-side Philox noise keyed by (seed, step, position in the GLOBAL batch): ranks draw what one GPU would draw
        model.core.draw_noise(seed=1000, step=step_no[0], global_batch=B * world, b0=rank * B)
        step_no[0] += 1
        model.core.forward(use_graph=use_graph)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        # RCCL writes its start-up banner (NCCL_DEBUG=VERSION on the GPU boxes) to the C stdout buffer at communicator
        # creation: push it out now, on every rank, so that rank 0's JSON line is the LAST thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    elapsed = timed_steps(step, args.steps, dist, device)
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
        trainer = Trainer(model, Ftr, use_graph=use_graph, comm=comm)
        tstep = lambda: trainer.step(seed=2000, global_batch=B * world, b0=rank * B)  # noqa: E731
        for _ in range(max(2, min(args.warmup, 3))):
            tstep()
        el = timed_steps(tstep, n_train, dist, device)
        allreduce_ms = None
        if world > 1:  # the collective alone, same buffer, same stream (20 back to back between one event pair)
            with core.on_stream():
                from sqair_amd.dist import allreduce_flat_grads
                g = core.flat_grad.clone()
                for _ in range(3):
                    allreduce_flat_grads(g, comm=comm, stream=core.stream)
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(core.stream)
                for _ in range(20):
                    allreduce_flat_grads(g, comm=comm, stream=core.stream)
                e1.record(core.stream)
                torch.cuda.synchronize()
                allreduce_ms = e0.elapsed_time(e1) / 20
        train = dict(value=frames_per_step / (el / n_train), unit="frames/s", ms_per_step=el / n_train * 1e3, steps=n_train,
                     scaling="weak", graph_nodes=getattr(core, "train_graph_nodes", None),
                     rccl_ranks=rccl_ranks, collective_path=collective_path,
                     allreduce_on_launch_stream=bool(comm is not None) if world > 1 else None,
                     native_comm_error=native_comm_error,
                     allreduce_ms=allreduce_ms,
                     collective="all-reduce(sum) of {} fp32 gradients ({:.1f} MB) per step, {}; 1/world folded into the fused "
                                "RMSProp kernel".format(core.n_params, core.n_params * 4 / 1e6,
                                                        "ncclAllReduce (RCCL C API) on the launch stream" if comm is not None
                                                        else ("torch.distributed nccl (= RCCL) on torch's collective stream: the native "
                                                              "communicator could not be built" if native_comm_error else
                                                              "torch.distributed " + backend + " (ranks share devices: functional "
                                                              "check only)")) if world > 1 else "none (1 rank)",
                     what="draw noise + forward(train) + VIMCO target + backward (one HIP-graph replay) + all-reduce + "
                          "RMSProp + re-pack; finite={}".format(bool(torch.isfinite(core.flat).all())))
        core.set_params(P)  # back to the benchmark parameters for the parity / roofline legs below

    # ---- N > 1: what the training step SHOULD take = this rank's step without the collective + the collective alone; the ratio
    # flags a degraded all-reduce (or a stream / queue problem) in the driver's scaling curve by itself
    if train is not None and world > 1:
        if rank == 0:
            try:
                from sqair_amd.timeline import time_steps
                tr1 = Trainer(model, Ftr, use_graph=use_graph, comm=None, collective=False)
                with core.on_stream():
                    ms_1 = time_steps(core, lambda: tr1.step(seed=2000, global_batch=B * world, b0=rank * B), steps=max(3, n_train // 2), warm=2)
                del tr1
                core.set_params(P)
                train["single_rank_ms_per_step_no_collective"] = ms_1
                if train.get("allreduce_ms") is not None:
                    train["expected_ms"] = ms_1 + train["allreduce_ms"]
                    train["measured_over_expected"] = train["ms_per_step"] / train["expected_ms"]
            except Exception as e:
                train["expected_ms_error"] = "{}: {}".format(type(e).__name__, e)
        torch.cuda.synchronize()
        dist.barrier()

    # ---- N > 1: the strong-scaling reference.  Rank 0 alone processes the GLOBAL batch (B * world sequences) on its GPU, after
    # the timed regions and outside `value`: weak scaling says N GPUs do N x the work in the same time; this says what ONE GPU
    # needs for the same N x work (the small-batch step is latency-bound, so one GPU is far better than 1/N of the job).
    single = None
    if world > 1:
        if rank == 0:
            try:
                obs_g, nums_g = shard_data(B * world, 99)
                core_g = SqairCore(F, hw, device=device)
                with core_g.on_stream():
                    core_g.set_params(P)
                    model_g = Model(obs_g, None, core_g, K, presence=nums_g, outputs="minimal")
                    k = [0]

                    def gstep():
                        core_g.draw_noise(seed=1000, step=k[0], global_batch=B * world, b0=0)
                        k[0] += 1
                        core_g.forward(use_graph=use_graph)
                    from sqair_amd.timeline import time_steps
                    ms_g = time_steps(core_g, gstep, steps=max(5, args.steps // 4), warm=3)
                    single = dict(sequences=B * world, forward_ms_per_step=ms_g, forward_value=B * world * T / (ms_g * 1e-3), unit="frames/s")
                    if n_train > 0:
                        tr_g = Trainer(model_g, Ftr, use_graph=use_graph, comm=None, collective=False)
                        ms_gt = time_steps(core_g, lambda: tr_g.step(seed=2000, global_batch=B * world, b0=0), steps=max(3, n_train // 4), warm=2)
                        single.update(train_ms_per_step=ms_gt, train_value=B * world * T / (ms_gt * 1e-3))
                        del tr_g
                    single["speedup_of_{}_gpus_over_one_gpu_same_global_batch".format(world)] = dict(
                        forward=value / single["forward_value"],
                        train=(train["value"] / single["train_value"]) if train and "train_value" in single else None)
                del model_g, core_g
                torch.cuda.empty_cache()
                torch.cuda.set_stream(core.stream)
            except Exception as e:  # never take the bench line down
                single = dict(error="{}: {}".format(type(e).__name__, e))
        torch.cuda.synchronize()
        dist.barrier()

    # ---- independent batches in flight (extra object "streams"): handles share nothing, so n of them replaying their graphs on
    # n streams overlap their dependent-launch chains.  Throughput of evaluation / serving, NOT the metric's one-pass-at-a-time rate.
    streams = None
    if world == 1 and use_graph and args.streams > 1:
        try:
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

    # (after the `streams` leg: every stream ever created takes a hardware queue slot round-robin, and an extra one ahead of that leg
    #  put two of its three streams on one queue: 131 k instead of 206 k frames/s)
    # ---- what the evaluation path costs: the same forward step with ALL 38 per-frame outputs of SequentialAIR requested (canvases,
    # glimpses, every log-probability: reference sqair/seq.py:121-177 writes them all, scripts/eval.py:191-224 and the notebook read
    # them); the timed region above asks for `elbo_iwae` only (outputs="minimal", what `sess.run(model.elbo_iwae)` evaluates).
    all_out = None
    if rank == 0 and use_graph:
        try:
            from sqair_amd.timeline import time_steps
            core_a = SqairCore(F, hw, device=device)
            with core_a.on_stream():
                core_a.set_params(P)
                Model(obs, None, core_a, K, presence=nums, outputs="all")
                ka = [0]

                def astep():
                    core_a.draw_noise(seed=1000, step=ka[0], global_batch=B * world, b0=rank * B)
                    ka[0] += 1
                    core_a.forward(use_graph=True)
                ms_a = time_steps(core_a, astep, steps=max(5, args.steps // 2), warm=3)
            all_out = dict(ms_per_step=ms_a, value=B * T / (ms_a * 1e-3), unit="frames/s (this rank)", graph_nodes=core_a.graph_nodes(),
                           outputs=len(core_a.out), what="noise draw + graph replay + ELBO with every SqairOutputs field written")
            del core_a
            torch.cuda.empty_cache()
            torch.cuda.set_stream(core.stream)
        except Exception as e:  # never take the bench line down
            all_out = dict(error="{}: {}".format(type(e).__name__, e))
    # ---- the in-launch slot chain (extra object "slot_chain"; option `slot_chain` of the library, off by default and NOT part of
    # `value`): the same forward step with the slot launches of every frame's propagation / discovery loop run as one persistent
    # launch each (sqair_amd/csrc/sqair_chain.h; bit-identical results: tests/test_slot_chain.py) -- at this workload and at half
    # the sequences, where every XCD serves a single 16-row tile.
    chain = None
    if rank == 0 and world == 1 and use_graph and args.transition == "VanillaRNN" and args.time_transition == "GRU":
        try:
            from sqair_amd.timeline import time_steps
            chain = {}
            for tag, nb in (("this_workload", B), ("half_the_sequences", max(B // 2, 1))):
                ob, nu = (obs, nums) if nb == B else (obs[:, :nb], nums[:, :nb])
                res = {}
                for name, opts in (("launches", None), ("chain", {"slot_chain": 1})):
                    if name == "launches" and nb == B:
                        res[name] = dict(ms_per_step=ms_per_step, graph_nodes=core.graph_nodes())
                        continue
                    c_c = SqairCore(F, hw, device=device, options=opts)
                    with c_c.on_stream():
                        c_c.set_params(P)
                        m_c = Model(ob, None, c_c, K, presence=nu, outputs="minimal")
                        kc = [0]

                        def cstep():
                            c_c.draw_noise(seed=1000, step=kc[0], global_batch=nb, b0=0)
                            kc[0] += 1
                            c_c.forward(use_graph=True)
                        ms_c = time_steps(c_c, cstep, steps=max(10, args.steps // 2), warm=3)
                        if opts:
                            c_c.check_chain()
                        m_c._collect()
                        res[name] = dict(ms_per_step=ms_c, graph_nodes=c_c.graph_nodes(), elbo_iwae=float(m_c.elbo_iwae))
                    del m_c, c_c
                    torch.cuda.empty_cache()
                chain[tag] = dict(sequences=nb, particle_rows=nb * K, launches=res["launches"], chain=res["chain"],
                                  chain_over_launches=res["chain"]["ms_per_step"] / res["launches"]["ms_per_step"],
                                  value=nb * T / (res["chain"]["ms_per_step"] * 1e-3), unit="frames/s")
            chain["what"] = ("option slot_chain (default off): noise draw + graph replay + ELBO, the slot loops as two persistent launches "
                             "per frame; needs the device to itself (its 256 workgroups must be co-resident)")
            torch.cuda.set_stream(core.stream)
        except Exception as e:  # never take the bench line down
            chain = dict(error="{}: {}".format(type(e).__name__, e))
    if rank != 0:
        if dist is not None:
            dist.barrier()
            if comm is not None:
                comm.destroy()
            dist.destroy_process_group()
        return

    # ---- roofline: per-dispatch timeline of one step, stamped by the kernels (module docstring; sqair_amd/timeline.py) ----
    algo_flops_step = float(B * T) * 2.0 * K * MACS_PER_FRAME_PARTICLE[args.cfg]
    nh_in = 256 + 4 + 2 * int(F.n_what)
    if args.time_transition == "LSTM":  # a 4th gate over the same [x 360 | h 256] -> 256 input: + (360 + 256) * 256 MACs / slot
        algo_flops_step += float(B * T) * 2.0 * K * N * (nh_in + 256) * 256
    if args.prior_transition == "LSTM":  # likewise over [what, where 54 | h 256]
        algo_flops_step += float(B * T) * 2.0 * K * N * (int(F.n_what) + 4 + 256) * 256
    gates = {"VanillaRNN": 1, "GRU": 3, "LSTM": 4}[args.transition] - 1   # extra gate blocks of the two slot RNNs
    algo_flops_step += float(B * T) * 2.0 * K * N * gates * ((416 + 256) + (256 + 256 + int(F.n_what) + 5 + 256)) * 256
    roofline = roofline_hbm = None
    if use_graph and not args.no_timeline:
        try:
            roofline, roofline_hbm, train_tl = timeline_roofline(
                F, Ftr, hw, P, obs, nums, algo_flops_step, ms_per_step, train["ms_per_step"] if train else None, args.cfg,
                args.batch, bid)
            if train is not None:
                train["timeline"] = train_tl
            torch.cuda.set_stream(core.stream)
        except Exception as e:  # the timeline library missing / failing must not take the bench line down, but it must show
            roofline = dict(error="{}: {}".format(type(e).__name__, e), bound="mfma", achieved=None, peak=PEAK_FP32_MFMA_TFLOPS,
                            unit="TFLOP/s", frac=None, traffic=None)
    # the whole-step figure needs no instrumentation at all: algorithmic FLOPs of the step / the timed step
    whole = algo_flops_step / (ms_per_step * 1e-3) / 1e12
    if roofline is None:
        roofline = dict(kernel="whole step (no timeline requested)", bound="mfma", achieved=whole, peak=PEAK_FP32_MFMA_TFLOPS,
                        unit="TFLOP/s", frac=whole / PEAK_FP32_MFMA_TFLOPS, traffic=None)

    cpu = None
    if not args.no_cpu_baseline:
        # parity + timing on the same frames / parameters / noise as one GPU step
        full = Model(obs, None, core, K, presence=nums, outputs=["log_weights_per_timestep", "discrete_log_prob", "presence"])
        gen = torch.Generator(device=device)
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
        "metric": "frames/sec (forward IWAE ELBO, {}-step {}x{} moving glyphs, K={})".format(T, hw[0], hw[1], K),
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # ranks of the RCCL communicator that carried the gradient all-reduce of the `train` leg (0: none did -- one rank, or gloo)
        "rccl_ranks": rccl_ranks, "collective_path": collective_path,
        "native_comm_error": native_comm_error, "dist_backend": (backend if dist is not None else None),
        "config": {"workload": "cfg{}: T={} HxW={}x{} B={}/GPU K={} N={} cells {}/{}/{} forward (elbo_iwae), HIP-graph replay={}".format(
            args.cfg, T, hw[0], hw[1], B, K, N, args.transition, args.time_transition, args.prior_transition, use_graph), "global_batch": B * world, "seq_len": T,
            "parallelism": "dp{}".format(world), "graph_nodes": core.graph_nodes()},
        "elbo_iwae_nats_per_seq": elbo, "elbo_vae_nats_per_seq": elbo_vae,
        "roofline": roofline, "roofline_hbm": roofline_hbm, "cpu_baseline": cpu, "train": train,
        "forward_all_outputs": all_out, "forward_all_outputs_ms": (all_out or {}).get("ms_per_step"),
        "single_gpu_at_global_batch": single, "streams": streams, "slot_chain": chain, "build_id": bid,
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
