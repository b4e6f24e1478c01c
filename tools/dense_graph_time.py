#!/usr/bin/env python
"""A third leg for the dense family's roofline, on the PRODUCT library and with plain HIP events (no stamps, no profiler):

  1. census: every dense launch of one cfg-2 forward pass (layer id, rows, K, N) read from the library's host-side launch log;
  2. for every distinct (rows, K, N) of it, a HIP graph of 1000 launches of that layer shape -- the same launch repeated, and
     with every launch reading what the previous one wrote (in place) -- replayed between two HIP events: microseconds per graph NODE, i.e.
     kernel + dependent-dispatch boundary, the quantity bench.py's timeline calls the SLOT of a dense launch;
  3. census x node time = what the k_linear* launches of the step should take; next to it the same sum from the committed
     per-dispatch timeline (profiles/<tag>_timeline_fwd.csv, stamped build) and the roofline fraction either gives.

    python tools/dense_graph_time.py [--cfg 2] [--out profiles/r06_dense_b2b.json] [--timeline profiles/r06_timeline_fwd.csv]

The ad-hoc layers are single-segment, bias + activation epilogue: the GRU epilogues and multi-segment A operands of the real pass
are timed as a plain layer of the same shape (the instantiation differs in its epilogue, not in its tile or K loop)."""
import argparse
import collections
import csv
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sqair_amd import _capi  # noqa: E402
from sqair_amd.data import config_inputs  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import Model, SqairCore  # noqa: E402
from sqair_amd.params import init_params  # noqa: E402

PEAK = 157.3
MACS = {1: 10166288, 2: 13543744, 3: 13543744, 4: 20298656, 5: 27760960}   # SURVEY.md 8(d), per frame-particle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--out", default=None)
    ap.add_argument("--timeline", default=None, help="per-dispatch timeline CSV of the same workload to compare with")
    ap.add_argument("--nodes", type=int, default=1000)
    ap.add_argument("--replays", type=int, default=10)
    args = ap.parse_args()
    ov, obs, nums, _ = config_inputs(args.cfg)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    T, B, K = int(obs.shape[0]), int(obs.shape[1]), int(F.k_particles)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    core = SqairCore(F, hw)
    core.set_params(P)
    m = Model(obs, None, core, K, presence=nums, outputs="minimal")
    lib = core.lib
    m.run(use_graph=False)
    assert lib.sqair_debug_dense_log(core.handle, 1) == 0
    m.run(use_graph=False)
    n = lib.sqair_debug_dense_log(core.handle, 0)
    census = collections.Counter()
    e = (C.c_int * 4)()
    for i in range(n):
        assert lib.sqair_debug_dense_log_entry(core.handle, i, e) == 0
        census[(e[0] < 0, e[1], e[2], e[3])] += 1
    # the product step itself between HIP events (graph replay)
    from sqair_amd import timeline as TL
    k = [0]

    def fwd():
        core.draw_noise(seed=1000, step=k[0], global_batch=B, b0=0)
        k[0] += 1
        core.forward(use_graph=True)
    ms_step = TL.time_steps(core, fwd, steps=30, warm=5)
    torch.cuda.set_stream(core.stream)   # (HIP refuses to capture the legacy stream)
    s = C.c_void_p(core.stream.cuda_stream)
    h = core.handle
    shapes = []
    tot_plain = tot_rep = 0.0
    for (fused, M, Kd, N), cnt in sorted(census.items(), key=lambda kv: -kv[1]):
        x = torch.randn(M, Kd, device="cuda") * 0.1
        w = torch.randn(Kd, N, device="cuda") / np.sqrt(Kd)
        b = torch.randn(N, device="cuda") * 0.1
        y = torch.zeros(M, N, device="cuda")
        nt, kc = (N + 15) // 16, (Kd + 15) // 16
        scratch = torch.zeros(2 * nt * kc * 256 + 2 * nt * 16 + 256 + M * ((Kd + 3) // 4 * 4) + M * (max(Kd, N) + 4) + 128, device="cuda")
        res = {}
        for dep in (0, 1):
            us = C.c_float()
            nodes = args.nodes if M * N <= (1 << 20) else max(50, args.nodes // 10)
            rc = lib.sqair_debug_linear_graph_time(h, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, Kd, N, 2, scratch.data_ptr(),
                                                   scratch.numel() * 4, nodes, args.replays, dep, C.byref(us), s)
            assert rc == 0, lib.sqair_last_error(h)
            res["dependent" if dep else "repeated"] = float(us.value)
        node = res.get("dependent", res["repeated"])
        shapes.append(dict(rows=M, K=Kd, N=N, launches_per_step=cnt, with_slot_tail_in_front=bool(fused), node_us=res,
                           mflop=2.0 * M * Kd * N / 1e6))
        if not fused:
            tot_plain += cnt * node
        tot_rep += cnt * res["repeated"]
        print("rows %5d K %4d N %4d x %3d%s : node %s us" % (M, Kd, N, cnt, " (+tail)" if fused else "",
                                                            ", ".join("%s %.2f" % kv for kv in res.items())), flush=True)
    n_plain = sum(c for (f, _, _, _), c in census.items() if not f)
    n_all = sum(census.values())
    algo = float(B * T) * 2.0 * K * MACS[args.cfg]
    out = dict(build_id=_capi.build_id(), library="product (libsqair_hip.so), HIP events around graph replays; no stamps, no profiler",
               cfg=args.cfg, product_step_ms=ms_step, dense_launches_per_step=n_all, k_linear_launches_per_step=n_plain,
               nodes_per_graph=args.nodes, replays=args.replays, shapes=shapes,
               k_linear_sum_us=tot_plain, k_linear_avg_node_us=tot_plain / max(n_plain, 1),
               algorithmic_flops_per_step=algo, algorithmic_mflop_per_dense_launch=algo / n_all / 1e6,
               what="node_us = microseconds per node of a 1000-node HIP graph of that layer shape (kernel + dependent-dispatch boundary): "
                    "`repeated` = the same launch again and again, `dependent` = every launch reads the buffer the previous one wrote (in place)")
    # fraction of the fp32-MFMA peak from THIS file alone: algorithmic FLOPs per dense launch over the average node of the k_linear
    # launches (the fused RNN + tail launches cannot be timed as ad-hoc layers: they are priced at the timeline's figure below)
    out["frac_from_graph_nodes_k_linear_only"] = algo / n_all / (out["k_linear_avg_node_us"] * 1e-6) / 1e12 / PEAK
    if args.timeline and os.path.exists(args.timeline):
        rows = [r for r in csv.reader(l for l in open(args.timeline) if not l.startswith("#"))][1:]
        lin = [r for r in rows if r[1].startswith("k_linear")]
        rnn = [r for r in rows if r[1].startswith("k_rnn_tail")]
        slot = lambda rr: sum(float(r[6]) for r in rr if r[6])  # noqa: E731
        busy = lambda rr: sum(float(r[4]) for r in rr)  # noqa: E731
        out["timeline"] = dict(file=os.path.relpath(args.timeline, ROOT), k_linear_launches=len(lin), k_linear_slot_sum_us=slot(lin),
                               k_linear_busy_sum_us=busy(lin), k_rnn_tail_launches=len(rnn), k_rnn_tail_slot_sum_us=slot(rnn))
        out["graph_nodes_over_timeline_slots_k_linear"] = tot_plain / max(slot(lin), 1e-9)
        fam_us = tot_plain + slot(rnn)          # dense family = k_linear* (this file) + k_rnn_tail (timeline)
        out["frac_dense_family"] = algo / (fam_us * 1e-6) / 1e12 / PEAK
        out["frac_dense_family_note"] = ("algorithmic FLOPs of the step / (census x node time of the k_linear launches + the timeline's slots of the "
                                         "k_rnn_tail launches) / 157.3 TFLOP/s -- the same quantity as bench.py's roofline.frac")
    print(json.dumps({k: v for k, v in out.items() if k != "shapes"}, indent=1))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
