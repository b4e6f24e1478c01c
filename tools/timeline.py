#!/usr/bin/env python
"""Per-dispatch timeline of ONE forward step and ONE training step (no profiler attached): writes

    <out>/<tag>_timeline_fwd.csv, <out>/<tag>_timeline_train.csv   one row per dispatch: start, end, busy, gap before
    <out>/<tag>_timeline.json                                         sums, closure against the step time, dense / HBM fractions

    python tools/timeline.py [--cfg 2] [--batch 0] [--out gpurun_out] [--tag r03]      (on the GPU box; copy to profiles/)
    python tools/timeline.py --recompute profiles/r03_timeline_fwd.csv [--flops 43.34e9]   (anywhere: the fractions from a CSV)

How: sqair_amd/timeline.py — the library compiled with -DSQAIR_TIMELINE, every wave stamps start / end on the device wall clock.
A step is exactly what bench.py times: forward = noise draw + graph replay + ELBO kernel; training = noise draw + one graph
replay (forward with tape, VIMCO target, backward) + RMSProp + re-pack."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
MACS = {1: 10166288, 2: 13543744, 3: 13543744, 4: 20298656, 5: 27760960}  # per frame-particle, SURVEY.md Appendix D


def recompute(path, flops):
    from sqair_amd import timeline as TL
    rows = TL.read_csv(path)
    s = TL.summarise(rows)
    d = TL.dense_stats(rows)
    print("{}: {} dispatches, span {:.1f} us = busy {:.1f} + gaps {:.1f} - overlap {:.1f}".format(
        os.path.basename(path), s["dispatches"], s["span_us"], s["busy_us"], s["gap_us"], s["overlap_us"]))
    print("sum of slots (start-to-start) {:.1f} us".format(s["slot_sum_us"]))
    print("dense family (k_linear*, k_rnn_tail): {} launches, slot avg {:.3f} us, busy avg {:.3f} us".format(
        d["launches"], d["avg_slot_us"], d["avg_busy_us"]))
    if flops:
        per = flops / d["launches"]
        msg = "algorithmic {:.2f} MFLOP per launch -> frac (slot) {:.4f}".format(
            per / 1e6, per / (d["avg_slot_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
        if d["avg_busy_us"] > 0:
            msg += ", frac (busy only) {:.4f}".format(per / (d["avg_busy_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
        print(msg + ", whole step (sum of slots) {:.4f}".format(flops / (s["slot_sum_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS))
    for fam, f in sorted(s["families"].items(), key=lambda kv: -kv[1]["slot_us"]):
        print("  {:28s} n={:5d} slot {:9.1f} us (avg {:7.2f})  busy {:9.1f} us (avg {:7.2f})".format(
            fam, f["launches"], f["slot_us"], f["avg_slot_us"], f["busy_us"], f["avg_busy_us"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--recompute", default=None)
    ap.add_argument("--flops", type=float, default=2 * 5 * 13543744 * 320.0)
    ap.add_argument("--waves", default=None, help="also print wave lifetimes of the dispatches whose kernel name contains this")
    args = ap.parse_args()
    if args.recompute:
        return recompute(args.recompute, args.flops)

    import numpy as np
    import torch
    from sqair_amd import _capi
    from sqair_amd import timeline as TL
    from sqair_amd.data import config_inputs
    from sqair_amd.flags import make_flags
    from sqair_amd.params import init_params
    from sqair_amd.train import Trainer

    ov, obs, nums, _ = config_inputs(args.cfg, B=args.batch or None)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    T, B = int(obs.shape[0]), int(obs.shape[1])
    K, N = int(F.k_particles), int(F.n_steps_per_image)
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    os.makedirs(args.out, exist_ok=True)
    algo_flops = 2.0 * K * MACS[args.cfg] * B * T
    sfx = "" if args.cfg == 2 and not args.batch else "_cfg{}{}".format(args.cfg, "_b{}".format(args.batch) if args.batch else "")
    res = dict(build_id=_capi.build_id(), cfg=args.cfg, T=T, B=B, K=K, N=N, hw=list(hw), algorithmic_flops_per_step=algo_flops,
               peak_fp32_mfma_tflops=PEAK_FP32_MFMA_TFLOPS, command=" ".join(sys.argv),
               how="sqair_amd/timeline.py: the library compiled with -DSQAIR_TIMELINE, every wave stores {start, end} on the 100 MHz "
                   "device wall clock; nothing is rescaled")
    Ftr = make_flags(**dict(ov, learning_rate=1e-5, train_itr=1000000))

    def fwd_step(core):
        n = [0]

        def f():
            core.draw_noise(seed=1000, step=n[0], global_batch=B, b0=0)
            n[0] += 1
            core.forward(use_graph=True)
        return f

    core_p, model_p = TL.make_model(F, hw, P, obs, nums)
    tr_p = Trainer(model_p, Ftr, use_graph=True)
    ms_p = dict(fwd=TL.time_steps(core_p, fwd_step(core_p), steps=args.steps),
                train=TL.time_steps(core_p, lambda: tr_p.step(seed=2000, global_batch=B, b0=0), steps=args.steps))
    alg = {False: TL.algorithmic_hbm_bytes(T, B, K, N, hw[0], hw[1], G=int(F.glimpse_size), nh=core_p.nh, nw=core_p.nw,
                                           snh=core_p.snh, psnh=core_p.psnh, masked=bool(F.masked_glimpse), train=False)}
    alg[True] = TL.algorithmic_hbm_bytes(T, B, K, N, hw[0], hw[1], G=int(F.glimpse_size), nh=core_p.nh, nw=core_p.nw,
                                         snh=core_p.snh, psnh=core_p.psnh, masked=bool(F.masked_glimpse), train=True)
    del tr_p, model_p, core_p

    core_t, model_t = TL.make_model(F, hw, P, obs, nums, timeline=True)
    tl = TL.Timeline(core_t, mbytes=768 * max(1, B // 32))
    tr_t = Trainer(model_t, Ftr, use_graph=True)
    for name, step_t, train in (("fwd", fwd_step(core_t), False),
                                ("train", lambda: tr_t.step(seed=2000, global_batch=B, b0=0), True)):
        ms_t = TL.time_steps(core_t, step_t, steps=args.steps)
        rows, ev_ms = tl.measure(step_t, warm=2)
        s = TL.summarise(rows, ev_ms)
        d = TL.dense_stats(rows)
        path = os.path.join(args.out, "{}_timeline_{}{}.csv".format(args.tag, name, sfx))
        out = dict(product_ms_per_step=ms_p[name], timeline_build_ms_per_step=ms_t, overhead_vs_product=ms_t / ms_p[name] - 1.0,
                   this_step_hip_event_ms=ev_ms, dispatches=s["dispatches"], span_us=s["span_us"], busy_us=s["busy_us"],
                   gap_us=s["gap_us"], overlap_us=s["overlap_us"], slot_sum_us=s["slot_sum_us"],
                   closure_span_over_product_step=s["span_us"] / (ms_p[name] * 1e3),
                   closure_span_over_this_step=s["span_us"] / (ev_ms * 1e3),
                   dense=d, families=s["families"], hbm_class=TL.hbm_class(rows, alg[train]), csv=os.path.basename(path))
        if not train:
            per = algo_flops / max(d["launches"], 1)
            out["dense_frac"] = dict(
                algorithmic_flops_per_launch=per,
                frac_slot=per / (d["avg_slot_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                frac_busy_only=per / (d["avg_busy_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                frac_whole_step=algo_flops / (ms_p[name] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
        res[name] = out
        if args.waves:
            for r in rows:
                if args.waves in r["kernel"]:
                    print("  {} [{}]: busy {:.2f} us, {} workgroups / {} waves, wave lifetime p50 {:.2f} max {:.2f} us, starts spread "
                          "over {:.2f} us".format(r["kernel"], name, r["busy_us"], r["workgroups"], r["waves"], r["wave_life_p50_us"],
                                                  r["wave_life_max_us"], r["wave_start_spread_us"]))
        TL.write_csv(rows, path, "per-dispatch timeline of one {} step, cfg-{} (T={} B={} K={} N={} {}x{}), build {}\n"
                     "stamped by the kernels (100 MHz device wall clock): start = first wave's start, end = last wave's end, "
                     "gap_before = start - previous end, slot = next start - start\n"
                     "step time: product library {:.4f} ms, timeline build {:.4f} ms (HIP events, {} steps); this step "
                     "between events {:.4f} ms".format(name, args.cfg, T, B, K, N, hw[0], hw[1], res["build_id"],
                                                       ms_p[name], ms_t, args.steps, ev_ms))
        print("{}: product {:.3f} ms, timeline build {:.3f} ms ({:+.1f} %), this step {:.3f} ms; {} dispatches, span {:.1f} us = "
              "busy {:.1f} + gaps {:.1f} (- overlap {:.1f}); dense n={} avg slot {:.2f} us avg busy {:.2f} us".format(
                  name, ms_p[name], ms_t, 100 * (ms_t / ms_p[name] - 1), ev_ms, s["dispatches"], s["span_us"],
                  s["busy_us"], s["gap_us"], s["overlap_us"], d["launches"], d["avg_slot_us"], d["avg_busy_us"]))
    tl.close()
    with open(os.path.join(args.out, "{}_timeline{}.json".format(args.tag, sfx)), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
