"""Per-dispatch timeline of a step, measured by the kernels themselves (no profiler attached).

``libsqair_hip_timeline.so`` is the product library's source compiled with ``-DSQAIR_TIMELINE`` (csrc/sqair_common.h): every
wave of every kernel stores {start, end} of its life on the 100 MHz device wall clock (``s_memrealtime``, one counter for the
whole chip) into its own 16-byte slot — plain stores, nothing shared.  Reduced per dispatch this gives first-wave start and
last-wave end: **busy** time of the kernel; the difference between one dispatch's end and the next one's start is the **gap**
(the dependent launch boundary: end-of-kernel release, the next dispatch's launch, its first wave's arrival).  By
construction  sum(busy) + sum(gap) = last end - first start = the step, which is checked against a HIP-event pair around the
same step and against the product library's step time.

This is measurement plumbing for ``bench.py`` (``roofline``) and ``tools/timeline.py``; the product path never loads it.
The reference has no counterpart (its only timer is wall-clock around eval loops, sqair/eval_tools.py:344,363)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi

TICK_US = 0.01  # s_memrealtime counts at 100 MHz

DENSE_PREFIXES = ("k_linear", "k_rnn_tail")            # fp32-MFMA dense layers (forward) incl. the fused slot tail
DENSE_BWD_PREFIXES = ("k_linear_dx", "k_wgrad")        # their adjoints


def family(kernel):
    """Kernel name without template arguments / parentheses: `(k_linear_mt<4, 1, 2, true>)` -> `k_linear_mt`."""
    k = kernel.strip("() ")
    k = k.split("<")[0].strip()
    return _FAMILY_ALIAS.get(k, k)


# formulations of one operation chosen by the frame size: one family in the summaries (the CSVs keep the kernel's own name)
_FAMILY_ALIAS = {"k_insert_loglik_rows": "k_insert_loglik", "k_insert_loglik_bwd_rows": "k_insert_loglik_bwd"}


class Timeline(object):
    """Owns the stamp buffer of a ``SqairCore`` built on the timeline library and turns it into per-dispatch rows.
    Create it BEFORE the core captures a graph (the slot addresses are kernel arguments frozen at capture time)."""

    def __init__(self, core, mbytes=768):
        """`mbytes`: the stamp buffer (16 bytes per wave of every dispatch issued while recording, eager launches included: the
        headline configuration needs ~30 MB per training step; scale it with the batch)."""
        if core.lib.sqair_timeline_available() != 1:
            raise RuntimeError("Timeline needs a SqairCore(lib_path=_capi.TIMELINE_LIB_PATH)")
        self.core = core
        with torch.cuda.device(core.device):
            self.buf = torch.zeros((mbytes << 20) // 8, dtype=torch.int64, device=core.device)
        core.check(core.lib.sqair_timeline_begin(core.handle, self.buf.data_ptr(), self.buf.numel() * 8),
                    "sqair_timeline_begin")

    def close(self):
        self.core.lib.sqair_timeline_end(self.core.handle)

    def records(self):
        lib, h = self.core.lib, self.core.handle
        n = lib.sqair_timeline_count(h)
        if n == -4:
            raise RuntimeError("the timeline's stamp buffer ({} MB) overflowed: later dispatches carry no stamps; "
                               "give Timeline(core, mbytes=...) more".format(self.buf.numel() * 8 >> 20))
        out = []
        name, off, waves, wgs = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
        for i in range(n):
            lib.sqair_timeline_record(h, i, C.byref(name), C.byref(off), C.byref(waves), C.byref(wgs))
            out.append((name.value.decode(), int(off.value), int(waves.value), int(wgs.value)))
        return out

    def measure(self, step_fn, warm=3):
        """Runs `step_fn` (everything it issues must go to core.stream) `warm` times, idles the device for 3 ms, then runs it
        once more between a HIP-event pair.  Returns (rows, event_ms): one row per dispatch of THAT step (selected by time:
        everything stamped within the step's duration before the last stamp; the idle gap separates it from older stamps),
        ordered by start: dict(kernel, family, start_us, end_us, busy_us, gap_us (to the previous end), slot_us (to the next
        start), workgroups, waves)."""
        import time
        core = self.core
        with core.on_stream():
            for _ in range(warm):
                step_fn()
            core.stream.synchronize()
            time.sleep(0.003)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(core.stream)
            step_fn()
            e1.record(core.stream)
            core.stream.synchronize()
        event_ms = e0.elapsed_time(e1)
        recs = self.records()
        used = max((off + 2 * waves for _, off, waves, _ in recs), default=0)
        raw = self.buf[:used].cpu().numpy().view(np.uint64)
        rows = []
        for kernel, off, waves, wgs in recs:
            pairs = raw[off:off + 2 * waves].reshape(-1, 2)
            live = pairs[:, 0] != 0
            if not live.any():
                continue
            life = (pairs[live, 1] - pairs[live, 0]).astype(np.float64) * TICK_US
            rows.append(dict(kernel=kernel.strip("() "), family=family(kernel), start=int(pairs[live, 0].min()),
                             end=int(pairs[live, 1].max()), workgroups=wgs, waves=waves,
                             # (not in the CSV) lifetimes of the waves and the spread of their starts: lifetimes ~ busy = one
                             # latency-bound round of waves; lifetimes << busy = several rounds / throughput-bound
                             wave_life_p50_us=float(np.median(life)), wave_life_max_us=float(life.max()),
                             wave_start_spread_us=float((pairs[live, 0].max() - pairs[live, 0].min()) * TICK_US)))
        t_last = max(r["end"] for r in rows)
        lo = t_last - int((event_ms * 1e3 + 1500.0) / TICK_US)
        rows = [r for r in rows if r["start"] >= lo]
        rows.sort(key=lambda r: r["start"])
        t0 = rows[0]["start"] if rows else 0
        prev_end = None
        for i, r in enumerate(rows):
            r["start_us"] = (r["start"] - t0) * TICK_US
            r["end_us"] = (r["end"] - t0) * TICK_US
            r["busy_us"] = r["end_us"] - r["start_us"]
            r["gap_us"] = 0.0 if prev_end is None else r["start_us"] - prev_end
            prev_end = r["end_us"] if prev_end is None else max(prev_end, r["end_us"])
            nxt = rows[i + 1]["start"] if i + 1 < len(rows) else r["end"]
            r["slot_us"] = (nxt - r["start"]) * TICK_US
        for r in rows:
            del r["start"], r["end"]
        return rows, event_ms


def summarise(rows, event_ms=None):
    """Busy / gap sums per kernel family and for the whole step.  `span_us` = last end - first start = sum(busy) + sum(gap)
    when dispatches do not overlap (overlap, if any, is reported)."""
    fams = {}
    for r in rows:
        f = fams.setdefault(r["family"], dict(launches=0, busy_us=0.0, gap_before_us=0.0, slot_us=0.0))
        f["launches"] += 1
        f["busy_us"] += r["busy_us"]
        f["gap_before_us"] += r["gap_us"]
        f["slot_us"] += r.get("slot_us", 0.0)
    for f in fams.values():
        f["avg_busy_us"] = f["busy_us"] / f["launches"]
        f["avg_slot_us"] = f["slot_us"] / f["launches"]
    span = max(r["end_us"] for r in rows) - min(r["start_us"] for r in rows) if rows else 0.0
    busy = sum(r["busy_us"] for r in rows)
    gap = sum(max(r["gap_us"], 0.0) for r in rows)
    overlap = -sum(min(r["gap_us"], 0.0) for r in rows)
    out = dict(dispatches=len(rows), span_us=span, busy_us=busy, gap_us=gap, overlap_us=overlap,
               slot_sum_us=sum(r.get("slot_us", 0.0) for r in rows), families=fams)
    if event_ms is not None:
        out["hip_event_ms"] = event_ms
    return out


def dense_stats(rows, prefixes=DENSE_PREFIXES, exclude=DENSE_BWD_PREFIXES):
    sel = [r for r in rows if r["family"].startswith(prefixes) and not r["family"].startswith(exclude)]
    n = len(sel)
    busy = sum(r["busy_us"] for r in sel)
    slot = sum(r.get("slot_us", 0.0) for r in sel)
    return dict(launches=n, busy_us=busy, slot_us=slot, avg_busy_us=busy / max(n, 1), avg_slot_us=slot / max(n, 1))


def write_csv(rows, path, header_comment=None):
    with open(path, "w") as fh:
        if header_comment:
            for line in header_comment.splitlines():
                fh.write("# " + line + "\n")
        fh.write("idx,kernel,start_us,end_us,busy_us,gap_before_us,slot_us,workgroups,waves\n")
        for i, r in enumerate(rows):
            fh.write('{},"{}",{:.2f},{:.2f},{:.2f},{:.2f},{:.2f},{},{}\n'.format(
                i, r["kernel"], r["start_us"], r["end_us"], r["busy_us"], r["gap_us"], r["slot_us"], r["workgroups"], r["waves"]))


def read_csv(path):
    import csv
    rows = []
    with open(path) as fh:
        lines = [l for l in fh if not l.startswith("#")]
    for r in csv.DictReader(lines):
        rows.append(dict(kernel=r["kernel"], family=family(r["kernel"]), start_us=float(r["start_us"]), end_us=float(r["end_us"]),
                         busy_us=float(r["busy_us"]), gap_us=float(r["gap_before_us"]), slot_us=float(r["slot_us"]),
                         workgroups=int(r["workgroups"]),
                         waves=int(r["waves"])))
    return rows


# ---------------------------------------------------------------------------------------------------------------------
# Algorithmic HBM bytes of the gather / scatter / reduce class (SURVEY.md 8(d) "reported separately"):
# the DISTINCT bytes a kernel family has to read and write once per STEP, from the shapes alone.  achieved = bytes / busy time
# of the family in the timeline; peak = 8 TB/s (MI355X_MICROARCH.md).
# ---------------------------------------------------------------------------------------------------------------------
HBM_PEAK_TBS = 8.0


def algorithmic_hbm_bytes(T, B, K, N, H, W, G=20, nh=256, nw=50, snh=None, psnh=None, masked=True, train=False, canvas=False):
    """Per-step totals per kernel family (bytes).  R = B K particle rows, P = H W, G2 = G G, record = 168 floats."""
    R, P, G2 = B * K, H * W, G * G
    snh = snh or nh
    psnh = psnh or nh
    f = 4
    out = {}
    # k_crop_row per frame: 1 batched launch for crop #1 of all N slots + N launches for crop #2 + N for discovery; each launch
    # reads the B frames once; per (row, slot): writes the G2 glimpse, reads the mask (propagation crops), the transform's hidden
    # layer (nh: its 8-wide output layer is evaluated in the launch; crop #1 reads the 4-wide where-bias instead), the previous
    # where (4), the noise (4) and writes where / loc / scale (12)
    per_rs = lambda mask, hid: f * (G2 + (G2 if mask else 0) + hid + 4 + 4 + 12)
    out["k_crop_row"] = T * ((1 + 2 * N) * B * P * f + R * N * (per_rs(masked, 4) + per_rs(masked, nh) + per_rs(False, nh)))
    # k_insert_loglik, once per pass over all T frames: glimpses, (where 4, presence 1) per slot, the frames, the mean image;
    # writes data_ll + 5 per-row scalars (+ the canvas when requested)
    out["k_insert_loglik"] = f * (T * R * N * (G2 + 5) + T * B * P + P + T * R * 6 + (T * R * P if canvas else 0))
    # k_compact per frame: reads the 2N candidate records (168) + temporal and prior states of the N propagated slots, writes the
    # N surviving records + states, ids
    out["k_compact"] = T * f * R * N * (2 * 168 + snh + psnh + 168 + snh + psnh + 2)
    # k_logprob once per pass: the 2N records + previous records + prior statistics (112) + conditioning (128 / row); ~40 scalars out
    out["k_logprob"] = T * f * R * (N * (3 * 168 + 112) + 128 + 40)
    if train:
        out["k_insert_loglik_bwd"] = f * (T * R * N * (2 * G2 + 5 + 4) + T * B * P + 2 * P + T * R * 2)
        out["k_logprob_bwd"] = T * f * R * (N * (3 * 168 + 112 + 3 * 168 + 112) + 2 * 128 + 8)
        out["k_compact_bwd"] = out["k_compact"]
        # adjoint of the crops: per (row, slot) reads the glimpse gradient (+ mask and glimpse for the mask gradient), the frame;
        # writes d where (+ d mask)
        out["k_crop_chain_bwd"] = T * ((1 + 2 * N) * B * P * f + R * N * f * (2 * (3 * G2 + nh + 24) + (G2 + nh + 24)))
    return out


def hbm_class(rows, alg_bytes, traffic_per_launch=None):
    """Per family: launches, busy time, algorithmic bytes per step -> achieved TB/s and fraction of the 8 TB/s HBM peak.
    `traffic_per_launch`: optional {family: PMC bytes per launch} measured by rocprofv3 --pmc (profiles/rNN_hbm_traffic.json,
    `family_bytes_per_launch`), quoted beside it as bytes per step."""
    s = summarise(rows)["families"]
    out = []
    for fam, b in alg_bytes.items():
        if fam not in s:
            continue
        busy = s[fam]["busy_us"]
        e = dict(kernel=fam, launches=s[fam]["launches"], busy_us=busy, avg_busy_us=s[fam]["avg_busy_us"],
                 algorithmic_bytes_per_step=b, achieved=b / (busy * 1e-6) / 1e12, peak=HBM_PEAK_TBS, unit="TB/s")
        e["frac"] = e["achieved"] / HBM_PEAK_TBS
        if traffic_per_launch and fam in traffic_per_launch:
            e["traffic_bytes_per_step"] = traffic_per_launch[fam] * s[fam]["launches"]
            e["traffic_over_algorithmic"] = e["traffic_bytes_per_step"] / b
        out.append(e)
    return out


def make_model(F, hw, P, obs, nums, timeline=False, device="cuda:0", outputs="minimal"):
    """(core, Model) on the product library or on the timeline library (same parameters and frames)."""
    from .model import Model, SqairCore
    core = SqairCore(F, hw, device=device, lib_path=_capi.TIMELINE_LIB_PATH if timeline else None)
    with core.on_stream():
        core.set_params(P)
        m = Model(obs, None, core, int(F.k_particles), presence=nums, outputs=outputs)
    return core, m


def time_steps(core, step_fn, steps=20, warm=3):
    with core.on_stream():
        for _ in range(warm):
            step_fn()
        core.stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(core.stream)
        for _ in range(steps):
            step_fn()
        e1.record(core.stream)
        core.stream.synchronize()
    return e0.elapsed_time(e1) / steps
