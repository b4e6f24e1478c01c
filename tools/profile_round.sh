#!/bin/bash
# Produces the round's profile artefacts on the GPU box into gpurun_out/<R>/ (copy them to profiles/ afterwards):
#   rNN_timeline_fwd.csv, rNN_timeline_train.csv, rNN_timeline.json
#                          per-dispatch timeline of ONE forward step and ONE training step, stamped by the kernels themselves
#                          (tools/timeline.py, sqair_amd/timeline.py): start / end / busy / gap per dispatch; sums to the step
#   rNN_timeline*_cfg{4,5}*  the same for BASELINE configs 4 and 5
#   rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (forward + training legs).  NOTE: under the
#                          profiler the forward step takes ~5.2 ms instead of 3.5 ms (every dispatch is intercepted), its per-dispatch
#                          durations include the dependent boundary; kept as the contract's profiler-side cross-check
#   rNN_rocprof_trace_one_step.csv  rocprofv3 --kernel-trace begin / end timestamps of one replayed forward step + one training step
#   rNN_hbm_traffic.json   PMC FETCH_SIZE / WRITE_SIZE per kernel, two separate --pmc passes (eager: counter collection
#                          segfaults inside rocprofv3 when the step is a 2000-node graph replay)
#   rNN_profile_meta.json  build_id of the library the profiles were taken on + the commands
#   rNN_bench.json, rNN_bench_cfg{4,5}.json   bench lines of the same build WITH cpu_baseline
# usage: gpurun --timeout 1800 -- 'bash tools/profile_round.sh r06'
R=${1:-r06}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $REPO/tools/timeline.py --cfg 2 --out $OUT --tag $R 2>&1 | grep -v amdgpu.ids
python $REPO/tools/timeline.py --cfg 4 --out $OUT --tag $R 2>&1 | grep -v amdgpu.ids
python $REPO/tools/timeline.py --cfg 5 --out $OUT --tag $R 2>&1 | grep -v amdgpu.ids
CMD="python $REPO/bench.py --steps 20 --warmup 3 --train-steps 20 --no-cpu-baseline --no-timeline --streams 0"
rm -rf /tmp/prof_s /tmp/pmc_f /tmp/pmc_w /tmp/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- $CMD > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name '*kernel_stats.csv' | head -1) $OUT/${R}_kernel_stats.csv
grep '^{"metric"' /tmp/prof_s.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line under rocprofv3: forward', d['ms_per_step'], 'ms, training', d['train']['ms_per_step'], 'ms')"
python - <<PY
import csv, sys
sys.path.insert(0, "$REPO")
# one replayed forward step and one training step out of the kernel trace (begin / end timestamps per dispatch)
import glob
f = glob.glob("/tmp/prof_s/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_fill_noise")]
segs = [(a, b) for a, b in zip(idx[:-1], idx[1:])]
fwd = [s for s in segs if 800 < s[1] - s[0] < 1000]
trn = [s for s in segs if 1700 < s[1] - s[0] < 2100]
with open("$OUT/${R}_rocprof_trace_one_step.csv", "w") as fh:
    fh.write("# rocprofv3 --kernel-trace begin/end of one replayed forward step and one training step (profiler attached: slower than the product run)\n")
    fh.write("leg,idx,kernel,start_ns,end_ns,duration_ns,gap_before_ns\n")
    for leg, ss in (("fwd", fwd), ("train", trn)):
        if not ss:
            continue
        a, b = ss[len(ss) // 2]
        t0, prev = ev[a][0], None
        for i, e in enumerate(ev[a:b]):
            fh.write("{},{},\"{}\",{},{},{},{}\n".format(leg, i, e[2], e[0] - t0, e[1] - t0, e[1] - e[0], 0 if prev is None else e[0] - prev))
            prev = e[1]
        span = ev[b - 1][1] - ev[a][0]
        busy = sum(e[1] - e[0] for e in ev[a:b])
        print(leg, "under rocprofv3: span %.1f us, sum of durations %.1f us" % (span / 1e3, busy / 1e3))
PY
PCMD="python $REPO/bench.py --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline --no-graph --no-timeline --streams 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $PCMD > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $PCMD > /tmp/pmc_w.log 2>&1
python $REPO/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $OUT/${R}_hbm_traffic.json > /dev/null
# the same two counter passes for BASELINE configs 4 and 5 (rNN_hbm_traffic_cfg{4,5}.json)
for C in 4 5; do
  PC="python $REPO/bench.py --cfg $C --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline --no-graph --no-timeline --streams 0"
  rm -rf /tmp/pmc_f$C /tmp/pmc_w$C
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f$C -- $PC > /tmp/pmc_f$C.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w$C -- $PC > /tmp/pmc_w$C.log 2>&1
  python $REPO/tools/pmc_traffic.py /tmp/pmc_f$C /tmp/pmc_w$C $OUT/${R}_hbm_traffic_cfg$C.json > /dev/null
done
python - <<PY
import json, sys
sys.path.insert(0, "$REPO")
from sqair_amd._capi import build_id
json.dump(dict(build_id=build_id(), stats_command="rocprofv3 --kernel-trace --stats --output-format csv -- $CMD",
               pmc_command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- $PCMD",
               timeline_command="python tools/timeline.py --cfg 2|4|5"),
          open("$OUT/${R}_profile_meta.json", "w"), indent=1)
PY
# the bench lines themselves, reading the fresh profiles (copied into place for this run)
cp $OUT/${R}_hbm_traffic*.json $OUT/${R}_timeline*.json $OUT/${R}_timeline_*.csv $REPO/profiles/ 2>/dev/null
cd $REPO
python bench.py --steps 50 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
python bench.py --cfg 4 --steps 30 --warmup 5 > $OUT/${R}_bench_cfg4.json 2> $OUT/${R}_bench_cfg4.err
python bench.py --cfg 5 --steps 30 --warmup 5 > $OUT/${R}_bench_cfg5.json 2> $OUT/${R}_bench_cfg5.err
python bench.py --batch 256 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${R}_bench_b256.json 2> $OUT/${R}_bench_b256.err
python - <<PY
import json
for n in ("", "_cfg4", "_cfg5", "_b256"):
    try:
        d = json.load(open("$OUT/${R}_bench%s.json" % n))
        print(n or "cfg2", "fwd %.3f ms %.0f f/s | train %.3f ms | frac_slot %.4f busy %.4f whole %.4f | cpu %s" % (
            d["ms_per_step"], d["value"], d["train"]["ms_per_step"], d["roofline"]["frac_slot"], d["roofline"]["frac_busy_only"],
            d["roofline"]["frac_whole_step"], (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e:
        print(n, "failed", e)
PY
