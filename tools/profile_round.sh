#!/bin/bash
# Produces the round's profile artefacts on the GPU box into gpurun_out/ (copy them to profiles/ afterwards):
#   rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (forward + training legs)
#   rNN_hbm_traffic.json   PMC FETCH_SIZE / WRITE_SIZE per kernel, two separate --pmc passes (eager: counter collection
#                          segfaults inside rocprofv3 when the step is a 2000-node graph replay)
#   rNN_profile_meta.json  build_id of the library the profiles were taken on + the commands
#   rNN_bench.json         the bench line of the same build (reads the files above through profiles/ once they are copied)
#   rNN_k_linear_device_clock.csv  per-launch device-clock stamps of the dense launches of one pass
# usage: gpurun -- 'bash tools/profile_round.sh r02'
R=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 20 --warmup 3 --train-steps 20 --no-cpu-baseline"
rm -rf /tmp/prof_s /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- $CMD > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name '*kernel_stats.csv' | head -1) $OUT/${R}_kernel_stats.csv
PCMD="python $REPO/bench.py --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline --no-graph"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $PCMD > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $PCMD > /tmp/pmc_w.log 2>&1
python $REPO/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $OUT/${R}_hbm_traffic.json > /dev/null
python - <<PY
import json, sys
sys.path.insert(0, "$REPO")
from sqair_amd._capi import build_id
json.dump(dict(build_id=build_id(), stats_command="rocprofv3 --kernel-trace --stats --output-format csv -- $CMD",
               pmc_command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- $PCMD"),
          open("$OUT/${R}_profile_meta.json", "w"), indent=1)
PY
# the bench line itself, reading the fresh profiles (copied into place for this run only if profiles/ lacks them)
cp $OUT/${R}_kernel_stats.csv $OUT/${R}_hbm_traffic.json $OUT/${R}_profile_meta.json $REPO/profiles/ 2>/dev/null
cd $REPO
SQAIR_PROF_DUMP=$OUT/${R}_k_linear_device_clock.csv python bench.py --steps 50 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
tail -c 1500 $OUT/${R}_bench.json
python tools/roofline_from_rocprof.py $OUT/${R}_kernel_stats.csv
python tools/layer_clock.py $OUT/${R}_k_linear_device_clock.csv | tail -1
