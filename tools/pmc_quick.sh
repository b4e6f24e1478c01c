#!/bin/bash
# PMC FETCH_SIZE / WRITE_SIZE of one configuration (two separate --pmc passes, eager launches): gpurun_out/<tag>_hbm_traffic<sfx>.json
# usage: bash tools/pmc_quick.sh <tag> <cfg>
TAG=${1:-x}; CFG=${2:-2}
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
SFX=""; [ "$CFG" != "2" ] && SFX="_cfg$CFG"
PCMD="python $REPO/bench.py --cfg $CFG --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline --no-graph --no-timeline --streams 0"
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $PCMD > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $PCMD > /tmp/pmc_w.log 2>&1
mkdir -p $REPO/gpurun_out
python $REPO/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $REPO/gpurun_out/${TAG}_hbm_traffic${SFX}.json > /dev/null
python - <<PY
import json
j = json.load(open("$REPO/gpurun_out/${TAG}_hbm_traffic${SFX}.json"))
for k, v in sorted(j["family_bytes_per_launch"].items()):
    if k.split("<")[0] in ("k_crop_row", "k_insert_loglik", "k_compact", "k_logprob", "k_crop_chain_bwd", "k_insert_loglik_bwd", "k_logprob_bwd", "k_compact_bwd", "k_wgrad_group", "k_linear_big", "k_linear_mt"):
        print("%-24s %10.0f bytes / launch" % (k, v))
print("dominant", j["dominant"], j["dominant_bytes_per_launch"])
PY
