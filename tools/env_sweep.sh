#!/bin/bash
# Runtime environment knobs that could change the cost of a dependent graph node: forward ms per pass at BASELINE configs[1]
run() { echo -n "$1 : "; env $1 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --train-steps 10 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); print(round(j['ms_per_step'],3), round(j['train']['ms_per_step'],3))"; }
run "SQ_NONE=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_INTERRUPT=0"
run "HIP_LAUNCH_BLOCKING=0 AMD_DIRECT_DISPATCH=1"
run "DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_USE_RUNTIME_UNBUNDLER=0"
