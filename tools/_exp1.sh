set -x
python -m pytest tests/test_hip_kernels.py tests/test_hip_forward.py -x -q 2>&1 | tail -5
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 2 --train-steps 3 --no-cpu-baseline --streams 0 > /tmp/kt.log 2>&1
tail -c 600 /tmp/kt.log
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1); ls -la $f; head -3 $f
mkdir -p $R/gpurun_out/exp1; cp $f $R/gpurun_out/exp1/kernel_trace.csv; cp /tmp/kt.log $R/gpurun_out/exp1/
