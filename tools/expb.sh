cd /root/repo
B="python bench.py --no-cpu-baseline --train-steps 20 --steps 20 --warmup 3"
for i in 1 2; do
$B 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); print('ms_per_step', j['ms_per_step'], 'train', j['train']['ms_per_step'], 'nodes', j['train']['graph_nodes'])"
done
python -m pytest tests/test_hip_backward.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep "BAD\|^E \|passed\|failed\|FAILED" | cut -c1-200 | head -20
for b in 256; do python bench.py --no-cpu-baseline --train-steps 5 --steps 10 --warmup 2 --batch $b 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); r=j['roofline']; print('batch $b fwd ms', j['ms_per_step'], 'frames/s', j['value'], 'train ms', j['train']['ms_per_step'], 'frac_hip_events', r['frac_hip_events'], 'frac_exec_clock', r['frac_executed_device_clock'], 'frac_clock', r['frac_device_clock'])"; done
python bench.py --no-cpu-baseline --train-steps 5 --steps 10 --warmup 2 --cfg 5 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); r=j['roofline']; print('cfg5 fwd ms', j['ms_per_step'], 'frames/s', j['value'], 'train ms', j['train']['ms_per_step'], 'frac_hip_events', r['frac_hip_events'], 'frac_clock', r['frac_device_clock'])"
