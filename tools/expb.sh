cd /root/repo
B="python bench.py --no-cpu-baseline --train-steps 20 --steps 20 --warmup 3"
for i in 1 2; do
$B 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); print('ms_per_step', j['ms_per_step'], 'train', j['train']['ms_per_step'], 'nodes', j['train']['graph_nodes'])"
done
python -m pytest tests/test_hip_backward.py -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep "BAD\|^E \|passed\|failed\|FAILED" | cut -c1-200 | head -20
