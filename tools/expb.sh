cd /root/repo
B="python bench.py --no-cpu-baseline --train-steps 20 --steps 40 --warmup 5"
for i in 1 2; do
$B 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); print('ms_per_step', j['ms_per_step'], 'train', j['train']['ms_per_step'])"
done
python -m pytest tests/test_hip_backward.py -m gpu -q -x 2>&1 | grep "passed\|failed" | head -3
