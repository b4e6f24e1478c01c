cd /root/repo
for m in 0 1 2 3 5 6 7; do SQAIR_EMIT_EXTRA=$m python bench.py --no-cpu-baseline --train-steps 0 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; j=json.load(sys.stdin); r=j['roofline']; print('extra=$m graph_ms', r['dense_only_graph_ms'], 'nodes', r['avg_launch_us'] and round(r['dense_only_graph_ms']*1e3/r['avg_launch_us']))"; done
