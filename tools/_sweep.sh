mkdir -p gpurun_out/art
for k in 4 1 2 5 10 20 4; do
  for r in 1 2; do
    LMRS_STEPS_PER_GRAPH=$k timeout 60 python bench.py --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k', d['value'], d.get('value_128_steps'))"
  done
done
timeout 100 python bench.py --steps 128 --warmup 16 --cpu-steps 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('128:', d['value'], d['parity'])"
