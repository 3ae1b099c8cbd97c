#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/ab_bench.py "default" > gpurun_out/r4o_ab.txt 2>&1; cat gpurun_out/r4o_ab.txt
timeout 300 python tools/first_run.py > gpurun_out/r4o_first.txt 2>&1; cat gpurun_out/r4o_first.txt
LMRS_BENCH_IMAGE_CACHE=/tmp timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py 20 steps:', d['value'], 'tok/s', d['ms_per_step'], 'ms; device events', d['roofline']['path']['device_event_us_per_step'])"
