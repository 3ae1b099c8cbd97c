#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/make_profiles.sh r4 extra > gpurun_out/make_profiles_r4_extra.log 2>&1; tail -6 gpurun_out/make_profiles_r4_extra.log
export LMRS_BENCH_IMAGE_CACHE=/tmp
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps:', d['value'], 'tok/s', d['ms_per_step'], 'ms; 128:', d['value_128_steps'])"; done
ls gpurun_out/art | grep -E "stats|traffic"
