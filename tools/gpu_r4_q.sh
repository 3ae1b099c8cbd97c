#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp; O=gpurun_out/r4q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python bench.py --model gemma-2-2b --qtype q4_0 --steps 4 --cpu-steps 0 > /dev/null 2>&1
try() { name=$1; shift; rm -rf $O/$name; ( "$@" ) > $O/$name.log 2>&1; echo "$name: rc $? $(ls $O/$name/*/*kernel_stats.csv 2>/dev/null | head -1)"; }
try aql env LMRS_AQL=1 LMRS_AQL_VERBOSE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/aql -- python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 --child /tmp/gemma-2-2b_q4_0_seed1234.lmrs
head -5 $O/aql/*/*kernel_stats.csv 2>/dev/null
