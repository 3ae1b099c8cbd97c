#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp; O=gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python bench.py --model gemma-2-2b --qtype q4_0 --steps 4 --cpu-steps 0 > /dev/null 2>&1
try() { name=$1; shift; rm -rf $O/$name; ( "$@" ) > $O/$name.log 2>&1; echo "$name: rc $? $(ls $O/$name/*/*kernel_stats.csv 2>/dev/null | head -1)"; }
try a_trace_only env LMRS_STEPS_PER_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/a_trace_only -- python bench.py --model gemma-2-2b --qtype q4_0 --steps 16 --cpu-steps 0
try b_separate env LMRS_STEPS_PER_GRAPH=1 LMRS_QKV_ATT=0 LMRS_CLS_TAIL=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b_separate -- python bench.py --model gemma-2-2b --qtype q4_0 --steps 16 --cpu-steps 0
try c_noprefetch env LMRS_STEPS_PER_GRAPH=1 HSA_ENABLE_SDMA=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_noprefetch -- python bench.py --model gemma-2-2b --qtype q4_0 --steps 16 --cpu-steps 0
try d_ab env timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d_ab -- python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 --child /tmp/gemma-2-2b_q4_0_seed1234.lmrs
ls $O/*/* 2>/dev/null | head; grep -l "SIGSEGV" $O/*.log
