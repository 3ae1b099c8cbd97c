#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp; O=gpurun_out/r4r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python bench.py --model gemma-2-2b --qtype q4_0 --steps 4 --cpu-steps 0 > /dev/null 2>&1
LMRS_AQL=1 LMRS_AQL_HOST_KERNARG=1 LMRS_AQL_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python bench.py --model gemma-2-2b --qtype q4_0 --steps 64 --cpu-steps 0 > $O/st.log 2>&1; echo "rc $?"
ls $O/st/*/ 2>/dev/null; head -8 $O/st/*/*kernel_stats.csv 2>/dev/null | cut -c1-200; grep -v "^W2026\|^E2026\|^I2026" $O/st.log | tail -5
