#!/bin/bash
# round 4, GPU call U: L2 run-ahead of the gate/up stream from spare workgroups of the merged qkv + attention launch (LMRS_PF_*)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4u; mkdir -p $O
A="LMRS_PF_MB=16.8,LMRS_PF_PASS=0,LMRS_PF_EARLY_B=1,LMRS_PF_BLOCKS=128"
timeout 900 python tools/ab_bench.py new earlyb-only:LMRS_PF_EARLY_B=1 A-d0:$A A-d2:$A,LMRS_PF_DELAY=2 A-d3:$A,LMRS_PF_DELAY=3 A-d4:$A,LMRS_PF_DELAY=4 \
  A-b64-d3:$A,LMRS_PF_BLOCKS=64,LMRS_PF_DELAY=3 A-noearly-d3:LMRS_PF_MB=16.8,LMRS_PF_PASS=0,LMRS_PF_BLOCKS=128,LMRS_PF_DELAY=3 \
  B16-d3:LMRS_PF_MB=16.8,LMRS_PF_BLOCKS=128,LMRS_PF_DELAY=3 B8-d3:LMRS_PF_MB=8,LMRS_PF_BLOCKS=128,LMRS_PF_DELAY=3 A12-d3:$A,LMRS_PF_MB=12,LMRS_PF_DELAY=3 "new(again)" > $O/ab_pf2.txt 2>&1; cat $O/ab_pf2.txt
A="LMRS_PF_MB=16.8 LMRS_PF_PASS=0 LMRS_PF_EARLY_B=1 LMRS_PF_BLOCKS=128 LMRS_PF_DELAY=3"
env $A timeout 200 python tools/timeline.py llama-3.2-1b 30 > $O/tl_A_d3.txt 2>&1
grep -A8 "mean per kernel" $O/tl_A_d3.txt; grep "step span" $O/tl_A_d3.txt
