#!/bin/bash
# round 4, GPU call Z: early second tile for three-pass gate/up launches (Gemma-2-2B Q4_0)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4z; mkdir -p $O
E=lm.rs_amd/liblmrs_hip_eb.so
timeout 500 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 new off:LMRS_EARLY_B=0@$E on@$E "off(again):LMRS_EARLY_B=0@$E" "on(again)@$E" > $O/ab_gemma_eb.txt 2>&1; cat $O/ab_gemma_eb.txt
timeout 500 python tools/ab_bench.py --model gemma-2-2b --qtype q8_0 off:LMRS_EARLY_B=0@$E on:LMRS_EARLY_B=1@$E > $O/ab_gemma_q8_eb.txt 2>&1; cat $O/ab_gemma_q8_eb.txt
