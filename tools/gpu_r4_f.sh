#!/bin/bash
# round 4, GPU call F: Gemma-2-2B Q4_0 - passes per workgroup of the gate/up launch
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4f; mkdir -p $O
timeout 900 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 "glu-2-passes(288wg):LMRS_GLU_PASSES=2" "glu-3-passes(192wg)" "glu-4-passes(144wg):LMRS_GLU_PASSES=4" "no-chain-spread:LMRS_CHAIN_SPREAD=0" > $O/ab_gemma.txt 2>&1; cat $O/ab_gemma.txt
timeout 900 python tools/ab_bench.py --model gemma-2-2b --qtype q8_0 "glu-2-passes:LMRS_GLU_PASSES=2" "default" > $O/ab_gemma_q8.txt 2>&1; cat $O/ab_gemma_q8.txt
