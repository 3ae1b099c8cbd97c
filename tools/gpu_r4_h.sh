#!/bin/bash
# round 4, GPU call H: Gemma-2-2B Q4_0 - key warm-up while the attention workgroups wait, passes per gate/up workgroup with the early second tile
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4h; mkdir -p $O
timeout 900 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 "no-warm,3-passes:LMRS_ATT_KWARM=0" "warm,3-passes" "warm,2-passes+early:LMRS_GLU_PASSES=2" "warm,2-passes,late:LMRS_GLU_PASSES=2,LMRS_EARLY2=0" > $O/ab_gemma.txt 2>&1; cat $O/ab_gemma.txt
timeout 300 python tools/timeline.py gemma-2-2b 30 q4_0 > $O/timeline_gemma2b_q4.txt 2>&1; tail -22 $O/timeline_gemma2b_q4.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "gemma or golden or random_geom or merged" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
