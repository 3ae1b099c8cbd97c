#!/bin/bash
# round 4, GPU call G: Gemma-2-2B - second tile requested inside the folded prologue
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4g; mkdir -p $O
timeout 900 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 "late-second-tile:LMRS_EARLY2=0" "early-second-tile" > $O/ab_gemma.txt 2>&1; cat $O/ab_gemma.txt
timeout 900 python tools/ab_bench.py --model gemma-2-2b --qtype q8_0 "late-second-tile:LMRS_EARLY2=0" "early-second-tile" > $O/ab_gemma_q8.txt 2>&1; cat $O/ab_gemma_q8.txt
timeout 300 python tools/timeline.py gemma-2-2b 30 q4_0 > $O/timeline_gemma2b_q4.txt 2>&1; tail -22 $O/timeline_gemma2b_q4.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "gemma or golden or random_geom" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
