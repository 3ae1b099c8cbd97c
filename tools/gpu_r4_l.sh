#!/bin/bash
# round 4, GPU call L: wo + w1/w3 as one launch (prototype of a persistent all-to-all edge)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4l; mkdir -p $O
timeout 600 python tools/ab_bench.py "separate" "wo+w13,both-tiles-early:LMRS_WO_W13=1" "wo+w13,first-tile-early:LMRS_WO_W13=1,LMRS_WO_W13_TILES=1" "separate(again)" > $O/ab.txt 2>&1; cat $O/ab.txt
LMRS_WO_W13=1 timeout 300 python tools/timeline.py llama-3.2-1b 100 > $O/timeline_wo13.txt 2>&1; tail -24 $O/timeline_wo13.txt
LMRS_WO_W13=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "greedy_token_ids or merged or mini_q4_and or golden" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
