#!/bin/bash
# round 4, GPU call M: wo + w1/w3 as one launch, gate/up tile requests delayed behind the wo rows
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4m; mkdir -p $O
timeout 900 python tools/ab_bench.py "separate" "wo+w13,sleep=2:LMRS_WO_W13=1,LMRS_WO_W13_SLEEP=2" "wo+w13,sleep=3:LMRS_WO_W13=1,LMRS_WO_W13_SLEEP=3" "wo+w13,sleep=5:LMRS_WO_W13=1,LMRS_WO_W13_SLEEP=5" "wo+w13,sleep=3,one-tile:LMRS_WO_W13=1,LMRS_WO_W13_SLEEP=3,LMRS_WO_W13_TILES=1" "separate(again)" > $O/ab.txt 2>&1; cat $O/ab.txt
LMRS_WO_W13=1 timeout 300 python tools/timeline.py llama-3.2-1b 100 > $O/timeline_wo13.txt 2>&1; tail -24 $O/timeline_wo13.txt
