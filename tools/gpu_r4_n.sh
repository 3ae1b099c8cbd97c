#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4n; mkdir -p $O
timeout 900 python tools/ab_bench.py "4-per-graph" "5-per-graph:LMRS_STEPS_PER_GRAPH=5" "10-per-graph:LMRS_STEPS_PER_GRAPH=10" "20-per-graph:LMRS_STEPS_PER_GRAPH=20" "2-per-graph:LMRS_STEPS_PER_GRAPH=2" "4-per-graph(again)" > $O/ab.txt 2>&1; cat $O/ab.txt
