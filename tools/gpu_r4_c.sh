#!/bin/bash
# round 4, GPU call C: the decode step as hand-written AQL packets against graph replays; sleep knobs of the three-part launch
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c; mkdir -p $O
export LMRS_AQL_VERBOSE=1
timeout 900 python tools/ab_bench.py "graph:LMRS_WO_MERGED=0" "aql:LMRS_WO_MERGED=0,LMRS_AQL=1" "aql-nofence:LMRS_WO_MERGED=0,LMRS_AQL=1,LMRS_AQL_FENCE=0" \
  "aql+wo:LMRS_AQL=1" "aql+wo-s0=4:LMRS_AQL=1,LMRS_WO_SLEEP0=4" "aql+wo-s0=6,s1=2:LMRS_AQL=1,LMRS_WO_SLEEP0=6,LMRS_WO_SLEEP1=2" "aql+wo-s0=8,s1=0:LMRS_AQL=1,LMRS_WO_SLEEP0=8,LMRS_WO_SLEEP1=0" \
  "graph+wo-s0=6,s1=2:LMRS_WO_SLEEP0=6,LMRS_WO_SLEEP1=2" > $O/ab.txt 2>&1
cat $O/ab.txt
LMRS_AQL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "greedy or merged or generate or chat or full_size" > $O/pytest_aql.txt 2>&1; tail -8 $O/pytest_aql.txt
