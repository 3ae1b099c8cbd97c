#!/bin/bash
# round 4, GPU call E: Infinity-Cache stream rate; the stalled-peer test with and without the double-buffered partials; full GPU tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4e; mkdir -p $O
timeout 300 tools/ubench/mall > $O/mall.txt 2>&1; cat $O/mall.txt
LMRS_SHARD_SINGLE_PARTIALS=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stalled_peer" > $O/stall_single.txt 2>&1; echo "single buffer (expected to FAIL):"; tail -4 $O/stall_single.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stalled_peer" > $O/stall_double.txt 2>&1; echo "double buffer:"; tail -4 $O/stall_double.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 300 python tools/ab_bench.py "default" "aql:LMRS_AQL=1" > $O/ab.txt 2>&1; cat $O/ab.txt
