#!/bin/bash
# round 4, GPU call B: A/B of the three-part launch (fixed LDS attribute), full GPU tests with the new defaults
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4b; mkdir -p $O
timeout 900 python tools/ab_bench.py "base:LMRS_WO_MERGED=0@lm.rs_amd/liblmrs_hip_nokp.so" "kp:LMRS_WO_MERGED=0" "kp+wo" "nokp+wo@lm.rs_amd/liblmrs_hip_nokp.so" > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 200 python tools/vision_rate.py 2 24 > $O/vision.txt 2>&1; cat $O/vision.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
