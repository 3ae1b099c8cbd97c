#!/bin/bash
# round 4, GPU call A: AQL micro-benchmark, A/B of the kernel-argument preload and the three-part launch, CLIP attention forms, GPU tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4a; mkdir -p $O
timeout 120 tools/ubench/aql tools/ubench/aql_kernels.hsaco > $O/aql.txt 2>&1; echo "aql rc $?" >> $O/aql.txt
timeout 900 python tools/ab_bench.py "base:LMRS_WO_MERGED=0@lm.rs_amd/liblmrs_hip_nokp.so" "wo@lm.rs_amd/liblmrs_hip_nokp.so" "kp:LMRS_WO_MERGED=0" "kp+wo" "kp+wo(again)" > $O/ab.txt 2>&1
for v in "0 0" "0 1" "1 1" "2 1"; do set -- $v; echo "LMRS_VIS_ATT_SPLIT=$1 LMRS_VIS_ATT_SROW=$2" >> $O/vision.txt; LMRS_VIS_ATT_SPLIT=$1 LMRS_VIS_ATT_SROW=$2 timeout 200 python tools/vision_rate.py 2 24 >> $O/vision.txt 2>&1; done
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
cat $O/aql.txt $O/ab.txt $O/vision.txt
