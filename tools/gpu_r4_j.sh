#!/bin/bash
# round 4, GPU call J: cluster carry by DPP instead of ds_bpermute; top-p rates with the sort-merge shortcut
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4j; mkdir -p $O
timeout 600 python tools/ab_bench.py "shfl-carry@lm.rs_amd/liblmrs_hip_shfl.so" "dpp-carry" "shfl-carry(again)@lm.rs_amd/liblmrs_hip_shfl.so" "dpp-carry(again)" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 600 python tools/ab_bench.py --model llama-3.2-3b "shfl-carry@lm.rs_amd/liblmrs_hip_shfl.so" "dpp-carry" > $O/ab_3b.txt 2>&1; cat $O/ab_3b.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "matmul or golden or mini or random_geom or sharding_is_bit" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/sampler_rate.py > $O/sampler_rate.txt 2>&1; cat $O/sampler_rate.txt
