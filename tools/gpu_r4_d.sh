#!/bin/bash
# round 4, GPU call D: RMSNorm chains from registers (DPP) against the LDS-fed chains; Gemma-2-2B Q4_0 baseline + timelines
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4d; mkdir -p $O
timeout 600 python tools/ab_bench.py "lds-chain@lm.rs_amd/liblmrs_hip_rms0.so" "dpp-chain" "lds-chain(again)@lm.rs_amd/liblmrs_hip_rms0.so" "dpp-chain(again)" > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
timeout 600 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 "lds-chain@lm.rs_amd/liblmrs_hip_rms0.so" "dpp-chain" > $O/ab_gemma.txt 2>&1; cat $O/ab_gemma.txt
timeout 600 python tools/ab_bench.py --model llama-3.2-3b "lds-chain@lm.rs_amd/liblmrs_hip_rms0.so" "dpp-chain" > $O/ab_3b.txt 2>&1; cat $O/ab_3b.txt
timeout 300 python tools/timeline.py llama-3.2-1b 100 > $O/timeline_llama1b.txt 2>&1
timeout 300 python tools/timeline.py gemma-2-2b 100 q4_0 > $O/timeline_gemma2b_q4.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q -x -k "rmsnorm or quantize or golden or merged or greedy_token_ids or random_geom or mini" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
tail -25 $O/timeline_llama1b.txt; tail -25 $O/timeline_gemma2b_q4.txt
