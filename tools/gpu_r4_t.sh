#!/bin/bash
# round 4 (second session), GPU call T: grouped quantiser passes / asm DPP maxima / division-free scale / three gate-up passes per activation
# - the whole GPU tier, then same-box A/B of the builds (base = the sources of commit 83f57cb)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4t; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
B=lm.rs_amd/liblmrs_hip_base.so
timeout 500 python tools/ab_bench.py base@$B new nogrp@lm.rs_amd/liblmrs_hip_nogrp.so new-w2l32:LMRS_W2_L32=1 "base(again)@$B" "new(again)" > $O/ab_llama1b.txt 2>&1; cat $O/ab_llama1b.txt
timeout 500 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 base@$B new pairs@lm.rs_amd/liblmrs_hip_pairs.so "new(again)" > $O/ab_gemma2b_q4.txt 2>&1; cat $O/ab_gemma2b_q4.txt
