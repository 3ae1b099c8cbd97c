#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4w; mkdir -p $O
B=lm.rs_amd/liblmrs_hip_base.so
timeout 500 python tools/ab_bench.py base@$B new "base(again)@$B" "new(again)" > $O/ab_llama1b.txt 2>&1; cat $O/ab_llama1b.txt
timeout 500 python tools/ab_bench.py --model llama-3.2-3b base@$B new > $O/ab_llama3b.txt 2>&1; cat $O/ab_llama3b.txt
timeout 500 python tools/ab_bench.py --model phi-3.5 base@$B new > $O/ab_phi35.txt 2>&1; cat $O/ab_phi35.txt
