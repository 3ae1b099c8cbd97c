#!/bin/bash
# round 4, GPU call Y: the build without SLP vectorisation (no v_pk_*_f32) against the default build: decode, prefill, CLIP tower
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4y; mkdir -p $O
N=lm.rs_amd/liblmrs_hip_noslp.so
timeout 400 python tools/ab_bench.py new noslp@$N "new(again)" "noslp(again)@$N" > $O/ab_llama1b.txt 2>&1; cat $O/ab_llama1b.txt
for v in default noslp default noslp; do
  if [ $v = noslp ]; then export LMRS_LIB=$PWD/$N; else unset LMRS_LIB; fi
  echo "== $v" >> $O/rates.txt
  timeout 300 python tools/prefill_rate.py llama-3.2-1b 512 2>&1 | grep fill_kv >> $O/rates.txt
  timeout 300 python tools/vision_rate.py 2 24 2>&1 | grep tower >> $O/rates.txt
done
cat $O/rates.txt | cut -c1-250
