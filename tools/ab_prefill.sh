#!/bin/bash
# A/B of builds on the batched paths, one GPU call: the GEMM parity test, fill_kv_cache(512 / 256), the CLIP tower.  usage: ab_prefill.sh name...
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/abprefill; mkdir -p $O; rm -f $O/*.txt
for n in "$@"; do
  L=$PWD/lm.rs_amd/liblmrs_hip_$n.so; [ $n = base ] && L=$PWD/lm.rs_amd/liblmrs_hip.so
  echo "== $n" >> $O/out.txt
  [ $n != base ] && LMRS_LIB=$L timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k "matmul_q8_token_batch or fill_kv_cache_batched" 2>&1 | tail -2 >> $O/out.txt
  LMRS_LIB=$L timeout 120 python tools/prefill_rate.py llama-3.2-1b 512 2>&1 | grep -v "^token by token" | tail -3 >> $O/out.txt
  LMRS_LIB=$L timeout 120 python tools/vision_rate.py 2 24 2>&1 | tail -3 >> $O/out.txt
done
cat $O/out.txt
