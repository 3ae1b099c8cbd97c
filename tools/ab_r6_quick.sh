# Round 6, one GPU call: GEMM parity tests + fill_kv_cache of the models + the CLIP tower on the current build (used after each GEMM change).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
{
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "matmul_q8_token_batch or fill_kv_cache_batched or w13_with or random_geometries or vision_tower" 2>&1 | tail -3
for spec in "llama-3.2-1b 512" "llama-3.2-1b 256" "llama-3.2-1b 128" "llama-3.2-3b 512" "phi-3.5 320" "gemma-2-2b 256 q4_0"; do
  echo "== $spec"; timeout 120 python tools/prefill_rate.py $spec 2>&1 | grep "fill_kv" | sed 's/.*on the device alone/   device/'
done
timeout 120 python tools/vision_rate.py 2 24 2>&1 | tail -2
} > $O/ab_quick.txt 2>&1
cat $O/ab_quick.txt
