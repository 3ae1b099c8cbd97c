# Round 6, one GPU call: Gemma / Q4_0 parity tests, Gemma-2-2B Q4_0 fill_kv_cache at 512 / 256 / 128 tokens, rocprofv3 kernel statistics of the 256-token fill.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
{
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "matmul_q8_token_batch or fill_kv_cache or batched or gemma or random_geometries or q4" 2>&1 | tail -3
for spec in "gemma-2-2b 256 q4_0" "gemma-2-2b 512 q4_0" "gemma-2-2b 128 q4_0" "llama-3.2-1b 512" "llama-3.2-1b 256"; do
  echo "== $spec"; timeout 120 python tools/prefill_rate.py $spec 2>&1 | grep "fill_kv" | sed 's/.*on the device alone/   device/'
done
} > $O/ab_gemma_prefill.txt 2>&1
cat $O/ab_gemma_prefill.txt
bash tools/prof_r6_prefill.sh "gemma-2-2b 256 q4_0"
