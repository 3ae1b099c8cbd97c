# Round 6, one GPU call: parity + fill_kv_cache under LMRS_GEMM_GPB = 1 / 2 / 4 (groups per barrier) -> profiles/r6_ab_gemm_structure.txt (1).
# The switch exists only in tools/ubench/r6_gemm_gpb_and_column.patch (the form measured no faster and left the library); kept as the record of how the numbers were taken.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
{
for g in 2 4; do
  echo "== parity LMRS_GEMM_GPB=$g"; LMRS_GEMM_GPB=$g timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "matmul_q8_token_batch or fill_kv_cache_batched or full_size or gemma_2b_q4_batched or random_geometries" 2>&1 | tail -3
done
for g in 1 2 4; do
  for spec in "llama-3.2-1b 512" "llama-3.2-1b 256" "llama-3.2-1b 128" "llama-3.2-3b 512" "phi-3.5 320" "gemma-2-2b 256 q4_0" "gemma-2-2b 512 q4_0"; do
    echo "== GPB=$g $spec"; LMRS_GEMM_GPB=$g timeout 120 python tools/prefill_rate.py $spec 2>&1 | grep "fill_kv" | sed 's/.*on the device alone/   device/'
  done
done
echo "== GPB=2 + Q4 wave column, gemma 256"; LMRS_GEMM_GPB=2 LMRS_Q4_WAVE_COLUMN=1 timeout 120 python tools/prefill_rate.py gemma-2-2b 256 q4_0 2>&1 | grep "fill_kv" | sed 's/.*on the device alone/   device/'
} > $O/ab_gemm_gpb.txt 2>&1
cat $O/ab_gemm_gpb.txt
