# Round 6, one GPU call: parity + fill_kv_cache with / without LMRS_GEMM_COL (the column kernel) -> profiles/r6_ab_gemm_structure.txt (3), (3b).
# The switch exists only in tools/ubench/r6_gemm_gpb_and_column.patch / r6_gemm_column_v2.patch (measured slower; not in the library).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
{
echo "== parity LMRS_GEMM_COL=1"; LMRS_GEMM_COL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "matmul_q8_token_batch or fill_kv_cache_batched or full_size or gemma_2b_q4_batched or random_geometries" 2>&1 | tail -3
for e in LMRS_X=1 LMRS_GEMM_COL=1; do
  for spec in "llama-3.2-1b 512" "llama-3.2-1b 256" "llama-3.2-1b 128" "llama-3.2-3b 512" "phi-3.5 320" "gemma-2-2b 256 q4_0"; do
    echo "== $e $spec"; env $e timeout 120 python tools/prefill_rate.py $spec 2>&1 | grep "fill_kv" | sed 's/.*on the device alone/   device/'
  done
done
} > $O/ab_gemm_col.txt 2>&1
cat $O/ab_gemm_col.txt
