# Round 6, one GPU call: Q4_0 / Gemma parity tests and the Q4_0 fill_kv_cache sweep of profiles/r6_ab_q4_pair_tiles.txt (3).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
{
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "q4 or gemma or fill_kv_cache or batched or random_geometries or vision or projector or multimodal" 2>&1 | grep -E "passed|failed" | tail -2
for spec in "gemma-2-2b 512 q4_0" "gemma-2-2b 384 q4_0" "gemma-2-2b 256 q4_0" "gemma-2-2b 128 q4_0" "gemma-2-2b 64 q4_0" "llama-3.2-1b 512 q4_0" "llama-3.2-1b 256 q4_0" "llama-3.2-1b 128 q4_0" "llama-3.2-3b 512 q4_0" "phi-3.5 320 q4_0"; do
  echo "== $spec"; timeout 120 python tools/prefill_rate.py $spec 2>&1 | grep "fill_kv" | sed 's/.*on the device alone/   device/'
done
} > $O/ab_q4_final.txt 2>&1
cat $O/ab_q4_final.txt
