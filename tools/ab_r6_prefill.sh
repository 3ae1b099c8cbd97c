# Round 6, one GPU call: the batched-prefill parity tests, then fill_kv_cache of the four models with / without the two launch fusions
# (LMRS_NO_PREFILL_FUSION=1, a context flag) -> profiles/r6_ab_prefill_fusion.txt.  (LMRS_Q4_WAVE_COLUMN was the build's A/B switch for the Q4_0 wave columns, adopted since.)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "w13_with or fill_kv_cache or batched or prefill or full_size or gemma_2b or random_geometries or multimodal_prefill" 2>&1 | tail -5
for env in "" "LMRS_NO_PREFILL_FUSION=1"; do
  for n in 512 256; do echo "== llama-3.2-1b $n [$env]"; env $env timeout 120 python tools/prefill_rate.py llama-3.2-1b $n 2>&1 | grep "fill_kv"; done
done
for env in "" "LMRS_NO_PREFILL_FUSION=1"; do
  echo "== llama-3.2-3b 512 [$env]"; env $env timeout 120 python tools/prefill_rate.py llama-3.2-3b 512 2>&1 | grep "fill_kv"
  echo "== phi-3.5 320 [$env]"; env $env timeout 120 python tools/prefill_rate.py phi-3.5 320 2>&1 | grep "fill_kv"
done
for env in "" "LMRS_Q4_WAVE_COLUMN=1" "LMRS_NO_PREFILL_FUSION=1"; do
  echo "== gemma q4 256 [$env]"; env $env timeout 120 python tools/prefill_rate.py gemma-2-2b 256 q4_0 2>&1 | grep "fill_kv\|checksum"
done
} > $O/ab_prefill_fusion.txt 2>&1
cat $O/ab_prefill_fusion.txt
