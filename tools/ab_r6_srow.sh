#!/bin/bash
# batched attention value phase: LDS-fed (att_values_kernel) against scalar-fed rows (LMRS_ATT_VALUES_SROW=1; needs tools/ubench/r6_att_values_srow.patch applied - the form is not in the library)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp
mkdir -p gpurun_out/r6; O=gpurun_out/r6/ab_srow.txt; : > $O
echo "== parity, LMRS_ATT_VALUES_SROW=1 (fill_kv_cache / prefill / multimodal tests)" >> $O
LMRS_ATT_VALUES_SROW=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fill_kv or prefill or batched or multimodal or prompt" 2>&1 | tail -3 >> $O
for cfgs in "llama-3.2-1b 512" "llama-3.2-1b 256" "llama-3.2-1b 128" "llama-3.2-3b 512" "phi-3.5 320"; do
  for v in 0 1; do
    E=""; [ $v = 1 ] && E="LMRS_ATT_VALUES_SROW=1"
    echo "== $cfgs srow=$v" >> $O
    for i in 1 2; do env $E timeout 300 python tools/prefill_rate.py $cfgs 2>&1 | grep -o "on the device.*" | cut -c1-60 >> $O; done
  done
done
cat $O
