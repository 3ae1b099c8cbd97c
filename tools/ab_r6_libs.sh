# same-box A/B of library builds on fill_kv_cache: usage ab_r6_libs.sh name...   (base = lm.rs_amd/liblmrs_hip_base.so, cur = the library)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
for rep in 1 2; do
for n in "$@"; do
  L=$PWD/lm.rs_amd/liblmrs_hip_$n.so; [ $n = cur ] && L=$PWD/lm.rs_amd/liblmrs_hip.so
  for spec in "llama-3.2-1b 512" "llama-3.2-1b 256" "llama-3.2-1b 128" "llama-3.2-3b 512" "llama-3.2-3b 256" "phi-3.5 320" "gemma-2-2b 256 q4_0" "gemma-2-2b 256"; do
    echo -n "$n  $spec: "; LMRS_LIB=$L timeout 120 python tools/prefill_rate.py $spec 2>&1 | grep "fill_kv" | sed 's/.*on the device alone//'
  done
done
done > $O/ab_libs.txt 2>&1
cat $O/ab_libs.txt
