#!/bin/bash
# A/B of compiler-flag builds of the same sources (make V=name EXTRA="..."), one GPU call: decode of Llama-3.2-1B Q8_0 and Gemma-2-2B Q4_0
# (PREFILL=1: fill_kv_cache(512) too).  Output under gpurun_out/abflags/.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/abflags; mkdir -p $O
V=""; for n in "$@"; do V="$V $n@lm.rs_amd/liblmrs_hip_$n.so"; done
timeout 400 python tools/ab_bench.py base $V base2 > $O/llama1b.txt 2>&1
timeout 400 python tools/ab_bench.py --model gemma-2-2b --qtype q4_0 base $V base2 > $O/gemma2b_q4.txt 2>&1
if [ "${PREFILL:-0}" = 1 ]; then
  rm -f $O/prefill512.txt
  for n in base "$@"; do
    L=lm.rs_amd/liblmrs_hip_$n.so; [ $n = base ] && L=lm.rs_amd/liblmrs_hip.so
    echo "== $n" >> $O/prefill512.txt; LMRS_LIB=$PWD/$L timeout 120 python tools/prefill_rate.py llama-3.2-1b 512 2>&1 | tail -4 >> $O/prefill512.txt
  done
fi
cat $O/*.txt
