#!/bin/bash
# durations of the batched attention's three launches (and the GEMMs) against the batch length: rocprofv3 kernel statistics of fill_kv_cache(n), Llama-3.2-1B
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp
mkdir -p gpurun_out/r6; O=gpurun_out/r6/prefill_attention_model.txt; : > $O
for n in 64 128 192 256 320 384 448 512; do
  rm -rf gpurun_out/pm; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pm -- python tools/prefill_rate.py llama-3.2-1b $n > gpurun_out/r6/pm_$n.log 2>&1
  f=$(ls gpurun_out/pm/*/*kernel_stats.csv | head -1)
  python - $f $n >> $O <<'PY'
import csv, sys
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
def avg(sub):
    t = c = 0
    for k, r in rows.items():
        if sub in k: t += float(r["TotalDurationNs"]); c += int(r["Calls"])
    return t / c / 1000 if c else float("nan")
print(f"tokens {int(sys.argv[2]):4d}: scores {avg('att_scores_kernel'):6.2f}  softmax {avg('att_softmax_kernel'):6.2f}  values {avg('att_values_kernel'):6.2f} us per launch")
PY
  rm -rf gpurun_out/pm gpurun_out/r6/pm_$n.log
done
cat $O
