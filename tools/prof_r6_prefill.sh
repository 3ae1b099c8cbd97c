#!/bin/bash
# rocprofv3 kernel statistics of fill_kv_cache: usage (on the GPU box)  bash tools/prof_r6_prefill.sh "<model> <n> <q8_0|q4_0> [ENV=..]" ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
for spec in "$@"; do
  set -- $spec; M=$1; N=$2; Q=$3; E=${4:-LMRS_X=1}
  tag=$(echo ${M}_${N}_${Q}_${E} | tr -c 'a-zA-Z0-9_\n' '_')
  rm -rf /tmp/pf
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python tools/prefill_rate.py $M $N $Q > $O/prefill_$tag.log 2>&1
  cp $(ls /tmp/pf/*/*kernel_stats.csv | head -1) $O/prefill_${tag}_kernel_stats.csv
  grep fill_kv $O/prefill_$tag.log
done
