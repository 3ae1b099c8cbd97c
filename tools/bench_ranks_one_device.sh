#!/bin/bash
# the driver's N-rank launch with every rank on ONE GPU (LMRS_BENCH_ONE_DEVICE=1): verification of the sharded paths (token parity in every leg), not performance
#   usage: bash tools/bench_ranks_one_device.sh "<model> <qtype> <ranks>" ...
cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp
i=0
for spec in "$@"; do
  set -- $spec; i=$((i+1))
  LMRS_BENCH_ONE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $3 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --model $1 --qtype $2 --gpus $3 --steps 12 --warmup 5 --cpu-steps 6 2> gpurun_out/ranks_$i.err | grep "^{" | python -c "
import json,sys
n=0
for l in sys.stdin:
    n+=1; d=json.loads(l); print('$1 $2 x$3:', d['value'], 'tok/s |', d['config']['parallelism'][:60], '|', d.get('transport'), d.get('parity'), {k:(d[k].get('transport'), (d[k].get('parity') or {}).get('tokens_equal'), d[k].get('error')) for k in d if k in ('library_choice','tp_split_out')})
if not n: print('$1 $2 x$3: NO JSON LINE')"
  grep -i "Traceback\|failed\|error:" gpurun_out/ranks_$i.err | sort | uniq -c | head -4
done
