#!/bin/bash
# the driver's invocation N times in fresh processes: wall-clock per step against the device's own time (start-up hiccups of the single timed call)
cd "$GRAFT_REPO_ROOT"; N=${1:-8}; shift
for i in $(seq 1 $N); do
  env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline']['path']
print('run $i:', d['value'], 'tok/s  wall', p['us_per_step'], 'us/step  device', p.get('device_event_us_per_step'), 'us/step  128-step', d.get('value_128_steps'))"
done
