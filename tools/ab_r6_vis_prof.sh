#!/bin/bash
# CLIP tower: per-kernel durations under rocprofv3, with and without the stray-query workgroups (LMRS_VIS_NO_STRAY=1)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; export LMRS_BENCH_IMAGE_CACHE=/tmp
mkdir -p gpurun_out/r6
echo "== parity (vision / image / multimodal tests)"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vision or image or multimodal" 2>&1 | tail -2
echo "== parity (LMRS_VIS_NO_STRAY=1)"; LMRS_VIS_NO_STRAY=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vision_tower_matches" 2>&1 | tail -2
for v in stray no_stray; do
  E=""; [ $v = no_stray ] && E="LMRS_VIS_NO_STRAY=1"
  for i in 1 2; do env $E timeout 300 python tools/vision_rate.py 2 24 2>&1 | grep tower | cut -c1-120; done
  rm -rf gpurun_out/vp; env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vp -- python tools/vision_rate.py 2 24 > gpurun_out/r6/vis_$v.log 2>&1
  cp $(ls gpurun_out/vp/*/*kernel_stats.csv | head -1) gpurun_out/r6/vis_stats_$v.csv; rm -rf gpurun_out/vp
  echo "== $v (under rocprofv3)"; grep tower gpurun_out/r6/vis_$v.log | cut -c1-100
  python - gpurun_out/r6/vis_stats_$v.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"  {r['Name'][:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:8.1f} us")
PY
done
