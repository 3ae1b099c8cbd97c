#!/bin/bash
# Regenerates the round's measurement artefacts on a GPU box (run from the repo root through gpurun); everything lands in
# gpurun_out/art/ and is copied into profiles/ by hand afterwards.  PMC counters are collected in their own rocprofv3 runs
# (--kernel-trace only, one counter per pass), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
#   usage: bash tools/make_profiles.sh <tag> [pmc|lite|extra|hash]   (`hash`: only what bench.py matches by the build identifier - the Llama-3.2-1B and Gemma
#          traffic summaries, the rocprofv3 summary - and the two Llama-3.2-1B bench lines that quote them)
#   modes: [pmc|lite|extra]   (e.g. r4; `pmc`: only the counter passes and the bench lines that quote them;
#          `lite`: counters, rocprofv3 kernel statistics and the bench lines of the four models - no timeline / prefill / vision / two-rank runs)
set -u
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export LMRS_BENCH_IMAGE_CACHE=/tmp          # the synthetic images are built once, outside the profiler
OUT=gpurun_out/art; ONLY=${2:-all}
if [ "$ONLY" = all ] || [ "$ONLY" = lite ]; then rm -rf $OUT; fi
if [ "$ONLY" = hash ]; then HASHONLY=1; ONLY=lite; fi
mkdir -p $OUT
pmc() {   # model qtype out-json
    for c in FETCH_SIZE WRITE_SIZE; do
        # (one step per graph launch: rocprofv3 1.1's counter collection crashes on the 260-node graphs of the default four steps per launch;
        # and it segfaults now and then at start-up whatever the workload - or hangs: a pass that works takes 25 s, so 60 s per try, up to three tries)
        for try in 1 2 3; do
            rm -rf $OUT/pmc_$1_$c
            KT=--kernel-trace; [ $try = 3 ] && KT=            # (last try: counters alone - the trace + counter combination is what crashed on Gemma-2-2B in round 3)
            # (LMRS_NO_GRAPH=1: the step's launches enqueued one by one - rocprofv3 1.1 segfaults on the graph launches of every model but Llama-3.2-1B)
            env LMRS_NO_GRAPH=1 timeout -k 5 90 rocprofv3 $KT --pmc $c --output-format csv -d $OUT/pmc_$1_$c -- python bench.py --model $1 --qtype $2 --steps 16 --warmup 4 --cpu-steps 0 > $OUT/pmc_$1_$c.log 2>&1
            ls $OUT/pmc_$1_$c/*/*counter_collection.csv > /dev/null 2>&1 && break
        done
    done
    f=$(ls $OUT/pmc_$1_FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -1); w=$(ls $OUT/pmc_$1_WRITE_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -z "$f" ] || [ -z "$w" ]; then echo "pmc $1: no counter data (rocprofv3 failed on every try)"; return; fi
    python tools/pmc_summary.py "$f" "$w" $3 $1 $2
    cp $3 profiles/                                   # bench.py reads profiles/*traffic*.json (matching model / qtype / source hash)
    rm -rf $OUT/pmc_$1_FETCH_SIZE $OUT/pmc_$1_WRITE_SIZE
}
stats() {   # model qtype short-name: their bench line (which also leaves the image in the cache), then rocprofv3 kernel statistics of the same 64-step bench
    timeout 400 python bench.py --model $1 --qtype $2 --steps 64 > $OUT/${TAG}_bench_$3.json 2>> $OUT/bench.err
    for try in 1 2 3; do                                # (rocprofv3 1.1 segfaults at start-up now and then, whatever the workload)
        rm -rf $OUT/st_$3
        # (LMRS_NO_GRAPH=1: the same launches enqueued one by one - rocprofv3 1.1 segfaults inside hipGraphLaunch for these three models, every time,
        # with or without kernel-argument preload, one or four steps per graph: profiles/README.md; same kernels, same order)
        LMRS_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$3 -- python bench.py --model $1 --qtype $2 --steps 64 --cpu-steps 0 > $OUT/st_$3.log 2>&1
        ls $OUT/st_$3/*/*kernel_stats.csv > /dev/null 2>&1 && break
    done
    cp $(ls $OUT/st_$3/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats_$3.csv 2>/dev/null; rm -rf $OUT/st_$3
}
if [ "$ONLY" != extra ]; then pmc llama-3.2-1b q8_0 $OUT/${TAG}_traffic_llama1b_q8.json; pmc gemma-2-2b q4_0 $OUT/${TAG}_traffic_gemma2b_q4.json; fi
if [ "$ONLY" = extra ]; then stats gemma-2-2b q4_0 gemma2b_q4; stats llama-3.2-3b q8_0 llama3b; stats phi-3.5 q8_0 phi35; exit 0; fi
if [ "$ONLY" = pmc ]; then timeout 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20steps.json 2>> $OUT/bench.err; exit 0; fi
timeout 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --cpu-steps 0 > $OUT/stats.log 2>&1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv; rm -rf $OUT/stats
python tools/rocprof_summary.py $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_rocprof_llama1b_q8.json llama-3.2-1b q8_0 && cp $OUT/${TAG}_rocprof_llama1b_q8.json profiles/   # bench.py reads it (frac_rocprof)
timeout 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err                      # again: now with frac_rocprof and traffic of this build
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20steps.json 2>> $OUT/bench.err   # the driver's invocation
if [ "${HASHONLY:-0}" = 1 ]; then exit 0; fi
stats gemma-2-2b q4_0 gemma2b_q4; stats llama-3.2-3b q8_0 llama3b; stats phi-3.5 q8_0 phi35        # (each also rewrites its bench line)
if [ "$ONLY" = lite ]; then exit 0; fi          # (`lite`: counters, kernel statistics and the bench lines of the four models only)
timeout 200 python tools/timeline.py llama-3.2-1b 100 > $OUT/${TAG}_timeline_llama1b.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf -- python tools/prefill_rate.py llama-3.2-1b 512 > $OUT/${TAG}_prefill512.log 2>&1
cp $(ls $OUT/pf/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prefill512_kernel_stats.csv; rm -rf $OUT/pf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf -- python tools/prefill_rate.py llama-3.2-1b 256 > $OUT/${TAG}_prefill256.log 2>&1
cp $(ls $OUT/pf/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prefill256_kernel_stats.csv; rm -rf $OUT/pf
{ python tools/prefill_rate.py gemma-2-2b 256 q4_0; python tools/prefill_rate.py llama-3.2-3b 512; python tools/prefill_rate.py phi-3.5 320; } 2>&1 | grep fill_kv > $OUT/${TAG}_prefill_other_models.txt
timeout 300 python tools/sampler_rate.py > $OUT/${TAG}_sampler_rate.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vis -- python tools/vision_rate.py 2 24 > $OUT/${TAG}_vision_rate.log 2>&1
cp $(ls $OUT/vis/*/*kernel_stats.csv | head -1) $OUT/${TAG}_vision_kernel_stats.csv; rm -rf $OUT/vis
LMRS_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 2 --steps 64 --warmup 16 --cpu-steps 8 2> $OUT/tp2.err | grep "^{" > $OUT/${TAG}_bench_two_ranks_one_device_both_plans.json
timeout 400 python bench.py --model phi-3.5 --steps 32 --vision > $OUT/${TAG}_bench_phi35_vision.json 2>> $OUT/bench.err
for f in $OUT/${TAG}_bench*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], "tok/s  frac", r.get("frac"), "traffic", r.get("traffic"), "parity", d.get("parity"))
PY
done
