#!/bin/bash
# The one GPU-box runner (through gpurun, from the repo root):  bash tools/gpu_run.sh <stage> [<stage> ...]
# Every stage writes under gpurun_out/<tag>/ (tag = $LMRS_RUN_TAG, default r5); nothing here is timed by the driver.
#   gemmpipe   tools/ubench/gemmpipe: the pipelined int8 GEMM variants, self-check + timings
#   tests      pytest -m gpu
#   bench      the driver's bench invocation (20 steps) and the 128-step default
#   models     bench lines of the four BASELINE models (64 steps)
#   stats M Q  rocprofv3 --kernel-trace --stats of a 64-step bench of model M, qtype Q (PROF_ENV, default LMRS_NO_GRAPH=1: the step's launches
#              enqueued one by one - rocprofv3 1.1 segfaults on the graph launches of every model but Llama-3.2-1B, see profiles/README.md)
#   pmc M Q    FETCH_SIZE / WRITE_SIZE passes of the same (separate runs, kernel-trace only)
#   prefill N  fill_kv_cache(N) rate + rocprofv3 kernel statistics
#   vision     CLIP tower rate + rocprofv3 kernel statistics
#   ab "..."   tools/ab_bench.py with the quoted arguments
#   crash M Q  rocprofv3 probes: with / without kernel-argument preload (build the second library first: make -C lm.rs_amd/csrc V=nokp KPRELOAD=), 1 / 4 steps per graph launch
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export LMRS_BENCH_IMAGE_CACHE=/tmp
TAG=${LMRS_RUN_TAG:-r5}; OUT=gpurun_out/$TAG; mkdir -p $OUT
short() { case $1 in llama-3.2-1b) echo llama1b;; gemma-2-2b) echo gemma2b;; llama-3.2-3b) echo llama3b;; phi-3.5) echo phi35;; *) echo $1;; esac; }
while [ $# -gt 0 ]; do
    st=$1; shift
    case $st in
    gemmpipe) make -s -C tools/ubench gemmpipe && timeout 300 tools/ubench/gemmpipe "${GEMMPIPE_ARGS:-}" > $OUT/ubench_gemmpipe.txt 2>&1; tail -5 $OUT/ubench_gemmpipe.txt;;
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt;;
    bench) timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench.err; timeout 300 python bench.py > $OUT/bench.json 2>> $OUT/bench.err; cat $OUT/bench_20steps.json;;
    models)
        timeout 400 python bench.py --model gemma-2-2b --qtype q4_0 --steps 64 > $OUT/bench_gemma2b_q4.json 2>> $OUT/bench.err
        timeout 400 python bench.py --model llama-3.2-3b --steps 64 > $OUT/bench_llama3b.json 2>> $OUT/bench.err
        timeout 400 python bench.py --model phi-3.5 --steps 64 > $OUT/bench_phi35.json 2>> $OUT/bench.err;;
    stats) M=$1; Q=$2; shift 2; N=$(short $M)_$Q
        rm -rf $OUT/st_$N
        env ${PROF_ENV:-LMRS_NO_GRAPH=1} timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$N -- python bench.py --model $M --qtype $Q --steps 64 --cpu-steps 0 > $OUT/stats_$N.log 2>&1
        echo "stats $N rc $?"; cp $(ls $OUT/st_$N/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats_$N.csv 2>/dev/null; rm -rf $OUT/st_$N;;
    pmc) M=$1; Q=$2; shift 2; N=$(short $M)_$Q
        for c in FETCH_SIZE WRITE_SIZE; do
            for try in 1 2 3; do                            # (rocprofv3 1.1's counter collection hangs or dies at start-up now and then, whatever the workload)
                rm -rf $OUT/pmc_${N}_$c
                env ${PROF_ENV:-LMRS_NO_GRAPH=1} timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${N}_$c -- python bench.py --model $M --qtype $Q --steps 16 --warmup 4 --cpu-steps 0 > $OUT/pmc_${N}_$c.log 2>&1
                echo "pmc $N $c try $try rc $?"
                ls $OUT/pmc_${N}_$c/*/*counter_collection.csv > /dev/null 2>&1 && break
            done
        done
        f=$(ls $OUT/pmc_${N}_FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -1); w=$(ls $OUT/pmc_${N}_WRITE_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
        if [ -n "$f" ] && [ -n "$w" ]; then python tools/pmc_summary.py "$f" "$w" $OUT/traffic_$N.json $M $Q; fi
        rm -rf $OUT/pmc_${N}_FETCH_SIZE $OUT/pmc_${N}_WRITE_SIZE;;
    prefill) N=$1; shift
        rm -rf $OUT/pf
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf -- python tools/prefill_rate.py llama-3.2-1b $N > $OUT/prefill$N.log 2>&1
        cp $(ls $OUT/pf/*/*kernel_stats.csv | head -1) $OUT/prefill${N}_kernel_stats.csv; rm -rf $OUT/pf; grep fill_kv_cache $OUT/prefill$N.log;;
    vision)
        rm -rf $OUT/vis
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vis -- python tools/vision_rate.py 2 24 > $OUT/vision_rate.log 2>&1
        cp $(ls $OUT/vis/*/*kernel_stats.csv | head -1) $OUT/vision_kernel_stats.csv; rm -rf $OUT/vis; tail -2 $OUT/vision_rate.log;;
    ab) A=$1; shift; f=$OUT/ab_$(date +%H%M%S).txt; timeout 900 python tools/ab_bench.py $A > $f 2>&1; tail -12 $f;;
    crash) M=$1; Q=$2; shift 2            # which of {kernel-argument preload, steps per graph launch} rocprofv3's interceptor dies on (profiles/README.md)
        for v in nokp1 kp4 nokp4; do
            case $v in nokp1) E="LMRS_LIB=$PWD/lm.rs_amd/liblmrs_hip_nokp.so LMRS_STEPS_PER_GRAPH=1";; kp4) E="LMRS_STEPS_PER_GRAPH=4";; nokp4) E="LMRS_LIB=$PWD/lm.rs_amd/liblmrs_hip_nokp.so LMRS_STEPS_PER_GRAPH=4";; esac
            rm -rf /tmp/cx; env $E timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cx -- python bench.py --model $M --qtype $Q --steps 16 --cpu-steps 0 > $OUT/crash_$(short $M)_$v.log 2>&1
            echo "crash-probe $M $v rc $? $(ls /tmp/cx/*/*kernel_stats.csv 2>/dev/null | wc -l) stats files"
        done;;
    sh) bash -c "$1" > $OUT/sh_$(date +%H%M%S).txt 2>&1; shift;;
    *) echo "unknown stage $st"; exit 2;;
    esac
done
