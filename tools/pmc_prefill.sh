#!/bin/bash
# PMC counters of a 512-token fill_kv_cache (two rocprofv3 passes, --kernel-trace only, as the chip guide prescribes): per-launch averages.
#   usage: bash tools/pmc_prefill.sh <out.txt> [n_tokens] [model] [qtype]
set -u
OUTF=${1:-gpurun_out/prefill_pmc.txt}; NT=${2:-512}; MODEL=${3:-llama-3.2-1b}; QT=${4:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/pmc_tmp; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $D/p1 -- python tools/prefill_rate.py $MODEL $NT $QT > $D/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $D/p2 -- python tools/prefill_rate.py $MODEL $NT $QT > $D/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $D/p3 -- python tools/prefill_rate.py $MODEL $NT $QT > $D/p3.log 2>&1
python - "$OUTF" $D/p1 $D/p2 $D/p3 <<'PY'
import collections, csv, glob, re, sys
out = open(sys.argv[1], "w")
out.write("# rocprofv3 --pmc, three passes (SQ; TCC + GRBM; LDS + MFMA), tools/prefill_rate.py: per-launch averages.\n"
          "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.\n")
for d in sys.argv[2:]:
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        out.write(f"# (no counters from {d})\n"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        n = max(len(v) for v in c.values())
        out.write(f"{k}\n    launches {n}  " + "  ".join(f"{cn}={sum(v)/len(v):.0f}" for cn, v in sorted(c.items())) + "\n")
out.close()
print(open(sys.argv[1]).read())
PY
rm -rf $D
