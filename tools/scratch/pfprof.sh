cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 256 512; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_$n -o s -- python $R/tools/prefill_rate.py llama-3.2-1b $n > $R/gpurun_out/pf_$n.log 2>&1
f=$(find $R/gpurun_out/pf_$n -name "*kernel_stats.csv" | head -1)
echo "== $n tokens"; grep "fill_kv" $R/gpurun_out/pf_$n.log | sed 's/.*MFMA): //'
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:86]:86s} n={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
done
