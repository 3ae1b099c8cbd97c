"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes  ->  profiles/<tag>_traffic.json  (HBM traffic per kernel launch).

Collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: separate --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots),
--kernel-trace only.  Units: the counters are in KiB.  gfx950 correction: FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (16 B/lane), which is what these kernels do, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is taken as is.

The summary records which model / quantisation / build of the kernels it was taken on (bench.py reports `traffic` only on a match).

usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/<tag>_traffic.json [model] [qtype]
"""
import collections
import csv
import json
import re
import sys


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


def main(fetch_csv, write_csv, out, model="llama-3.2-1b", qtype="q8_0"):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
    res = {}
    for k, (fv, n) in f.items():
        wv = w.get(k, (0.0, 0))[0]
        res[k] = {"launches": n, "FETCH_SIZE_KiB_avg": round(fv, 1), "WRITE_SIZE_KiB_avg": round(wv, 1),
                  "hbm_bytes_per_launch": round((2 * fv + wv) * 1024)}
    gemv = {k: v for k, v in res.items() if "gemv" in k or "qkv_attn" in k}
    # one decode step of Llama-3.2-1B launches, per layer, one of each of the four layer GEMVs, and one classifier GEMV
    # how the profiled run enqueued the step (tools/gpu_run.sh sets LMRS_NO_GRAPH=1 by default: rocprofv3 1.1 dies on the graph launches of most models);
    # the published tok/s comes from graph replays of the SAME launches - recorded here so that nobody has to guess which mode a summary saw
    mode = os.environ.get("LMRS_PROFILE_LAUNCH_MODE", "eager (LMRS_NO_GRAPH=1: the step's launches enqueued one by one)")
    doc = {"note": __doc__.split("usage")[0].strip(), "model": model, "qtype": qtype, "kernel_source_hash": kernel_source_hash(), "launch_mode": mode, "kernels": res,
           "gemv_bytes_weighted_by_launch_count": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in gemv.values()) / max(1, sum(v["launches"] for v in gemv.values())))}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k[:60]:60s} n={v['launches']:5d}  {v['hbm_bytes_per_launch'] / 1e6:9.3f} MB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:6])
