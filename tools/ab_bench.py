"""A/B decode timing of build / environment variants on one GPU box, in one call.

    python tools/ab_bench.py [--model llama-3.2-1b] [--qtype q8_0] name[:ENV=V,ENV=V...][@lib.so] ...

Builds the synthetic image once (a file under /tmp), then runs every variant in its own process (environment switches and the
library are read once per process): 5 runs of the driver's 20 steps after a 5-token prompt, one run of 128 steps after 16, the
greedy tokens of both compared with the first variant's.  Prints one line per variant."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, model_name, qtype):
    import ctypes
    import numpy as np
    import lmrs_amd
    from tools import synth_lmrs as S
    hip = ctypes.CDLL("libamdhip64.so")
    img = np.fromfile(path, dtype=np.uint8)
    cfg = S.CONFIGS[model_name]
    m = lmrs_amd.Transformer(img, device=0)
    out = {}
    for (W, K, reps) in ((5, 20, 6), (16, 128, 2)):
        prompt = S.prompt_tokens(cfg, W, 1234)
        best = 1e9
        for r in range(reps):
            first = m.generate_greedy(prompt, 1)
            hip.hipDeviceSynchronize()
            t1 = time.perf_counter()
            toks = m.generate_greedy(first, K, start_pos=W)
            hip.hipDeviceSynchronize()
            dt = time.perf_counter() - t1
            if r:
                best = min(best, dt)
        out[f"us_{K}"] = round(best / K * 1e6, 2)
        out[f"tok_{K}"] = [int(first[0])] + [int(t) for t in toks]
    out["launches"] = m.step_info(20)[0]
    print("AB_RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3.2-1b")
    ap.add_argument("--qtype", default="q8_0")
    ap.add_argument("--child", default=None)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.model, a.qtype)
    from tools import synth_lmrs as S
    path = f"/tmp/ab_{a.model}_{a.qtype}.lmrs"
    if not os.path.exists(path):
        img = S.build_image(S.CONFIGS[a.model], S.Q8_0 if a.qtype == "q8_0" else S.Q4_0, seed=1234)
        img.tofile(path)
    ref = None
    for v in a.variants or ["base"]:
        lib = None
        if "@" in v:
            v, lib = v.split("@", 1)
        name, _, envs = v.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        if lib:
            env["LMRS_LIB"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--model", a.model, "--qtype", a.qtype, "--child", path], env=env,
                           capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
        if p.returncode or not line:
            print(f"{name:28s} FAILED rc={p.returncode}: {(p.stderr or p.stdout)[-400:]}")
            continue
        r = json.loads(line[0][10:])
        if ref is None:
            ref = r
        same = r["tok_20"] == ref["tok_20"] and r["tok_128"] == ref["tok_128"]
        print(f"{name:28s} 20 steps: {r['us_20']:8.2f} us/step ({1e6 / r['us_20']:7.1f} tok/s)   128 steps: {r['us_128']:8.2f} us/step ({1e6 / r['us_128']:7.1f} tok/s)   "
              f"launches/step {r['launches']}   tokens {'equal' if same else 'DIFFER'}", flush=True)


if __name__ == "__main__":
    main()
