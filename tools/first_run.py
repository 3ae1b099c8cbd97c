"""Per-call decode time of successive generate calls after a fresh context (is the first timed call - what bench.py measures - slower?).
usage: python tools/first_run.py [model]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lmrs_amd
from tools import synth_lmrs as S
hip = ctypes.CDLL("libamdhip64.so")
model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
path = f"/tmp/ab_{model}_q8_0.lmrs"
img = np.fromfile(path, dtype=np.uint8) if os.path.exists(path) else S.build_image(S.CONFIGS[model], S.Q8_0, seed=1234)
for trial in range(2):
    m = lmrs_amd.Transformer(img, device=0)
    W, K = 5, 20
    prompt = S.prompt_tokens(S.CONFIGS[model], W, 1234)
    out = []
    first = m.generate_greedy(prompt, 1); hip.hipDeviceSynchronize()
    for r in range(6):
        t1 = time.perf_counter(); toks, dev = m.generate_greedy(first, K, start_pos=W, timing=True); hip.hipDeviceSynchronize(); dt = time.perf_counter() - t1
        out.append((round(dt / K * 1e6, 1), round(dev / K * 1e6, 1)))
        if r == 2: time.sleep(0.2)           # an idle gap before the fourth call
    print(f"fresh context {trial}: (wall us/step, device-event us/step) per call:", out)
    m.close()
