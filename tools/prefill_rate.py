"""Throughput of fill_kv_cache (reference transformer.rs:672-684: forward_layer over a batch of embeddings): the batched
int8-MFMA path vs the token-by-token decode graph (LMRS_NO_BATCHED_PREFILL=1).
usage: python tools/prefill_rate.py [model] [n_tokens] [q8_0|q4_0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import lmrs_amd  # noqa: E402
from tools import synth_lmrs as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
qt = S.Q4_0 if len(sys.argv) > 3 and sys.argv[3] == "q4_0" else S.Q8_0
img = S.build_image(model, qt, 1234)
m = lmrs_amd.Transformer(img)
toks = S.prompt_tokens(model, n, 1234)
emb = m.get_embeddings(toks)
out = {}
for rep in range(3):
    e = emb.copy()
    t0 = time.perf_counter()
    m.fill_kv_cache(e, 0)
    dt = time.perf_counter() - t0
    out[rep] = dt
    try:
        dev = min(dev, m.last_fill_ms()) if rep else m.last_fill_ms()
    except lmrs_amd.LmrsError:
        dev = float("nan")
best = min(out.values())
mode = "token-by-token" if os.environ.get("LMRS_NO_BATCHED_PREFILL") else "batched (int8 MFMA)"
cfg = S.CONFIGS[model]
macs = n * cfg.n_layers * (cfg.dim * (cfg.n_heads * cfg.head_size + 2 * cfg.n_kv_heads * cfg.head_size) + cfg.n_heads * cfg.head_size * cfg.dim + 3 * cfg.dim * cfg.hidden_dim)
print(f"{model} {'Q4_0' if qt == S.Q4_0 else 'Q8_0'} fill_kv_cache({n} tokens) {mode}: {best*1e3:.2f} ms = {n/best:.0f} tok/s, {2*macs/best/1e12:.1f} int8 TOP/s (host<->device copies of the embeddings included); "
      f"on the device alone {dev:.3f} ms = {2*macs/(dev*1e-3)/1e12:.1f} int8 TOP/s")
print("checksum", float(np.abs(e).sum()))
