"""Per-token rates of the three ways to drive the decode path (run on the GPU box):
   generate_greedy (device-resident loop)  |  forward_argmax (one host round trip per token, 4 B back)  |
   forward (reference-shaped API: 513 KB of logits cross PCIe every token)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import lmrs_amd  # noqa: E402
from tools import synth_lmrs as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
img = S.build_image(model, S.Q8_0, 1234)
m = lmrs_amd.Transformer(img)
prompt = S.prompt_tokens(model, 16, 1234)
first = m.generate_greedy(prompt, 1)
N = 128
toks, sec = m.generate_greedy(first, N, start_pos=16, timing=True)
print(f"generate_greedy : {N / sec:8.1f} tok/s  ({sec / N * 1e6:.1f} us/token, device events)")
t = int(first[0]); t0 = time.perf_counter()
for i in range(N):
    t = m.forward_argmax(t, 16 + i)
dt = time.perf_counter() - t0
print(f"forward_argmax  : {N / dt:8.1f} tok/s  ({dt / N * 1e6:.1f} us/token, host wall clock)")
t = int(first[0]); t0 = time.perf_counter()
for i in range(N):
    lg = m.forward(t, 16 + i); t = int(np.argmax(lg))
dt = time.perf_counter() - t0
print(f"forward+argmax  : {N / dt:8.1f} tok/s  ({dt / N * 1e6:.1f} us/token, host wall clock, logits over PCIe + numpy argmax)")
