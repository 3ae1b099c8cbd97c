"""Sampler::sample (reference src/sampler.rs:109-129, temperature != 0, sample_mult) after a decode step: on the device (lmrs_forward_sample:
the logits never leave HBM; the two sequential chains over the vocabulary run lane by lane in one wave) against the host
(lmrs_forward copies the 513 KB of logits to pinned memory, lmrs_sampler_sample runs the reference's loops on them).
usage: python tools/sampler_rate.py [model]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lmrs_amd  # noqa: E402
from tools import synth_lmrs as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
img = S.build_image(model, S.Q8_0, 1234)
m = lmrs_amd.Transformer(img)
prompt = S.prompt_tokens(model, 8, 1234)
cases = [(0.0, 0.9, "greedy (argmax fused into the step)"), (0.8, 1.0, "temperature 0.8, sample_mult"),
         (0.7, 0.9, "temperature 0.7, top-p 0.9 (the reference's defaults; synthetic weights: a nearly flat distribution, most of the vocabulary passes the cutoff)"),
         (0.02, 0.9, "temperature 0.02, top-p 0.9 (a peaked distribution, as a trained model's: a handful of candidates)")]
for t, top_p, name in cases:
    s = lmrs_amd.Sampler(m.args.vocab_size, t, top_p, 99)
    for pos, tk in enumerate(prompt):
        m.forward_argmax(int(tk), pos)
    tok = int(prompt[-1]); N = 64
    t0 = time.perf_counter()
    for i in range(N):
        tok = m.forward_sample(tok, 8 + i, s)
    dev = (time.perf_counter() - t0) / N
    tok = int(prompt[-1])
    t0 = time.perf_counter()
    for i in range(N):
        tok = s.sample(m.forward(tok, 8 + i))
    host = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for i in range(N):
        m.forward_argmax(tok, 8 + i)
    step = (time.perf_counter() - t0) / N
    print(f"{model} {name}: step alone {step*1e6:.0f} us; step + sample on the device {dev*1e6:.0f} us; step + logits to the host + host sampler {host*1e6:.0f} us")
