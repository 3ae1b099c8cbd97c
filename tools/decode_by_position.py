"""Decode step time by context position (run on the GPU box): greedy generation of 2000 tokens in segments of 32 steps, device events per
segment - where the attention forms switch (one / two waves per head up to 128 positions for 64-wide heads, one workgroup per head up to
LMRS_ATT_SPLIT_POS = 384, the two-launch split attention beyond) and what a step costs deep into a context.
    python tools/decode_by_position.py [model] [qtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lmrs_amd  # noqa: E402
from tools import synth_lmrs as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
qtype = {"q8_0": S.Q8_0, "q4_0": S.Q4_0}[sys.argv[2] if len(sys.argv) > 2 else "q8_0"]
img = S.build_image(model, qtype, 1234)
m = lmrs_amd.Transformer(img)
tok = m.generate_greedy(S.prompt_tokens(model, 8, 1234), 1)
pos, seg = 8, 32
print(f"{model}: us per decode step by position (segments of {seg} steps, device events)")
for _ in range(2):                                      # the second sweep: every graph captured
    pos, rows = 8, []
    t = tok
    while pos + seg <= min(2048, m.args.seq_len) - 1:
        out, sec = m.generate_greedy(t, seg, start_pos=pos, timing=True)
        rows.append((pos, sec / seg * 1e6))
        t = out[-1:]; pos += seg
for p, us in rows:
    if p < 520 or p % 256 == 8:
        print(f"  positions {p:5d}..{p + seg - 1:5d}: {us:7.1f} us per step")
