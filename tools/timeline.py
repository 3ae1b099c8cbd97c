"""Per-kernel timeline of one decode step from the in-kernel wall-clock stamps (LMRS_DEBUG_TIMELINE=1).
usage: LMRS_DEBUG_TIMELINE=1 python tools/timeline.py [model] [pos] [q8_0|q4_0]"""
import os
import sys

os.environ["LMRS_DEBUG_TIMELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import lmrs_amd  # noqa: E402
from tools import synth_lmrs as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
npos = int(sys.argv[2]) if len(sys.argv) > 2 else 100
qt = S.Q4_0 if len(sys.argv) > 3 and sys.argv[3] == "q4_0" else S.Q8_0
img = S.build_image(model, qt, 1234)
m = lmrs_amd.Transformer(img)
prompt = S.prompt_tokens(model, 16, 1234)
toks, sec = m.generate_greedy(prompt, npos - 15, timing=True)
print(f"{(16 + npos - 16) / sec:.0f} tok/s over {npos} steps")
tl = m.debug_timeline().astype(np.int64)
names = ["qkv", "attn", "wo", "w13", "w2"]
NK = len(names)
t0 = tl[0, 0]
print("node  name   start  | first WG: pro   pass1  end   | last WG: start pro pass1 end | gap_to_next   (us, 10ns clock)")
TAIL = 1 if (len(tl) - 1) % NK == 0 else 2          # 1: the final argmax is folded into the classifier launch (no node of its own)
L = (len(tl) - TAIL) // NK
tot = {}
for i, r in enumerate(tl):
    name = names[i % NK] if i < NK * L else ("cls" if i == NK * L else "argmax")
    f = (r[:4] - r[0]) / 100.0
    l = (r[4:] - r[0]) / 100.0
    end = max(r[3], r[7])
    nxt = tl[i + 1, 0] if i + 1 < len(tl) else end
    dur = (end - r[0]) / 100.0
    tot.setdefault(name, []).append((dur, (nxt - end) / 100.0, f[1], f[2] - f[1]))
    if i < 10 or i >= len(tl) - 3:
        print(f"{i:3d} {name:6s} {(r[0]-t0)/100.0:8.2f} | {f[1]:6.2f} {f[2]:6.2f} {f[3]:6.2f} | {l[0]:6.2f} {l[1]:6.2f} {l[2]:6.2f} {l[3]:6.2f} | {(nxt-end)/100.0:6.2f}")
print("\nmean per kernel type: duration(us)  gap_after(us)  prologue(us)  first-pass(us)")
for k, v in tot.items():
    a = np.array(v)
    print(f"  {k:7s} n={len(v):3d}  {a[:,0].mean():7.2f} {a[:,1].mean():7.2f} {a[:,2].mean():7.2f} {a[:,3].mean():7.2f}")
att = tl[1::5][:L] if NK == 5 else tl[0::NK][:L]
d = (att - att[:, :1]) / 100.0
print("attention block0 stamps (us from start): rope-inputs-issued, rope-done, scores-done, softmax-done, v-in-lds, end")
print("   ", np.round(d[:, [1, 2, 4, 5, 6, 7]].mean(axis=0), 2))
if NK == 5:
    # merged qkv + attention launch: the attention node's stamps belong to head 0's workgroup of the SAME launch as the qkv node
    q0 = tl[0::5][:L][:, :1]
    dm = (att - q0) / 100.0
    print("merged launch, head 0 (us from the qkv node's start; workgroup form: -, polled, rope, -, scores, softmax, v-in-lds, end;")
    print("   wave form: start, prefetch issued, polled, rope, scores, softmax, -, end)")
    print("   ", np.round(dm.mean(axis=0), 2))
    qk = tl[0::5][:L]
    print("   qkv first workgroup end:", np.round(((qk[:, 3] - qk[:, 0]) / 100.0).mean(), 2), " last workgroup end:", np.round(((qk[:, 7] - qk[:, 0]) / 100.0).mean(), 2))
for nm, off in (("wo", 2), ("w2", 4)):
    if NK == 5:
        g = tl[off::5][:L]; d = (g - g[:, :1]) / 100.0
        print(f"{nm} block0 (us from start): inputs landed+absmax {d[:,6].mean():.2f}  quantised(thread0) {d[:,7].mean():.2f}  barrier {d[:,1].mean():.2f}  rows done {d[:,2].mean():.2f}  end {d[:,3].mean():.2f}")
for nm, off in (("qkv", 0), ("w13", 3)):
    if NK == 5:
        g = tl[off::5][:L]; d = (g - g[:, :1]) / 100.0
        print(f"{nm} block0 (us from start): x landed+squares {d[:,4].mean():.2f}  chain done {d[:,5].mean():.2f}  quantised {d[:,1].mean():.2f}  rows done {d[:,2].mean():.2f}  end {d[:,3].mean():.2f}")
am = tl[-1]
if TAIL == 1:
    print("classifier launch (us from its start): first GEMV workgroup prologue / first pass / end:", np.round((am[1:4] - am[0]) / 100.0, 2),
          " consumer workgroup start / all partials seen / end:", np.round((am[[4, 5, 7]] - am[0]) / 100.0, 2))
if TAIL == 2:
    print(f"shader clock during argmax kernel: {(am[2]-am[1]) / ((am[3]-am[0]) / 100.0):.0f} MHz")
print(f"step span: {(max(tl[-1,3], tl[-1,7]) - t0)/100.0:.1f} us")
