"""Per-kernel timeline of one decode step from the in-kernel wall-clock stamps (LMRS_DEBUG_TIMELINE=1).
usage: LMRS_DEBUG_TIMELINE=1 python tools/timeline.py [model] [pos] [q8_0|q4_0]"""
import os
import sys

os.environ["LMRS_DEBUG_TIMELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import warnings  # noqa: E402
import numpy as np  # noqa: E402
warnings.filterwarnings("ignore", category=RuntimeWarning)   # (means over stamps a kernel form never writes)
import lmrs_amd  # noqa: E402
from tools import synth_lmrs as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
npos = int(sys.argv[2]) if len(sys.argv) > 2 else 100
qt = S.Q4_0 if len(sys.argv) > 3 and sys.argv[3] == "q4_0" else S.Q8_0
img = S.build_image(model, qt, 1234)
m = lmrs_amd.Transformer(img)
prompt = S.prompt_tokens(model, 16, 1234)
toks, sec = m.generate_greedy(prompt, npos - 15, timing=True)
print(f"run: 16 prompt + {npos - 16} generated positions in {sec * 1e3:.2f} ms of device time = {npos / sec:.0f} steps/s (stamped kernels: ~1 us slower per launch; "
      f"the table below is the LAST step, at position {npos - 1})")
tl = m.debug_timeline().astype(np.float64)
tl[tl == 0] = np.nan                                     # a stamp the kernel form in use never writes
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
    end = np.nanmax([r[3], r[7]])
    nxt = tl[i + 1, 0] if i + 1 < len(tl) else end
    dur = (end - r[0]) / 100.0
    tot.setdefault(name, []).append((dur, (nxt - end) / 100.0, f[1], f[2] - f[1]))
    if i < 10 or i >= len(tl) - 3:
        fm = lambda x: "     -" if np.isnan(x) else f"{x:6.2f}"
        print(f"{i:3d} {name:6s} {(r[0]-t0)/100.0:8.2f} | {fm(f[1])} {fm(f[2])} {fm(f[3])} | {fm(l[0])} {fm(l[1])} {fm(l[2])} {fm(l[3])} | {fm((nxt-end)/100.0)}")
print("\nmean per kernel type: duration(us)  gap_after(us)  prologue(us)  first-pass(us)")
for k, v in tot.items():
    a = np.array(v)
    mm = np.nanmean(a, axis=0)
    print(f"  {k:7s} n={len(v):3d}  {mm[0]:7.2f} {mm[1]:7.2f} {mm[2]:7.2f} {mm[3]:7.2f}")
att = tl[1::5][:L] if NK == 5 else tl[0::NK][:L]
d = (att - att[:, :1]) / 100.0
print("attention block0 stamps (us from start): rope-inputs-issued, rope-done, scores-done, softmax-done, v-in-lds, end")
print("   ", np.round(np.nanmean(d[:, [1, 2, 4, 5, 6, 7]], axis=0), 2))
if NK == 5:
    # merged qkv + attention launch: the attention node's stamps belong to head 0's workgroup of the SAME launch as the qkv node
    q0 = tl[0::5][:L][:, :1]
    dm = (att - q0) / 100.0
    print("merged launch, head 0 (us from the qkv node's start; workgroup form: -, polled, rope, -, scores, softmax, v-in-lds, end;")
    print("   wave form: start, prefetch issued, polled, rope, scores, softmax, -, end)")
    print("   ", np.round(np.nanmean(dm, axis=0), 2))
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
print(f"step span: {(np.nanmax([tl[-1,3], tl[-1,7]]) - t0)/100.0:.1f} us")
