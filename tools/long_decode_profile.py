"""rocprofv3 --kernel-trace --stats -- python tools/long_decode_profile.py: kernel statistics of 64 decode steps at positions 1024..1087 (Llama-3.2-1B Q8_0;
the prompt goes through the batched prefill, whose kernels appear in the same table)."""
import os, sys
sys.path.insert(0, os.getcwd())
import lmrs_amd
from tools import synth_lmrs as S
img = S.build_image("llama-3.2-1b", S.Q8_0, 1234)
m = lmrs_amd.Transformer(img)
p = S.prompt_tokens("llama-3.2-1b", 1024, 7)
out, sec = m.generate_greedy(p, 65, timing=True)
print("prompt 1024 (batched prefill) + 64 decode steps at positions 1024..1087:", sec)
