"""Time of VisionTransformer::forward (reference src/vision.rs:244-577) on the device, full CLIP ViT-L/14-336 depth.
usage: python tools/vision_rate.py [num_crops] [n_layers]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import lmrs_amd  # noqa: E402
from tools import synth_vision as V  # noqa: E402

crops = int(sys.argv[1]) if len(sys.argv) > 1 else 2
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = V.VisionCfg(n_layers=layers)
sec = V.build_vision_section(cfg)
m = lmrs_amd.VisionTransformer(sec)
pv = V.pixel_values(cfg, crops)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); out = m.forward(pv, crops); best = min(best, time.perf_counter() - t0)
ntok = crops * 577
macs = ntok * (layers - 1) * (4 * cfg.dim * cfg.dim + 2 * cfg.dim * cfg.hidden_dim)
print(f"CLIP ViT-L/14-336 tower, {crops} crops x 577 tokens x {layers - 1} layers: {best*1e3:.1f} ms "
      f"({2*macs/best/1e12:.1f} int8 TOP/s in the projections; host<->device copies included); checksum {float(np.abs(out).sum()):.3f}")
