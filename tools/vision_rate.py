"""Time of VisionTransformer::forward (reference src/vision.rs:244-577) and PHI3VProcessor::forward (src/processor.rs:234-342)
on the device, full CLIP ViT-L/14-336 depth.  usage: python tools/vision_rate.py [num_crops] [n_layers]
(num_crops = 1 global + w_crop x 1 sub-images)"""
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

if crops >= 2:
    psec = V.build_processor_section()
    pr = lmrs_amd.PHI3VProcessor(psec)
    w_crop, h_crop = crops - 1, 1
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); emb = pr.forward(out, 576 * cfg.dim, 12, w_crop, h_crop); best = min(best, time.perf_counter() - t0)
    pm = emb.shape[0] * (4096 * 3072 + 3072 * 3072)
    print(f"image projector, {emb.shape[0]} embeddings 4096 -> 3072 -> 3072: {best*1e3:.2f} ms ({2*pm/best/1e12:.1f} int8 TOP/s; host HD transform "
          f"and copies included); checksum {float(np.abs(emb).sum()):.3f}")
