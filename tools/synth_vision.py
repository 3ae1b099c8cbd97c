"""Synthetic vision section of an LMRS multimodal file (reference export.py:126-170, read back by
src/vision.rs:99-243): CLIP ViT-L/14-336 shapes (the reference hard-codes 577 positions, vision.rs:117, and C = 1024, H = 24 in
the processor), Q8_0, any number of layers.  Same conventions as tools/synth_lmrs.py: seeded numpy generator per tensor, weights
quantised with the restated exporter quantiser."""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

from tools.synth_lmrs import Q8_0, quantize_q80


@dataclass(frozen=True)
class VisionCfg:
    dim: int = 1024
    hidden_dim: int = 4096
    n_layers: int = 24          # the forward pass runs n_layers - 1 of them (vision.rs:303)
    n_heads: int = 16
    head_size: int = 64
    layernorm_eps: float = 1e-5
    patch_size: int = 14
    image_size: int = 336


N_POS = 577


def vision_header(cfg: VisionCfg, q_type: int = Q8_0, gs: int = 128) -> bytes:
    h = struct.pack("IIIIIfII", cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.head_size, cfg.layernorm_eps, cfg.patch_size, cfg.image_size)
    h += struct.pack("B", q_type) + struct.pack("I", gs)
    assert len(h) == 37
    return h + b"\0" * (128 - len(h))


def build_vision_section(cfg: VisionCfg = VisionCfg(), seed: int = 99, gs: int = 128) -> np.ndarray:
    """-> uint8 array: 128-byte header + tensors in the order VisionTransformer::new reads them (Q8_0)."""
    L, dim, hid = cfg.n_layers, cfg.dim, cfg.hidden_dim
    kdim = 3 * cfg.patch_size * cfg.patch_size
    rng = np.random.default_rng(seed)
    parts = [np.frombuffer(vision_header(cfg, Q8_0, gs), np.uint8)]

    def f32(shape, sigma, base=0.0):
        return (base + sigma * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)

    def put_f32(a):
        parts.append(np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint8))

    def put_quant(per_layer_shape, sigma):
        for _ in range(L):
            q, s = quantize_q80(f32(per_layer_shape, sigma), gs)
            parts.append(q.view(np.uint8).reshape(-1)); parts.append(np.ascontiguousarray(s, np.float32).reshape(-1).view(np.uint8))

    put_f32(f32((dim,), 0.05))                       # class_embedding
    put_f32(f32((dim, kdim), 0.03))                  # patch_embedding.weight [dim][3*14*14]
    put_f32(f32((N_POS, dim), 0.05))                 # position_embedding.weight
    put_f32(f32((L, dim), 0.1, 1.0)); put_f32(f32((L, dim), 0.05))      # layer_norm1 w, b
    put_f32(f32((L, dim), 0.1, 1.0)); put_f32(f32((L, dim), 0.05))      # layer_norm2 w, b
    for _ in range(3):                               # q, k, v
        put_quant((dim, dim), 0.03); put_f32(f32((L, dim), 0.05))
    put_quant((dim, dim), 0.03); put_f32(f32((L, dim), 0.05))           # out_proj
    put_quant((hid, dim), 0.03); put_f32(f32((L, hid), 0.05))           # fc1
    put_quant((dim, hid), 0.02); put_f32(f32((L, dim), 0.05))           # fc2
    put_f32(f32((dim,), 0.1, 1.0)); put_f32(f32((dim,), 0.05))          # pre_layrnorm w, b
    return np.concatenate(parts)


def pixel_values(cfg: VisionCfg, num_crops: int, seed: int = 7) -> np.ndarray:
    """Normalised patches as PHI3VProcessor::process hands them to the tower (processor.rs view_as_patches):
    [num_crops][576 patches][3 * 14 * 14] f32."""
    n = (cfg.image_size // cfg.patch_size) ** 2
    return np.random.default_rng(seed).standard_normal((num_crops, n, 3 * cfg.patch_size ** 2), dtype=np.float32)


def build_processor_section(hidden_dim: int = 4096, text_dim: int = 3072, seed: int = 123, gs: int = 128) -> np.ndarray:
    """Processor section (export.py:155-170, read back by src/processor.rs:168-232): 13-byte header padded to 128, glb_GN,
    sub_GN, the two projector matrices (Q8_0), their biases."""
    rng = np.random.default_rng(seed)
    h = struct.pack("II", hidden_dim, text_dim) + struct.pack("B", Q8_0) + struct.pack("I", gs)
    assert len(h) == 13
    parts = [np.frombuffer(h + b"\0" * (128 - len(h)), np.uint8)]

    def f32(shape, sigma):
        return (sigma * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)

    parts.append(f32((hidden_dim,), 0.5).view(np.uint8))                        # glb_GN
    parts.append(f32((hidden_dim,), 0.5).view(np.uint8))                        # sub_GN
    for shape, sigma in (((text_dim, hidden_dim), 0.02), ((text_dim, text_dim), 0.02)):
        q, s = quantize_q80(f32(shape, sigma), gs)
        parts.append(q.view(np.uint8).reshape(-1)); parts.append(np.ascontiguousarray(s, np.float32).reshape(-1).view(np.uint8))
    parts.append(f32((text_dim,), 0.05).view(np.uint8)); parts.append(f32((text_dim,), 0.05).view(np.uint8))
    return np.concatenate(parts)
