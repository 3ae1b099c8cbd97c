"""Synthetic vision section of an LMRS multimodal file (reference export.py:126-170, read back by
src/vision.rs:99-243): CLIP ViT-L/14-336 shapes (the reference hard-codes 577 positions, vision.rs:117, and C = 1024, H = 24 in
the processor), Q8_0 / Q4_0 / f32, any number of layers.  Same conventions as tools/synth_lmrs.py: seeded numpy generator per tensor, weights
quantised with the restated exporter quantiser."""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

from tools.synth_lmrs import Q4_0, Q8_0, Q_NONE, quantize_q40, quantize_q80


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


VIS_PREFIX = "model.vision_embed_tokens.img_processor.vision_model."


def vision_tensors(cfg: VisionCfg = VisionCfg(), seed: int = 99):
    """The float tensors of the tower in the order export.py writes them (:126-151): a list of
    (Hugging Face key suffix with `{l}` for per-layer tensors, [array per layer] or [array], quantised?)."""
    L, dim, hid = cfg.n_layers, cfg.dim, cfg.hidden_dim
    rng = np.random.default_rng(seed)

    def f32(shape, sigma, base=0.0):
        return (base + sigma * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)

    def per_layer_rows(a):                          # one [L, n] draw -> L vectors
        return [a[l] for l in range(L)]

    t = []
    t.append(("embeddings.class_embedding", [f32((dim,), 0.05)], False))
    t.append(("embeddings.patch_embedding.weight", [f32((dim, 3 * cfg.patch_size * cfg.patch_size), 0.03).reshape(dim, 3, cfg.patch_size, cfg.patch_size)], False))
    t.append(("embeddings.position_embedding.weight", [f32((N_POS, dim), 0.05)], False))
    t.append(("encoder.layers.{l}.layer_norm1.weight", per_layer_rows(f32((L, dim), 0.1, 1.0)), False))
    t.append(("encoder.layers.{l}.layer_norm1.bias", per_layer_rows(f32((L, dim), 0.05)), False))
    t.append(("encoder.layers.{l}.layer_norm2.weight", per_layer_rows(f32((L, dim), 0.1, 1.0)), False))
    t.append(("encoder.layers.{l}.layer_norm2.bias", per_layer_rows(f32((L, dim), 0.05)), False))
    for name, shape, sigma, nb in (("self_attn.q_proj", (dim, dim), 0.03, dim), ("self_attn.k_proj", (dim, dim), 0.03, dim), ("self_attn.v_proj", (dim, dim), 0.03, dim),
                                   ("self_attn.out_proj", (dim, dim), 0.03, dim), ("mlp.fc1", (hid, dim), 0.03, hid), ("mlp.fc2", (dim, hid), 0.02, dim)):
        t.append((f"encoder.layers.{{l}}.{name}.weight", [f32(shape, sigma) for _ in range(L)], True))
        t.append((f"encoder.layers.{{l}}.{name}.bias", per_layer_rows(f32((L, nb), 0.05)), False))
    t.append(("pre_layrnorm.weight", [f32((dim,), 0.1, 1.0)], False))
    t.append(("pre_layrnorm.bias", [f32((dim,), 0.05)], False))
    return t


def _section(header: bytes, tensors, gs: int, q_type: int = Q8_0) -> np.ndarray:
    parts = [np.frombuffer(header, np.uint8)]
    for _, arrays, quant in tensors:
        for a in arrays:
            if quant and q_type != Q_NONE:
                q, sc = (quantize_q80 if q_type == Q8_0 else quantize_q40)(np.ascontiguousarray(a, np.float32).reshape(a.shape[0], -1), gs)
                parts.append(q.view(np.uint8).reshape(-1)); parts.append(np.ascontiguousarray(sc, np.float32).reshape(-1).view(np.uint8))
            else:
                parts.append(np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint8))
    return np.concatenate(parts)


def build_vision_section(cfg: VisionCfg = VisionCfg(), seed: int = 99, gs: int = 128, q_type: int = Q8_0) -> np.ndarray:
    """-> uint8 array: 128-byte header + tensors in the order VisionTransformer::new reads them (Q8_0 / Q4_0 / plain f32)."""
    return _section(vision_header(cfg, q_type, gs), vision_tensors(cfg, seed), gs, q_type)


def pixel_values(cfg: VisionCfg, num_crops: int, seed: int = 7) -> np.ndarray:
    """Normalised patches as PHI3VProcessor::process hands them to the tower (processor.rs view_as_patches):
    [num_crops][576 patches][3 * 14 * 14] f32."""
    n = (cfg.image_size // cfg.patch_size) ** 2
    return np.random.default_rng(seed).standard_normal((num_crops, n, 3 * cfg.patch_size ** 2), dtype=np.float32)


def processor_tensors(hidden_dim: int = 4096, text_dim: int = 3072, seed: int = 123):
    """glb_GN, sub_GN, the projector MLP, in the order export.py writes them (:166-169); keys relative to model.vision_embed_tokens."""
    rng = np.random.default_rng(seed)

    def f32(shape, sigma):
        return (sigma * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)

    glb, sub = f32((hidden_dim,), 0.5), f32((hidden_dim,), 0.5)
    w0, w1 = f32((text_dim, hidden_dim), 0.02), f32((text_dim, text_dim), 0.02)
    b0, b1 = f32((text_dim,), 0.05), f32((text_dim,), 0.05)
    return [("glb_GN", [glb.reshape(1, 1, -1)], False), ("sub_GN", [sub.reshape(1, 1, 1, -1)], False),
            ("img_projection.0.weight", [w0], True), ("img_projection.2.weight", [w1], True),
            ("img_projection.0.bias", [b0], False), ("img_projection.2.bias", [b1], False)]


def processor_header(hidden_dim: int, text_dim: int, q_type: int = Q8_0, gs: int = 128) -> bytes:
    h = struct.pack("II", hidden_dim, text_dim) + struct.pack("B", q_type) + struct.pack("I", gs)
    assert len(h) == 13
    return h + b"\0" * (128 - len(h))


def build_processor_section(hidden_dim: int = 4096, text_dim: int = 3072, seed: int = 123, gs: int = 128, q_type: int = Q8_0) -> np.ndarray:
    """Processor section (export.py:155-170, read back by src/processor.rs:168-232): 13-byte header padded to 128, glb_GN,
    sub_GN, the two projector matrices (Q8_0), their biases."""
    return _section(processor_header(hidden_dim, text_dim, q_type, gs), processor_tensors(hidden_dim, text_dim, seed), gs, q_type)


def hf_vision_state_dict(cfg: VisionCfg, hidden_dim: int, text_dim: int, vseed: int, pseed: int) -> dict:
    """The same float tensors under the Hugging Face names export.py looks for (Phi-3.5-vision checkpoint layout)."""
    sd = {}
    for key, arrays, _ in vision_tensors(cfg, vseed):
        for l, a in enumerate(arrays):
            sd[VIS_PREFIX + key.format(l=l)] = a
    for key, arrays, _ in processor_tensors(hidden_dim, text_dim, pseed):
        sd["model.vision_embed_tokens." + key] = arrays[0]
    return sd


def hf_vision_config(cfg: VisionCfg) -> dict:
    return {"vision_config": {"hidden_size": cfg.dim, "intermediate_size": cfg.hidden_dim, "num_hidden_layers": cfg.n_layers,
                              "num_attention_heads": cfg.n_heads, "layer_norm_eps": cfg.layernorm_eps, "patch_size": cfg.patch_size,
                              "image_size": cfg.image_size}}
