"""Synthetic LMRS v4 model images at real shapes (no checkpoints are reachable offline).

Writes exactly the byte layout the reference exporter produces (reference export.py:51-126,
utils/io.py:21-56) with the reference's weight quantisers restated in numpy
(utils/quantization.py:4-39 quantize_q40, :42-66 quantize_q80; torch.round == np.rint).
tests/test_format.py checks this writer byte-for-byte against files produced by the reference's
own export.py from the same float tensors (tests/golden/, made by tests/golden/make_golden.py).

Determinism: every (tensor family, layer, row-chunk) draws from its own PCG64 stream derived from
(seed, family index, layer, chunk), so the bytes do not depend on thread count or chunking order.
"""
from __future__ import annotations

import struct
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, asdict
import os

import numpy as np

GEMMA, LLAMA, PHI = 0, 1, 2
Q_NONE, Q8_0, Q4_0 = 0, 1, 2


@dataclass(frozen=True)
class ModelCfg:
    name: str
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    head_size: int
    n_kv_heads: int
    vocab_size: int
    max_pos: int
    rms_norm_eps: float
    rope_theta: float
    model_type: int

    @property
    def att_dim(self):
        return self.n_heads * self.head_size

    @property
    def kv_dim(self):
        return self.n_kv_heads * self.head_size


CONFIGS = {
    # BASELINE.json configs[0..1] (the north-star model)
    "llama-3.2-1b": ModelCfg("llama-3.2-1b", 2048, 8192, 16, 32, 64, 8, 128256, 131072, 1e-5, 500000.0, LLAMA),
    # configs[3]
    "llama-3.2-3b": ModelCfg("llama-3.2-3b", 3072, 8192, 28, 24, 128, 8, 128256, 131072, 1e-5, 500000.0, LLAMA),
    # configs[2]
    "gemma-2-2b": ModelCfg("gemma-2-2b", 2304, 9216, 26, 8, 256, 4, 256000, 8192, 1e-6, 10000.0, GEMMA),
    # configs[4] (text part)
    "phi-3.5": ModelCfg("phi-3.5", 3072, 8192, 32, 32, 96, 32, 32064, 131072, 1e-5, 10000.0, PHI),
    # small shapes for CPU-side tests / golden fixtures
    "tiny-llama": ModelCfg("tiny-llama", 128, 256, 2, 2, 64, 1, 256, 64, 1e-5, 500000.0, LLAMA),
    "tiny-gemma": ModelCfg("tiny-gemma", 128, 256, 2, 2, 64, 1, 256, 64, 1e-6, 10000.0, GEMMA),
    "tiny-phi": ModelCfg("tiny-phi", 128, 256, 2, 4, 96, 4, 256, 64, 1e-5, 10000.0, PHI),
    # mid-size: real head geometry of the 1B model, few layers, small vocab (fast oracle runs)
    "mini-llama": ModelCfg("mini-llama", 2048, 8192, 2, 32, 64, 8, 4096, 256, 1e-5, 500000.0, LLAMA),
    # a vocabulary that is not a multiple of 4: its last two logits are never written by the reference (functional.rs:183)
    "mini-llama-v4102": ModelCfg("mini-llama-v4102", 2048, 8192, 2, 32, 64, 8, 4102, 256, 1e-5, 500000.0, LLAMA),
    # a vocabulary of several sort blocks: the device sort of top-p candidates (lmrs_forward_sample) runs its global merge steps
    "mini-llama-v40k": ModelCfg("mini-llama-v40k", 2048, 8192, 1, 32, 64, 8, 40000, 64, 1e-5, 500000.0, LLAMA),
    "mini-llama-long": ModelCfg("mini-llama-long", 2048, 8192, 2, 32, 64, 8, 4096, 2048, 1e-5, 500000.0, LLAMA),
    "mini-phi-long": ModelCfg("mini-phi-long", 3072, 8192, 2, 32, 96, 32, 4096, 1024, 1e-5, 10000.0, PHI),
    "mini-llama3b": ModelCfg("mini-llama3b", 3072, 8192, 2, 24, 128, 8, 4096, 256, 1e-5, 500000.0, LLAMA),
    "mini-gemma": ModelCfg("mini-gemma", 2304, 9216, 2, 8, 256, 4, 4096, 256, 1e-6, 10000.0, GEMMA),
    "mini-phi": ModelCfg("mini-phi", 3072, 8192, 2, 32, 96, 32, 4096, 256, 1e-5, 10000.0, PHI),
    # Gemma-2-9B geometry: n_heads * head_size (4096) > dim (3584), the case of transformer.rs:497-499
    "mini-gemma9b": ModelCfg("mini-gemma9b", 3584, 14336, 2, 16, 256, 8, 4096, 256, 1e-6, 10000.0, GEMMA),
    "mini-llama8b": ModelCfg("mini-llama8b", 4096, 14336, 2, 32, 128, 8, 4096, 256, 1e-5, 500000.0, LLAMA),
    "tiny-wide-att": ModelCfg("tiny-wide-att", 256, 512, 2, 4, 128, 2, 512, 64, 1e-6, 10000.0, GEMMA),
}


# ------------------------------------------------------------------ reference quantisers, in numpy
def quantize_q80(w: np.ndarray, gs: int):
    """utils/quantization.py:42-66."""
    w = w.astype(np.float32, copy=False).reshape(-1, gs)
    wmax = np.abs(w).max(axis=1)
    scale = (wmax / np.float32(127.0)).astype(np.float32)
    q = np.rint(w / scale[:, None]).astype(np.int8)
    return q.reshape(-1), scale


def quantize_q40(w: np.ndarray, gs: int):
    """utils/quantization.py:4-39 (weights: scale = wmax / -7.5, nibbles 0..15, even index -> low nibble)."""
    w = w.astype(np.float32, copy=False).reshape(-1, gs)
    wmax = np.abs(w).max(axis=1)
    scale = (wmax / np.float32(-7.5)).astype(np.float32)
    u = np.clip(np.rint(w / scale[:, None] + np.float32(8.0)), 0, 15).astype(np.uint8)
    u = u.reshape(u.shape[0], gs // 2, 2)
    packed = (u[..., 0] | (u[..., 1] << 4)).astype(np.uint8)
    return packed.reshape(-1), scale


# ------------------------------------------------------------------ tensor families in file order
def families(cfg: ModelCfg):
    """(name, kind, per-layer?, rows, cols, sigma) in the order transformer.rs:241-270 reads them."""
    d, h, a, kv, V = cfg.dim, cfg.hidden_dim, cfg.att_dim, cfg.kv_dim, cfg.vocab_size
    gem = cfg.model_type == GEMMA
    fam = [
        ("embed_tokens", "w", False, V, d, 0.02),
        ("input_layernorm", "n", True, 1, d, 0.0),
        ("q_proj", "w", True, a, d, 0.03),
        ("k_proj", "w", True, kv, d, 0.03),
        ("v_proj", "w", True, kv, d, 0.03),
        ("o_proj", "w", True, d, a, 0.03),
        ("post_attention_layernorm", "n", True, 1, d, 0.0),
    ]
    if gem:
        fam.append(("pre_feedforward_layernorm", "n", True, 1, d, 0.0))
    fam += [
        ("gate_proj", "w", True, h, d, 0.03),
        ("down_proj", "w", True, d, h, 0.02),
        ("up_proj", "w", True, h, d, 0.03),
    ]
    if gem:
        fam.append(("post_feedforward_layernorm", "n", True, 1, d, 0.0))
    fam.append(("norm", "n", False, 1, d, 0.0))
    if cfg.model_type == PHI:
        fam.append(("lm_head", "w", False, V, d, 0.02))
    return fam


def _rng(seed, fam_idx, layer, chunk):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, fam_idx, layer, chunk])))


ROWS_PER_CHUNK = 4096


def float_tensor(cfg: ModelCfg, seed: int, fam_idx: int, layer: int, row0: int = 0, rows: int | None = None):
    """The f32 'master' tensor slice (rows row0..row0+rows) of a family/layer; chunked by ROWS_PER_CHUNK rows."""
    name, kind, _, R, C, sigma = families(cfg)[fam_idx]
    rows = R - row0 if rows is None else rows
    assert row0 % ROWS_PER_CHUNK == 0
    out = np.empty((rows, C), np.float32)
    r = 0
    while r < rows:
        c = (row0 + r) // ROWS_PER_CHUNK
        take = min(ROWS_PER_CHUNK, rows - r)
        g = _rng(seed, fam_idx, layer, c)
        if kind == "n":
            base = 0.0 if cfg.model_type == GEMMA else 1.0      # Gemma kernels add 1 (functional.rs:68-70)
            out[r:r + take] = (base + 0.1 * g.standard_normal((take, C), dtype=np.float32)).astype(np.float32)
        else:
            out[r:r + take] = g.standard_normal((take, C), dtype=np.float32) * np.float32(sigma)
        r += take
    return out


def header_bytes(cfg: ModelCfg, q_type: int, gs: int, multimodal: int = 0) -> bytes:
    """export.py:54-84."""
    h = struct.pack("II", 0x73726D6C, 4)
    h += struct.pack("IIIIIIIIff", cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.head_size,
                     cfg.n_kv_heads, cfg.vocab_size, cfg.max_pos, cfg.rms_norm_eps, cfg.rope_theta)
    h += struct.pack("BB", q_type, cfg.model_type)
    h += struct.pack("I", gs)
    h += struct.pack("B", multimodal)
    return h + b"\0" * (256 - len(h))


def image_size(cfg: ModelCfg, q_type: int, gs: int = 128) -> int:
    n = 256
    for name, kind, per_layer, R, C, _ in families(cfg):
        cnt = (cfg.n_layers if per_layer else 1) * R * C
        if kind == "n" or q_type == Q_NONE:
            n += cnt * 4
        else:
            n += (cnt // 2 if q_type == Q4_0 else cnt) + cnt // gs * 4
    return n


def build_image(cfg: ModelCfg | str, q_type: int = Q8_0, seed: int = 1234, gs: int = 128, threads: int | None = None, multimodal: int = 0) -> np.ndarray:
    """Returns the whole LMRS image as a uint8 array (pass .ctypes.data / len to lmrs_create).
    multimodal = 1 sets the header flag (export.py:84); the vision and processor sections (tools/synth_vision.py) follow the image."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    if q_type != Q_NONE:
        while cfg.dim % gs:                       # export.py:74-76 BACKOFF (header value only)
            gs //= 2
        assert gs == 128 or cfg.dim % 128 != 0, "the reference always quantises with 128 (utils/io.py:21)"
    total = image_size(cfg, q_type, gs)
    img = np.zeros(total, np.uint8)
    img[:256] = np.frombuffer(header_bytes(cfg, q_type, gs, multimodal), np.uint8)

    tasks = []   # (fam_idx, layer, row0, rows, q_off, s_off)
    off = 256
    for fi, (name, kind, per_layer, R, C, _) in enumerate(families(cfg)):
        nl = cfg.n_layers if per_layer else 1
        for layer in range(nl):
            cnt = R * C
            if kind == "n" or q_type == Q_NONE:
                qb, sb = cnt * 4, 0
            else:
                qb, sb = (cnt // 2 if q_type == Q4_0 else cnt), cnt // gs * 4
            for row0 in range(0, R, ROWS_PER_CHUNK):
                rows = min(ROWS_PER_CHUNK, R - row0)
                tasks.append((fi, layer, row0, rows, off, off + qb, kind))
            off += qb + sb
    assert off == total

    def run(t):
        fi, layer, row0, rows, q_base, s_base, kind = t
        C = families(cfg)[fi][4]
        w = float_tensor(cfg, seed, fi, layer, row0, rows)
        if kind == "n" or q_type == Q_NONE:
            b = w.reshape(-1).view(np.uint8)
            o = q_base + row0 * C * 4
            img[o:o + b.size] = b
            return
        if q_type == Q8_0:
            q, s = quantize_q80(w, gs)
            o = q_base + row0 * C
            img[o:o + q.size] = q.view(np.uint8)
        else:
            q, s = quantize_q40(w, gs)
            o = q_base + row0 * C // 2
            img[o:o + q.size] = q
        so = s_base + row0 * C // gs * 4
        img[so:so + s.size * 4] = s.view(np.uint8)

    threads = threads or min(32, os.cpu_count() or 1)
    if threads > 1 and len(tasks) > 1:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(run, tasks))
    else:
        for t in tasks:
            run(t)
    return img


def random_cfg(rng, i: int, max_pos: int = 32) -> ModelCfg:
    """A small random geometry of one of the three families: heads / kv heads / head size / hidden / vocabulary / depth all drawn
    (within what a real LMRS file can hold: every quantised matmul input a multiple of the group size 128)."""
    mt = int(rng.integers(0, 3))
    # Phi: the LongRoPE table has 48 entries (transformer.rs:473): head sizes up to 96
    hs = int(rng.choice([[64, 128, 256], [64, 96, 128], [64, 96]][mt]))
    while True:                                                            # wo's input (n_heads * head_size) is quantised in groups of 128
        n_kv = int(rng.choice([1, 2]))
        n_heads = n_kv * int(rng.choice([1, 2, 3, 4]))
        if n_heads * hs % 128 == 0:
            break
    dim = int(rng.choice([128, 256]))
    hidden = int(rng.choice([128, 256, 384]))
    # >= dim: Gemma's soft-cap loop indexes logits[0..dim) (transformer.rs:375); also vocabularies that are not a multiple of 4 (tail rows, Q6)
    vocab = dim + int(rng.integers(1, 40)) * 4 + int(rng.choice([0, 1, 3]))
    return ModelCfg(f"rand{i}", dim, hidden, int(rng.integers(1, 4)), n_heads, hs, n_kv, vocab, max_pos, [1e-6, 1e-5][mt != 0],
                    [10000.0, 500000.0, 10000.0][mt], mt)


def prompt_tokens(cfg: ModelCfg | str, n: int = 16, seed: int = 1234) -> np.ndarray:
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    g = np.random.Generator(np.random.PCG64(seed + 1))
    return g.integers(0, cfg.vocab_size, size=n, dtype=np.uint32)


def hf_state_dict(cfg: ModelCfg, seed: int):
    """The same float tensors under Hugging Face names, for feeding the reference's export.py."""
    sd = {}
    fam = families(cfg)
    idx = {f[0]: i for i, f in enumerate(fam)}

    def T(name, layer=0):
        return float_tensor(cfg, seed, idx[name], layer)

    sd["model.embed_tokens.weight"] = T("embed_tokens")
    sd["model.norm.weight"] = T("norm").reshape(-1)
    if cfg.model_type == PHI:
        sd["lm_head.weight"] = T("lm_head")
    for l in range(cfg.n_layers):
        p = f"model.layers.{l}."
        sd[p + "input_layernorm.weight"] = T("input_layernorm", l).reshape(-1)
        sd[p + "post_attention_layernorm.weight"] = T("post_attention_layernorm", l).reshape(-1)
        if cfg.model_type == GEMMA:
            sd[p + "pre_feedforward_layernorm.weight"] = T("pre_feedforward_layernorm", l).reshape(-1)
            sd[p + "post_feedforward_layernorm.weight"] = T("post_feedforward_layernorm", l).reshape(-1)
        if cfg.model_type == PHI:
            sd[p + "self_attn.qkv_proj.weight"] = np.concatenate([T("q_proj", l), T("k_proj", l), T("v_proj", l)], 0)
            sd[p + "mlp.gate_up_proj.weight"] = np.concatenate([T("gate_proj", l), T("up_proj", l)], 0)
        else:
            sd[p + "self_attn.q_proj.weight"] = T("q_proj", l)
            sd[p + "self_attn.k_proj.weight"] = T("k_proj", l)
            sd[p + "self_attn.v_proj.weight"] = T("v_proj", l)
            sd[p + "mlp.gate_proj.weight"] = T("gate_proj", l)
            sd[p + "mlp.up_proj.weight"] = T("up_proj", l)
        sd[p + "self_attn.o_proj.weight"] = T("o_proj", l)
        sd[p + "mlp.down_proj.weight"] = T("down_proj", l)
    return sd


def hf_config(cfg: ModelCfg) -> dict:
    return {"hidden_size": cfg.dim, "intermediate_size": cfg.hidden_dim, "num_hidden_layers": cfg.n_layers,
            "num_attention_heads": cfg.n_heads, "head_dim": cfg.head_size, "num_key_value_heads": cfg.n_kv_heads,
            "vocab_size": cfg.vocab_size, "max_position_embeddings": cfg.max_pos, "rms_norm_eps": cfg.rms_norm_eps,
            "rope_theta": cfg.rope_theta}


if __name__ == "__main__":
    import argparse
    import time
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--model", default="llama-3.2-1b", choices=sorted(CONFIGS))
    ap.add_argument("--qtype", default="q8_0", choices=["none", "q8_0", "q4_0"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    t0 = time.time()
    img = build_image(a.model, {"none": 0, "q8_0": 1, "q4_0": 2}[a.qtype], a.seed)
    img.tofile(a.out)
    print(f"wrote {a.out}: {img.size} bytes in {time.time() - t0:.1f}s")
