"""LMRS v4 layout: the golden files were written by the REFERENCE's own export.py
(tests/golden/make_golden.py); the synthetic writer must reproduce them byte for byte and the
oracle's parser must consume exactly the file."""
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
from tools import synth_lmrs as S

FIX = [("tiny_llama_q8", "tiny-llama", S.Q8_0, 7), ("tiny_llama_q4", "tiny-llama", S.Q4_0, 7), ("tiny_gemma_q8", "tiny-gemma", S.Q8_0, 8),
       ("tiny_gemma_q4", "tiny-gemma", S.Q4_0, 8), ("tiny_phi_q8", "tiny-phi", S.Q8_0, 9), ("tiny_llama_f32", "tiny-llama", S.Q_NONE, 7)]


@pytest.mark.parametrize("name,cfg,q,seed", FIX)
def test_synth_writer_reproduces_reference_export(golden_dir, name, cfg, q, seed):
    ref = np.fromfile(os.path.join(golden_dir, name + ".lmrs"), np.uint8)
    mine = S.build_image(cfg, q, seed)
    assert mine.size == ref.size == S.image_size(S.CONFIGS[cfg], q)
    assert (mine == ref).all()
    # thread count / chunking must not change the bytes
    assert (S.build_image(cfg, q, seed, threads=1) == ref).all()


@pytest.mark.parametrize("name,cfg,q,seed", FIX)
def test_header_and_extent(golden_dir, name, cfg, q, seed):
    img = np.fromfile(os.path.join(golden_dir, name + ".lmrs"), np.uint8)
    c = S.CONFIGS[cfg]
    assert bytes(img[:4]) == b"lmrs" and struct.unpack("I", img[4:8])[0] == 4
    o = O.Oracle(img)
    a = o.args
    assert (a.dim, a.hidden_dim, a.n_layers, a.n_heads, a.head_size, a.n_kv_heads, a.vocab_size) == \
        (c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.head_size, c.n_kv_heads, c.vocab_size)
    assert a.q_type == q and a.model_type == c.model_type and a.group_size == 128 and a.seq_len == min(c.max_pos, 8192)
    assert o.bytes_consumed == img.size          # Transformer::new's second return value


def test_real_model_sizes_match_the_published_files():
    # README.md:31-42 file sizes (GB, 3 significant digits) pin the layout arithmetic at full scale
    gb = lambda cfg, q: S.image_size(S.CONFIGS[cfg], q) / 1e9
    assert abs(gb("llama-3.2-1b", S.Q8_0) - 1.27) < 0.01
    assert abs(gb("llama-3.2-3b", S.Q8_0) - 3.31) < 0.01
    assert abs(gb("llama-3.2-3b", S.Q4_0) - 1.71) < 0.01
    assert abs(gb("gemma-2-2b", S.Q4_0) - 1.39) < 0.01
    # (README lists Gemma-2-2B Q8_0 as 2.66GB = what group size 256 would give; the other five match gs=128.)
    assert abs(gb("phi-3.5", S.Q8_0) - 3.94) < 0.01


def test_seq_len_is_clamped_like_the_reference():
    img = S.build_image("tiny-llama", S.Q8_0, 1)
    img[36:40] = np.frombuffer(struct.pack("I", 131072), np.uint8)     # max_position_embeddings
    assert O.Oracle(img).args.seq_len == 8192                          # transformer.rs:158-160


def test_malformed_images_are_rejected():
    img = S.build_image("tiny-llama", S.Q8_0, 1)
    bad = img.copy(); bad[0] = 0x00
    with pytest.raises(RuntimeError, match="lm.rs format"):
        O.Oracle(bad)
    with pytest.raises(RuntimeError, match="truncated"):
        O.Oracle(img[:-4])
    with pytest.raises(RuntimeError):
        O.Oracle(img[:100])


@pytest.mark.parametrize("q", [S.Q8_0, S.Q4_0, S.Q_NONE])
def test_multimodal_sections_match_the_reference_exporter(golden_dir, q):
    """The vision and processor sections tools/synth_vision.py writes (the inputs of every image-path test) are, byte for byte,
    what the reference's export.py --vision-config wrote for the same tensors (tests/golden/make_golden.py recorded its size and
    SHA-256), and the CPU restatements of Transformer::new / VisionTransformer::new / PHI3VProcessor::new walk the file the way
    chat.rs:84-91 does: each consumes exactly its section."""
    import hashlib
    import json
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden as G
    fx = json.load(open(os.path.join(golden_dir, G.MM_VARIANTS[q] + ".json")))
    text, vis, proc = G.mm_expected_image(q)
    img = np.concatenate([text, vis, proc])
    assert (text.size, vis.size, proc.size, img.size) == (fx["text_bytes"], fx["vision_bytes"], fx["processor_bytes"], fx["bytes"])
    assert hashlib.sha256(img.tobytes()).hexdigest() == fx["sha256"]
    t = O.Oracle(img)
    assert t.bytes_consumed == fx["text_bytes"] and t.args.multimodal
    v = O.VisionOracle(img[t.bytes_consumed:])
    assert v.bytes_consumed == fx["vision_bytes"]
    p = O.ProcessorOracle(img[t.bytes_consumed + v.bytes_consumed:])
    assert p.bytes_consumed == fx["processor_bytes"]
