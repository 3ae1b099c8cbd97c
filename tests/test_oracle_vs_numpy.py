"""The C oracle against an independent numpy transcription of the same Rust sources (tests/numpy_ref.py),
on the golden fixtures the reference's export.py produced.  Bit-exact logits at every step."""
import os

import numpy as np
import pytest

import numpy_ref as NR
import oracle_lib as O
from tools import synth_lmrs as S

CASES = [("tiny_llama_q8", "tiny-llama", 7), ("tiny_llama_q4", "tiny-llama", 7), ("tiny_gemma_q8", "tiny-gemma", 8),
         ("tiny_gemma_q4", "tiny-gemma", 8), ("tiny_phi_q8", "tiny-phi", 9), ("tiny_llama_f32", "tiny-llama", 7)]


@pytest.mark.parametrize("name,cfg,seed", CASES)
def test_oracle_matches_numpy_transcription(golden_dir, name, cfg, seed):
    img = np.fromfile(os.path.join(golden_dir, name + ".lmrs"), np.uint8)
    orc = O.Oracle(img)
    ref = NR.NumpyModel(img)
    assert ref.end == orc.bytes_consumed == img.size
    prompt = S.prompt_tokens(cfg, 3, seed)
    tok = None
    for pos in range(5):
        t = int(prompt[pos]) if pos < len(prompt) else tok
        lo = orc.forward(t, pos).copy()
        ln = ref.forward(t, pos)
        assert (lo.view(np.uint32) == ln.view(np.uint32)).all(), f"{name} pos {pos}: {np.flatnonzero(lo != ln)[:5]}"
        tok = int(np.argmax(lo))


def test_ops_match_numpy():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(512) * 3).astype(np.float32)
    q, s = O.quantize(x); qn, sn = NR.quantize_q8(x)
    assert (q == qn).all() and (s.view(np.uint32) == sn.view(np.uint32)).all()
    q4, s4 = O.quantize_q4(x); q4n, s4n = NR.quantize_q4(x)
    assert (q4 == q4n).all() and (s4.view(np.uint32) == s4n.view(np.uint32)).all()
    w = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    for unit in (False, True):
        a = O.rmsnorm(x, w, 1e-5, unit); b = NR.rmsnorm(x, w, 1e-5, unit)
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("q", [S.Q8_0, S.Q4_0, S.Q_NONE])
def test_image_projector_oracle_matches_numpy_transcription(q):
    """processor.rs:234-342: the C restatement against a reshape/transpose statement of the HD transform and the numpy ops,
    on rows that cover the sub-image body, a row separator, glb_GN, the global crop and its last separator; Q8_0, Q4_0 and
    unquantised sections."""
    from tools import synth_vision as V
    sec = V.build_processor_section(seed=11, q_type=q)
    orc = O.ProcessorOracle(sec)
    rng = np.random.default_rng(4)
    w_crop, h_crop = 2, 1
    feats = (rng.standard_normal((1 + w_crop * h_crop, 576, 1024)) * 1.5).astype(np.float32)
    ref = orc.forward(feats, 576 * 1024, 12, w_crop, h_crop)
    n_sub = 12 * h_crop * (12 * w_crop + 1)
    rows = [0, 23, 24, 25, 12 * w_crop * 7 + 7 + 13, n_sub - 1, n_sub, n_sub + 1, n_sub + 13, n_sub + 14, ref.shape[0] - 2, ref.shape[0] - 1]
    ne, got = NR.processor_forward(sec.tobytes(), feats, w_crop, h_crop, rows)
    assert ne == ref.shape[0]
    for r in rows:
        assert (got[r].view(np.uint32) == ref[r].view(np.uint32)).all(), f"embedding {r}: {np.flatnonzero(got[r] != ref[r])[:5]}"


@pytest.mark.parametrize("q,small", [(S.Q8_0, False), (S.Q4_0, False), (S.Q_NONE, True), (S.Q4_0, True)])
def test_vision_tower_oracle_matches_numpy_transcription(q, small):
    """vision.rs:244-577 at the real ViT-L/14-336 geometry (small: 128 wide, 2 heads, two encoder layers run), one crop: the C
    restatement against a vectorised numpy statement written from the Rust source (patch conv with the matmul_rest tail, f32x8
    lane sums, sequential softmax sum, shared row quantisation for q/k/v, QuickGELU through libm expf); Q8_0, Q4_0 and
    unquantised sections."""
    from tools import synth_vision as V
    cfg = V.VisionCfg(dim=128, hidden_dim=512, n_layers=3, n_heads=2, head_size=64) if small else V.VisionCfg(n_layers=2)
    sec = V.build_vision_section(cfg, seed=21, q_type=q)
    orc = O.VisionOracle(sec)
    pv = V.pixel_values(cfg, 1, seed=8)
    ref = orc.forward(pv, 1)[0]
    end, got = NR.vision_forward(sec.tobytes(), pv[0])
    assert end == orc.bytes_consumed == sec.size
    assert got.shape == ref.shape
    bad = np.flatnonzero(got.view(np.uint32).reshape(-1) != ref.view(np.uint32).reshape(-1))
    assert bad.size == 0, f"{bad.size} of {ref.size} differ, first {bad[:5]}"


@pytest.mark.parametrize("q", [S.Q8_0, S.Q4_0])
def test_oracle_matches_numpy_when_attention_is_wider_than_the_model(q):
    """n_heads * head_size > dim (Gemma-2-9B's geometry; the reference grows its activation buffer, transformer.rs:497-499)."""
    img = S.build_image("tiny-wide-att", q, seed=3)
    orc = O.Oracle(img); ref = NR.NumpyModel(img)
    prompt = S.prompt_tokens("tiny-wide-att", 3, 3)
    tok = None
    for pos in range(6):
        t = int(prompt[pos]) if pos < 3 else tok
        lo = orc.forward(t, pos).copy(); ln = ref.forward(t, pos)
        assert (lo.view(np.uint32) == ln.view(np.uint32)).all(), f"pos {pos}"
        tok = int(np.argmax(lo))


@pytest.mark.parametrize("name,cfg,seed,n_tok,pos0", [("tiny_llama_q8", "tiny-llama", 7, 6, 0), ("tiny_phi_q8", "tiny-phi", 9, 5, 2),
                                                      ("tiny_gemma_q8", "tiny-gemma", 8, 5, 3), ("tiny_llama_q4", "tiny-llama", 7, 4, 1)])
def test_fill_kv_cache_oracle_matches_numpy_transcription(golden_dir, name, cfg, seed, n_tok, pos0):
    """Transformer::fill_kv_cache (transformer.rs:672-684, forward_layer with sl > 1): the mutated embeddings and the logits of
    the decode step that follows, C restatement against the numpy transcription.  pos0 > 0: positions before the batch are
    filled token by token first (Gemma: the batched call then masks with its first position, :525).  Q4_0: the decode-form
    product per token on both sides."""
    img = np.fromfile(os.path.join(golden_dir, name + ".lmrs"), np.uint8)
    orc = O.Oracle(img); ref = NR.NumpyModel(img)
    warm = S.prompt_tokens(cfg, pos0, seed + 1)
    for pos in range(pos0):
        orc.forward(int(warm[pos]), pos); ref.forward(int(warm[pos]), pos)
    toks = S.prompt_tokens(cfg, n_tok, seed)
    a = orc.get_embeddings(toks).reshape(n_tok, -1).copy()
    b = np.stack([ref.embed(int(t)) for t in toks])
    assert (a.view(np.uint32) == b.view(np.uint32)).all()            # get_embeddings: dequantised rows, no Gemma scaling (transformer.rs:659-669)
    flat = a.reshape(-1)
    assert orc.fill_kv_cache(flat, pos0) == ref.fill_kv_cache(b, pos0) == pos0 + n_tok
    assert (flat.view(np.uint32) == b.reshape(-1).view(np.uint32)).all(), np.flatnonzero(flat != b.reshape(-1))[:5]
    lo = orc.forward(3, pos0 + n_tok).copy(); ln = ref.forward(3, pos0 + n_tok)
    assert (lo.view(np.uint32) == ln.view(np.uint32)).all()


@pytest.mark.parametrize("i", range(10))
def test_oracle_matches_numpy_on_random_geometries(i):
    """The two transcriptions of the forward pass on geometries neither was written against: random family / heads / kv heads /
    head size / hidden / vocabulary / depth, all three weight formats (Q8_0, Q4_0, f32)."""
    rng = np.random.default_rng(1000 + i)
    cfg = S.random_cfg(rng, i)
    q = [S.Q8_0, S.Q4_0, S.Q_NONE][i % 3]
    img = S.build_image(cfg, q, seed=50 + i, threads=1)
    orc = O.Oracle(img); ref = NR.NumpyModel(img)
    assert ref.end == orc.bytes_consumed == img.size
    tok = int(rng.integers(0, cfg.vocab_size))
    for pos in range(4):
        lo = orc.forward(tok, pos).copy(); ln = ref.forward(tok, pos)
        assert (lo.view(np.uint32) == ln.view(np.uint32)).all(), f"{cfg} q{q} pos {pos}: {np.flatnonzero(lo != ln)[:5]}"
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
