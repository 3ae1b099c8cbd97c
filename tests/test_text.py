"""Tokenizer and sampler: the host code of lm.rs_amd/csrc/lmrs_text.cpp through the C ABI against a second,
pure-Python transcription of the Rust sources (tests/text_ref.py).  No GPU needed.  The reference ships no tokenizer fixture (its
tokenizer.bin comes from Hugging Face checkpoints that are not reachable offline), so the vocabularies here are synthetic files
in the exact layout src/tokenizer.rs:24-64 reads and utils/tokenizers/*.py write (struct "IIII" header, then "fI" + bytes per token)."""
import struct

import numpy as np
import pytest

import oracle_lib as O
import text_ref as R


def make_tokenizer_bin(n_fill=0, dup=False):
    """SentencePiece-like layout: 0 <unk>, 1 <s>, 2 </s>, 3..258 the byte tokens <0x00>..<0xFF>, then characters, merges, fillers."""
    toks = [("<unk>", 0.0), ("<s>", 0.0), ("</s>", 0.0)] + [("<0x%02X>" % b, 0.0) for b in range(256)]
    for i, ch in enumerate(" abcdefghijklmnopqrstuvwxyzABCDEFGH.,!?0123456789'\n"):
        toks.append((ch, -1.0 - i))
    toks += [("é", -80.0), ("ß", -81.0), ("世", -82.0), ("界", -83.0), ("😀", -84.0)]
    merges = ["th", "he", "the", " t", " the", "in", "ing", "er", "an", "and", " a", " an", " and", "ll", "hello", "he" + "ll", "lo", "wor", "or", "ld",
              "world", " w", " wor", " world", "世界", "tt", "ttt", "oo", "ooo", "é" + "é", "00", "000", "!!"]
    for i, m in enumerate(merges):
        toks.append((m, 10.0 - 0.37 * i if i % 3 else 10.0 - 0.37 * (i - 1)))          # some equal scores: the first pair in text order wins
    if dup:
        toks += [("the", 3.0), ("ing", 99.0), ("a", -5.0), ("a", -6.0)]                  # duplicate strings: binary-search flavour matters
    toks += [("<fill_%d>" % i, 0.0) for i in range(n_fill)]
    blob = struct.pack("IIII", len(toks), max(len(t.encode()) for t, _ in toks), 1, 2)
    for t, s in toks:
        b = t.encode("utf-8")
        blob += struct.pack("fI", s, len(b)) + b
    return blob, [t for t, _ in toks]


TEXTS = ["hello world", "the thing and the other thing", "a", "tttt", "ooooo", "hello, world!!", " and  and", "naïve café ééé", "世界 hello 世界世",
         "😀 ok 😀😀", "tab\there", "Zürich ß", "0000 000 00", "x" * 40, "the" * 9, "\n\nnew\nlines", "ÿþĀ", "hello\x00world"]


@pytest.fixture(scope="module")
def L():
    import lmrs_amd
    return lmrs_amd


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("dup", [False, True])
def test_encode_matches_the_second_transcription(L, monkeypatch, flavour, dup):
    blob, _ = make_tokenizer_bin(dup=dup)
    monkeypatch.setenv("LMRS_BSEARCH_FLAVOUR", str(flavour))
    dev = L.Tokenizer(blob); ref = R.Tokenizer(blob, flavour)
    assert (dev.vocab_size, dev.bos, dev.eos) == (ref.vocab_size, 1, 2)
    rng = np.random.default_rng(5)
    alphabet = list(" abcdefghijklmnopqrstuvwxyz.,!?0'\nthe ing and éß世界😀Ω")
    texts = TEXTS + ["".join(rng.choice(alphabet, size=int(rng.integers(1, 60)))) for _ in range(120)]
    for text in texts:
        for bos, eos in ((False, False), (True, True)):
            got = dev.encode(text, bos, eos, False, 1).tolist()
            assert got == ref.encode(text, bos, eos, False, 1), (text, bos, eos)
    if not dup:                                                    # without duplicate strings the std version cannot matter
        other = R.Tokenizer(blob, 1 - flavour)
        for text in texts[:40]:
            assert other.encode(text, False, False, False, 1) == ref.encode(text, False, False, False, 1)


@pytest.mark.parametrize("model_type,name", [(0, "GEMMA"), (1, "LLAMA"), (2, "PHI")])
def test_chat_format_templates(L, model_type, name):
    """The hard-coded template ids of tokenizer.rs:88-96 / 137-145 go through the merge loop like every other id (the prefix is part
    of `tokens` when the pairs are scanned; the suffix is appended afterwards): the vocabulary must hold them (128 007 for Llama)."""
    blob, _ = make_tokenizer_bin(n_fill=128100)
    dev = L.Tokenizer(blob); ref = R.Tokenizer(blob)
    for text in ("hello world", "the", "世界!"):
        got = dev.encode(text, False, False, True, model_type).tolist()
        want = ref.encode(text, False, False, True, model_type)
        assert got == want, (name, text)
    pre = {0: [1, 106, 1645, 108], 1: [128006, 882, 128007, 271], 2: [1, 32010, 29871, 13]}[model_type]
    post = {0: [107, 108, 106, 2516, 108], 1: [128009, 128006, 78191, 128007, 271], 2: [32007, 29871, 13, 32001, 29871, 13]}[model_type]
    assert got[:4] == pre and got[-len(post):] == post


def test_decode_every_token(L):
    blob, toks = make_tokenizer_bin()
    dev = L.Tokenizer(blob); ref = R.Tokenizer(blob)
    for i in range(len(toks)):
        assert dev.decode(i) == ref.decode(i), i
    assert dev.decode(3 + 0x41) == "A" and dev.decode(3 + 0xE9) == "é" and dev.decode(0) == "<unk>"      # <0xE9> -> char::from(0xE9) = U+00E9


def test_tokenizer_errors(L):
    blob, _ = make_tokenizer_bin()
    dev = L.Tokenizer(blob)
    with pytest.raises(L.LmrsError, match="should not be empty"):
        dev.encode("", False, False, False, 1)
    with pytest.raises(L.LmrsError, match="out of the vocabulary"):
        dev.encode("hi", False, False, True, 1)                   # Llama template ids in a 300-token vocabulary: vocab[128006] panics
    with pytest.raises(L.LmrsError, match="truncated"):
        L.Tokenizer(blob[: len(blob) // 2])
    bad = bytearray(blob); bad[16 + 8] = 0xFF                      # first token string no longer UTF-8
    with pytest.raises(L.LmrsError, match="token string"):
        L.Tokenizer(bytes(bad))
    with pytest.raises(L.LmrsError):
        dev.decode(10 ** 6)


def test_random_numbers():
    """functional.rs:34-44, against values worked out by hand from the definition."""
    assert R.random_u32(1) == ((((1 ^ (1 << 25)) ^ ((1 ^ (1 << 25)) >> 27)) * 0x2545F4914F6CDD1D) & ((1 << 64) - 1)) >> 32
    assert 0.0 <= float(R.random_f32(1234567)) < 1.0


@pytest.mark.parametrize("temperature,top_p,seed", [(0.0, 0.9, 1), (0.7, 0.9, 42), (0.7, 0.9, 1727000000123), (1.3, 0.5, 7), (0.7, 1.0, 42), (0.7, 0.0, 99), (0.2, 0.95, 3)])
def test_sampler_matches_the_second_transcription(L, temperature, top_p, seed):
    """Sampler::sample over several calls of ONE sampler: the candidate vector keeps stale entries between calls and is sorted as a
    whole (sampler.rs:81), and the random number is the same on every call (:119) - both are part of the reference's behaviour.
    Three implementations: the product's host sampler (lmrs_text.cpp), the ORACLE's (oracle/lmrs_oracle.c lmrs_ref_sampler_*: the
    checker of the GPU tier's sampler tests) and the pure-Python transcription."""
    V = 1500
    dev = L.Sampler(V, temperature, top_p, seed); orc = O.Sampler(V, temperature, top_p, seed); ref = R.Sampler(V, temperature, top_p, seed)
    rng = np.random.default_rng(seed % 1000)
    for call in range(5):
        lg = (rng.standard_normal(V) * (3.0 if call % 2 else 0.8)).astype(np.float32)
        if call == 3:
            lg[17] = lg.max() + 9.0                                # a dominant token: top-p keeps a single candidate
        a = lg.copy(); o = lg.copy(); b = [np.float32(v) for v in lg]
        t_dev = dev.sample(a); t_orc = orc.sample(o); t_ref = ref.sample(b)
        assert t_dev == t_ref == t_orc, (call, t_dev, t_orc, t_ref)
        assert (a.view(np.uint32) == np.array(b, np.float32).view(np.uint32)).all(), f"call {call}: logits after sample() differ"
        assert (o.view(np.uint32) == np.array(b, np.float32).view(np.uint32)).all(), f"call {call}: the oracle's logits after sample() differ"


def test_oracle_sampler_on_the_real_vocabulary_size():
    """The oracle's sampler against the product's host sampler at 128 256 entries over many calls of one sampler each: wide calls
    followed by narrow ones (stale candidates above the new ones' tail), ties (equal probabilities keep index order: the sort is stable),
    the same random number every call."""
    import lmrs_amd as L
    V = 128256
    rng = np.random.default_rng(3)
    for temperature, top_p in [(0.7, 0.9), (1.0, 0.3), (0.9, 1.0)]:
        a = L.Sampler(V, temperature, top_p, 20240925); b = O.Sampler(V, temperature, top_p, 20240925)
        for call in range(6):
            lg = (rng.standard_normal(V) * [0.3, 6.0, 25.0][call % 3]).astype(np.float32)
            if call == 4:
                lg[1000:1040] = lg.max() + 1.0                     # forty equal maxima
            x = lg.copy(); y = lg.copy()
            assert a.sample(x) == b.sample(y), (temperature, top_p, call)
            assert (x.view(np.uint32) == y.view(np.uint32)).all()


def test_oracle_random_numbers():
    for seed in (0, 1, 42, 1234567, 1727000000123, (1 << 64) - 1):
        assert O.random_u32(seed) == R.random_u32(seed)
        assert np.float32(O.random_f32(seed)) == R.random_f32(seed)


def test_topp_from_candidate_pairs_equals_the_whole_sampler(L):
    """lmrs_sampler_topp_pairs (what lmrs_forward_sample hands the host after the device has scaled, soft-maxed and filtered) against
    lmrs_sampler_sample on the same logits, one persistent sampler each over many calls: narrow calls after wide ones leave stale
    candidates in the vector the reference sorts as a whole (sampler.rs:81)."""
    rng = np.random.default_rng(11)
    V = 3000
    a = L.Sampler(V, 0.7, 0.9, 4242); b = L.Sampler(V, 0.7, 0.9, 4242)
    cutoff = np.float32((np.float32(1.0) - np.float32(0.9)) / np.float32(V - 1))
    for call in range(40):
        spread = [0.5, 8.0, 30.0][call % 3]                    # flat, peaked, very peaked
        logits = (rng.standard_normal(V) * spread).astype(np.float32)
        probs = logits.copy()
        want = a.sample(probs)                                 # probs now holds the probabilities (scaled and soft-maxed in place, sampler.rs:115-117)
        keep = np.flatnonzero(probs >= cutoff).astype(np.uint32)
        assert b.topp_pairs(probs[keep], keep) == want, call


def test_sampler_from_the_exponentials_equals_the_whole_sampler(L):
    """lmrs_sampler_sample_exps (what lmrs_forward_sample hands the host: exp(logits / temperature - max), formed on the device) against the
    ORACLE's sampler on the logits - token and probabilities, sample_mult and top-p, one persistent sampler each over many calls."""
    rng = np.random.default_rng(12)
    V = 5000
    for temperature, top_p in [(0.7, 0.9), (1.2, 1.0), (0.4, 0.0), (2.0, 0.5)]:
        a = L.Sampler(V, temperature, top_p, 99); b = O.Sampler(V, temperature, top_p, 99)
        for call in range(12):
            logits = (rng.standard_normal(V) * [0.5, 8.0, 30.0][call % 3]).astype(np.float32)
            t = (logits / np.float32(temperature)).astype(np.float32)
            exps = np.array([O.expf(float(v)) for v in (t - t.max()).astype(np.float32)], np.float32)
            want_p = logits.copy(); want = b.sample(want_p)
            assert a.sample_exps(exps) == want, (temperature, top_p, call)
            assert (exps.view(np.uint32) == want_p.view(np.uint32)).all(), "probabilities"


def test_sampler_argmax_rule(L):
    s = L.Sampler(6, 0.0, 0.9, 1)
    assert s.sample(np.array([1, 5, 5, 2, 5, 0], np.float32)) == 1                      # first index of the maximum
    assert s.sample(np.array([np.nan, 5, 9, 2, 5, 0], np.float32)) == 0                 # NaN at index 0 is never displaced
    assert s.sample(np.array([1, np.nan, 9, 2, 5, 0], np.float32)) == 2
    assert s.sample(np.full(6, -np.inf, np.float32)) == 0
