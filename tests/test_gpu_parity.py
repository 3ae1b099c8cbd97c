"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, through the C ABI, against the
CPU oracle on the same seeded inputs.  Bar: bit-exact (integer work and, by construction of the
kernels, every float too — see DESIGN.md "Parity")."""
import os

import numpy as np
import pytest

import oracle_lib as O
from tools import synth_lmrs as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import lmrs_amd
    return lmrs_amd


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, what
    ne = np.flatnonzero(bits(a) != bits(b)) if a.dtype == np.float32 else np.flatnonzero(a != b)
    assert ne.size == 0, f"{what}: {ne.size}/{a.size} elements differ, first at {ne[:5]}: {a.ravel()[ne[:5]]} vs {b.ravel()[ne[:5]]}"


# ------------------------------------------------------------------ L2 free functions
def test_expf_matches_host_libm(L):
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        rng.uniform(-30, 0, 1 << 20), rng.uniform(-104, 89, 1 << 20), rng.uniform(-1e-3, 1e-3, 1 << 16),
        np.array([0.0, -0.0, 88.0, 88.72, 88.73, 89.0, -87.3, -88.0, -103.0, -103.9, -104.0, -200.0, np.inf, -np.inf, 1e-40, -1e-40]),
    ]).astype(np.float32)
    dev = L.expf(xs)
    host = np.array([O.expf(float(v)) for v in xs[: 1 << 16]], np.float32)
    assert_bit_equal(dev[: 1 << 16], host, "expf (first 65536)")
    # the rest through the oracle's softmax-free path: vectorised host expf via the oracle's op
    tail = xs[-16:]
    assert_bit_equal(dev[-16:], np.array([O.expf(float(v)) for v in tail], np.float32), "expf special values")
    mid = xs[1 << 20: (1 << 20) + (1 << 16)]
    assert_bit_equal(dev[1 << 20: (1 << 20) + (1 << 16)], np.array([O.expf(float(v)) for v in mid], np.float32), "expf wide range")


@pytest.mark.parametrize("n,temperature", [(64, 1.0), (1000, 0.7), (4102, 1.3), (128256, 0.8), (32064, 2.0)])
def test_sampler_on_the_device(L, n, temperature):
    """Sampler::sample (sampler.rs:109-129; temperature != 0, sample_mult) on the device against the ORACLE's sampler
    (oracle/lmrs_oracle.c lmrs_ref_sampler_*; the product's host sampler, lmrs_text.cpp, is checked against it as well): the
    probabilities the logits are turned into bit for bit - the sequential softmax sum over the whole vocabulary - and the same draw
    for random numbers across the range, the first / last term and block-of-64 boundaries of the running cdf included."""
    rng = np.random.default_rng(n)
    logits = (rng.standard_normal(n) * 3).astype(np.float32)
    orc = O.Sampler(n, temperature, 1.0, 1)                       # top_p = 1: sample_mult
    host = L.Sampler(n, temperature, 1.0, 1)
    rnd0 = O.random_f32(1)
    assert np.float32(host.info()[3]) == np.float32(rnd0)
    want_probs = logits.copy(); want_tok = orc.sample(want_probs)
    host_probs = logits.copy(); assert host.sample(host_probs) == want_tok
    assert_bit_equal(host_probs, want_probs, "host sampler's softmax against the oracle's")
    tok, probs = L.sample_mult(logits, temperature, rnd0)
    assert_bit_equal(probs, want_probs, f"softmax of {n} logits at temperature {temperature}")
    assert tok == want_tok
    # the draw for other random numbers: sample_mult restated on the (bit-equal) probabilities - numpy's f32 cumsum is the same
    # sequential chain of f32 adds
    cdf = np.cumsum(want_probs, dtype=np.float32)
    def draw(r):
        hit = np.flatnonzero(np.float32(r) < cdf)
        return int(hit[0]) if hit.size else n - 1
    for r in [0.0, 1e-9, 0.25, 0.5, 0.9, 0.999, 0.9999999, 1.0, float(cdf[min(n - 1, 63)]), float(cdf[min(n - 1, 64)]), float(cdf[min(n - 1, 511)]), float(cdf[n // 2])]:
        t2, _ = L.sample_mult(logits, temperature, r)
        assert t2 == draw(r), f"n={n} r={r}: device {t2}, host {draw(r)}"


def test_forward_sample_matches_the_oracle(L):
    """lmrs_forward_sample (logits stay in HBM) and lmrs_forward + lmrs_sampler_sample (the host route) against the ORACLE's
    forward + the oracle's sampler (oracle/lmrs_oracle.c): greedy, temperature sampling, top-p (scaling, softmax and the cutoff
    filter on the device, the candidates' sort on the host).  Three independent runs, one persistent sampler each."""
    img = S.build_image("mini-llama", S.Q8_0, seed=77)
    prompt = S.prompt_tokens("mini-llama", 4, 77)
    for temperature, top_p, steps in [(0.0, 0.9, 12), (0.8, 1.0, 12), (1.5, 0.0, 12), (0.7, 0.9, 200), (0.05, 0.5, 60), (3.0, 0.999, 40)]:
        a = L.Transformer(img); b = L.Transformer(img); o = O.Oracle(img)
        V = a.args.vocab_size
        sa = L.Sampler(V, temperature, top_p, 12345); sb = L.Sampler(V, temperature, top_p, 12345); so = O.Sampler(V, temperature, top_p, 12345)
        ta = tb = to = None
        for pos in range(steps):
            t = int(prompt[pos]) if pos < len(prompt) else to
            ta = a.forward_sample(t, pos, sa)
            tb = sb.sample(b.forward(t, pos))
            to = so.sample(o.forward(t, pos).copy())
            assert ta == to, f"temperature {temperature}, top_p {top_p}, pos {pos}: device route {ta}, oracle {to}"
            assert tb == to, f"temperature {temperature}, top_p {top_p}, pos {pos}: host route {tb}, oracle {to}"


def test_topp_filter_on_peaked_and_flat_distributions(L):
    """sample_topp through lmrs_forward_sample (exponentials on the device, the sequential sum, the cutoff filter and the draw on the host) on a
    peaked and a flat distribution, with stale candidates of an earlier, wider call still in the sampler's persistent vector - the device route
    and the host route each against the ORACLE's sampler on the oracle's logits."""
    rng = np.random.default_rng(5)
    img = S.build_image("mini-llama", S.Q8_0, seed=78)
    a = L.Transformer(img); b = L.Transformer(img); o = O.Oracle(img)
    V = a.args.vocab_size
    for temperature, top_p in [(0.7, 0.9), (1.0, 0.3)]:
        sa = L.Sampler(V, temperature, top_p, 777); sb = L.Sampler(V, temperature, top_p, 777); so = O.Sampler(V, temperature, top_p, 777)
        for pos in range(30):                                  # one sampler per route across calls: the persistent-vector quirk is exercised
            t = int(rng.integers(0, V))
            want = so.sample(o.forward(t, pos).copy())
            assert a.forward_sample(t, pos, sa) == want, (temperature, top_p, pos)
            assert sb.sample(b.forward(t, pos)) == want, (temperature, top_p, pos)


@pytest.mark.parametrize("cfg,sort_min", [("mini-llama", 16), ("mini-llama-v40k", 16), ("mini-llama-v40k", 9000)])
def test_topp_candidates_sorted_on_the_device(L, monkeypatch, cfg, sort_min):
    """sample_topp's sort (sampler.rs:81) on the device - a bitonic network over the unique keys (prob descending, index ascending), i.e. the
    permutation the reference's stable sort gives - forced on from 16 candidates: one sort block (vocabulary 4096) and several (40 000: the global
    merge steps), flat and peaked distributions in turn on ONE sampler so that the merge with the stale rest of its vector runs too, wider calls
    before narrower ones and back; sort_min 9000: calls above and below the switch interleave host-sorted and device-sorted states.  Against the
    oracle's sampler on the oracle's logits, token by token."""
    monkeypatch.setenv("LMRS_TOPP_DEVICE_SORT_MIN", str(sort_min))
    rng = np.random.default_rng(6)
    img = S.build_image(cfg, S.Q8_0, seed=79)
    a = L.Transformer(img); o = O.Oracle(img)
    V = a.args.vocab_size
    for top_p in (0.9, 0.3):
        samplers = {t: (L.Sampler(V, t, top_p, 4242), O.Sampler(V, t, top_p, 4242)) for t in (1.0, 0.05)}
        for pos in range(24):
            t = int(rng.integers(0, V))
            lo = o.forward(t, pos).copy()
            for temperature in ((1.0, 0.05) if pos % 3 else (0.05, 1.0)):
                sa, so = samplers[temperature]
                assert a.forward_sample(t, pos, sa) == so.sample(lo.copy()), (cfg, top_p, temperature, pos)
    # ONE sampler fed flat and peaked distributions alternately: device-sorted candidates merged into the vector a host-sorted call left, and back
    sa, so = L.Sampler(V, 0.7, 0.9, 5), O.Sampler(V, 0.7, 0.9, 5)
    for pos in range(12):
        t = int(rng.integers(0, V))
        lo = o.forward(t, pos).copy()
        if pos % 2:                                              # a dominant logit: a handful of candidates, sorted on the host (Sampler::sample on copied logits)
            la = a.forward(t, pos).copy()                        # (the step itself runs on both sides: the later positions attend to this one)
            k = int(rng.integers(0, V))
            la[k] += 40.0; lo[k] += 40.0
            assert sa.sample(la) == so.sample(lo), (cfg, pos)
        else:                                                    # the model's own flat distribution through the device route
            assert a.forward_sample(t, pos, sa) == so.sample(lo), (cfg, pos)


@pytest.mark.parametrize("c", [1.0, 0.7978845608028654])
def test_tanh_matches_host_libm(L, c):
    """f64::tanh (Gemma's score / logit soft-caps, c = 1; the tanh-GELU, c = 0.79788...) is the one transcendental the device does
    not restate: it calls ocml's f64 tanh and rounds to f32, the reference calls the host libm.  Swept here over every f32 in four
    binades around 1 where tanh bends (|x| in [0.25, 4): 2 x 33.5 M inputs), every 61st f32 of the whole line, and the edges."""
    band = np.arange(np.float32(0.25).view(np.uint32), np.float32(4.0).view(np.uint32), dtype=np.uint32).view(np.float32)
    line = np.arange(0, 0x7f800000, 61, dtype=np.uint32).view(np.float32)
    edges = np.array([0.0, -0.0, 1e-40, -1e-40, 1e-8, 19.0, 19.06, 19.1, 20.0, 30.0, 88.0, 1e30, np.inf, -np.inf, np.nan], np.float32)
    for xs in (band, -band, line, -line, edges):
        dev = L.tanh_cast(xs, c)
        host = O.tanh_cast(xs, c)
        ne = np.flatnonzero(bits(dev) != bits(host))
        ne = ne[~(np.isnan(dev[ne]) & np.isnan(host[ne]))]
        assert ne.size == 0, f"tanh(c={c}): {ne.size}/{xs.size} inputs differ, first x = {xs[ne[:5]]}: device {dev[ne[:5]]} host {host[ne[:5]]}"


@pytest.mark.parametrize("n", [128, 256, 2048, 3072, 8192, 9216])
def test_quantize(L, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    x[5] = 0.5 * np.abs(x[:128]).max()          # exercise round-half-away ties: x/scale = 63.5
    if n >= 256:
        x[128:256] = 0.0                         # all-zero group: scale 0, 0/0 = NaN -> 0
    q, s = L.quantize(x)
    qo, so = O.quantize(x)
    assert_bit_equal(s, so, "q8 scales"); assert (q == qo).all()
    q4, s4 = L.quantize_q4(x)
    q4o, s4o = O.quantize_q4(x)
    assert_bit_equal(s4, s4o, "q4 scales"); assert (q4 == q4o).all()


@pytest.mark.parametrize("n", [2048, 3072, 8192, 2304, 9216, 1024])
def test_quantize_adversarial_rounding_boundaries(L, n):
    """The device quantiser multiplies by 1/scale and redoes the exact division only near a rounding boundary:
    inputs sitting exactly on / a few ulp around (k + 0.5) * scale, plus degenerate groups (zero, denormal, huge,
    inf, NaN maxima), must give the reference's round-half-away result (quantization.rs:57-63)."""
    rng = np.random.default_rng(7 + n)
    x = np.empty(n, np.float32)
    for g in range(n // 128):
        wmax = np.float32(np.ldexp(0.5 + rng.random(), int(rng.integers(-20, 10))))
        sc = np.float32(wmax / np.float32(127.0))
        k = rng.integers(0, 127, 128).astype(np.float32)
        v = ((k + np.float32(0.5)) * sc).astype(np.float32)
        u = v.view(np.uint32) + rng.integers(-4, 5, 128).astype(np.int64)
        v = u.astype(np.uint32).view(np.float32) * rng.choice(np.array([-1.0, 1.0], np.float32), 128)
        v[int(rng.integers(0, 128))] = wmax * rng.choice(np.array([-1.0, 1.0], np.float32))
        x[g * 128:(g + 1) * 128] = np.clip(v, -wmax, wmax)
    specials = [np.float32(0.0), np.float32(1e-42), np.float32(3e-39), np.float32(2e-38), np.float32(3e38), np.float32(np.inf), np.float32(np.nan)]
    for i, sp in enumerate(specials):
        if i + 1 < n // 128:
            grp = (rng.standard_normal(128) * 1e-3).astype(np.float32) if np.isfinite(sp) and sp != 0 else np.zeros(128, np.float32)
            grp = np.clip(grp, -abs(sp), abs(sp)) if np.isfinite(sp) else grp
            grp[3] = sp
            x[(i + 1) * 128:(i + 2) * 128] = grp
    q, s = L.quantize(x)
    qo, so = O.quantize(x)
    assert_bit_equal(s, so, "q8 scales"); assert (q == qo).all()


@pytest.mark.parametrize("n", [2048, 3072, 8192, 2304, 9216, 1024])
def test_quantize_q4_adversarial_rounding_boundaries(L, n):
    """The same for quantize_q4 (quantization.rs:69-95: nibble = clamp(round(x / scale + 8), 0, 15), scale = wmax / -8): inputs on / a few
    ulp around the boundaries k + 0.5, degenerate groups, through the static prologue quantiser of every decode shape (grouped passes at
    2048 / 8192, grouped + linear at 3072, grouped + ragged at 2304 / 9216) and the generic kernel (1024)."""
    rng = np.random.default_rng(70 + n)
    x = np.empty(n, np.float32)
    for g in range(n // 128):
        wmax = np.float32(np.ldexp(0.5 + rng.random(), int(rng.integers(-20, 10))))
        sc = np.float32(wmax / np.float32(-8.0))
        k = rng.integers(0, 16, 128).astype(np.float32)
        v = ((k + np.float32(0.5) - np.float32(8.0)) * sc).astype(np.float32)
        u = v.view(np.uint32) + rng.integers(-4, 5, 128).astype(np.int64)
        v = u.astype(np.uint32).view(np.float32)
        v[int(rng.integers(0, 128))] = wmax * rng.choice(np.array([-1.0, 1.0], np.float32))
        x[g * 128:(g + 1) * 128] = np.clip(v, -wmax, wmax)
    specials = [np.float32(0.0), np.float32(1e-42), np.float32(3e-39), np.float32(2e-38), np.float32(3e38), np.float32(np.inf), np.float32(np.nan)]
    for i, sp in enumerate(specials):
        if i + 1 < n // 128:
            grp = (rng.standard_normal(128) * 1e-3).astype(np.float32) if np.isfinite(sp) and sp != 0 else np.zeros(128, np.float32)
            grp = np.clip(grp, -abs(sp), abs(sp)) if np.isfinite(sp) else grp
            grp[3] = sp
            x[(i + 1) * 128:(i + 2) * 128] = grp
    q4, s4 = L.quantize_q4(x)
    q4o, s4o = O.quantize_q4(x)
    assert_bit_equal(s4, s4o, "q4 scales"); assert (q4 == q4o).all()


@pytest.mark.parametrize("n,unit", [(128, False), (2048, False), (2304, True), (3072, False), (4096, True)])
def test_rmsnorm(L, n, unit):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 2).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    assert_bit_equal(L.rmsnorm(x, w, 1e-5, unit), O.rmsnorm(x, w, 1e-5, unit), "rmsnorm")


@pytest.mark.parametrize("n", [1, 2, 17, 64, 145, 1000, 8192])
def test_softmax(L, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 4).astype(np.float32)
    assert_bit_equal(L.softmax(x), O.softmax(x), "softmax")


def _rand_q8(rng, o, n):
    wq = rng.integers(-127, 128, size=o * n, dtype=np.int8)
    ws = (rng.uniform(1e-4, 3e-3, size=o * n // 128)).astype(np.float32)
    return wq, ws


# the six GEMV shapes of config 2 (Llama-3.2-1B) + Llama-3B / Gemma-2B / ragged rows
@pytest.mark.parametrize("n,o", [(2048, 2048), (2048, 512), (2048, 16384), (8192, 2048), (2048, 32064), (3072, 3072), (2304, 2048),
                                 (9216, 2304), (128, 256), (256, 128), (2048, 4), (2048, 36)])
def test_matmul_q8(L, n, o):
    rng = np.random.default_rng(n * 7 + o)
    wq, ws = _rand_q8(rng, o, n)
    x = (rng.standard_normal(n) * 2).astype(np.float32)
    xq, xs = O.quantize(x)
    assert_bit_equal(L.matmul_q8(xq, xs, wq, ws, n, o), O.matmul_q8(xq, xs, wq, ws, n, o), f"matmul_q8 {n}->{o}")


def test_matmul_q8_skips_tail_rows_like_the_reference(L):
    # par_chunks_exact_mut(4): rows beyond o//4*4 are never written (functional.rs:179)
    rng = np.random.default_rng(3)
    n, o = 256, 10
    wq, ws = _rand_q8(rng, o, n)
    xq, xs = O.quantize(rng.standard_normal(n).astype(np.float32))
    got = L.matmul_q8(xq, xs, wq, ws, n, o)
    ref = O.matmul_q8(xq, xs, wq, ws, n, o)
    assert_bit_equal(got, ref, "ragged o")
    assert (got[8:] == 0).all()


def test_matmul_q8_batched_rows(L):
    rng = np.random.default_rng(4)
    n, o, sl = 256, 64, 5
    wq, ws = _rand_q8(rng, o, n)
    x = rng.standard_normal(sl * n).astype(np.float32)
    xq, xs = O.quantize(x)
    assert_bit_equal(L.matmul_q8(xq, xs, wq, ws, n, o, sl=sl), O.matmul_q8(xq, xs, wq, ws, n, o, sl=sl), "matmul_q8 sl=5")


@pytest.mark.parametrize("n,o", [(2304, 2048), (2304, 18432), (9216, 2304), (2048, 512), (128, 256), (256, 128), (8192, 2048), (3072, 100)])
def test_matmul_q4(L, n, o):
    rng = np.random.default_rng(n * 3 + o)
    wq = rng.integers(0, 256, size=o * n // 2, dtype=np.uint8)
    ws = (-rng.uniform(1e-3, 2e-2, size=o * n // 128)).astype(np.float32)
    xq, xs = O.quantize_q4((rng.standard_normal(n) * 2).astype(np.float32))
    assert_bit_equal(L.matmul_q4(xq, xs, wq, ws, n, o), O.matmul_q4(xq, xs, wq, ws, n, o), f"matmul_q4 {n}->{o}")


# ------------------------------------------------------------------ whole path, golden fixtures
GOLDEN = ["tiny_llama_q8", "tiny_llama_q4", "tiny_phi_q8", "tiny_gemma_q8", "tiny_gemma_q4", "tiny_llama_f32"]


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_fixture_forward(L, golden_dir, name):
    img = np.fromfile(os.path.join(golden_dir, name + ".lmrs"), np.uint8)
    toks_gold = np.load(os.path.join(golden_dir, name + ".tokens.npy"))
    logits_gold = np.load(os.path.join(golden_dir, name + ".logits.npy"))
    m = L.Transformer(img); orc = O.Oracle(img)
    assert m.bytes_consumed == orc.bytes_consumed == img.size
    cfg_seed = {"tiny_llama_q8": ("tiny-llama", 7), "tiny_llama_q4": ("tiny-llama", 7), "tiny_phi_q8": ("tiny-phi", 9),
                "tiny_gemma_q8": ("tiny-gemma", 8), "tiny_gemma_q4": ("tiny-gemma", 8), "tiny_llama_f32": ("tiny-llama", 7)}[name]
    prompt = S.prompt_tokens(*cfg_seed[:1], 5, cfg_seed[1])
    seq = list(prompt) + list(toks_gold[:-1])
    lg = None
    for pos, t in enumerate(seq):
        lg = m.forward(int(t), pos).copy()
        assert_bit_equal(lg, orc.forward(int(t), pos), f"{name} logits at pos {pos}")
    assert_bit_equal(lg, logits_gold, f"{name} last-step logits vs committed golden")
    m2 = L.Transformer(img)
    assert (m2.generate_greedy(prompt, len(toks_gold)) == toks_gold).all()


@pytest.mark.parametrize("cfg", ["mini-llama", "mini-llama3b", "mini-phi", "mini-llama8b", "mini-llama-v4102"])
def test_mini_models_logits_bit_exact(L, cfg):
    img = S.build_image(cfg, S.Q8_0, seed=11)
    m = L.Transformer(img); orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, 6, 11)
    tok = None
    for pos in range(20):
        t = int(prompt[pos]) if pos < len(prompt) else tok
        lg = m.forward(t, pos)
        lo = orc.forward(t, pos)
        assert_bit_equal(lg, lo, f"{cfg} logits at pos {pos}")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
        assert m.forward_argmax(t, pos) == tok        # re-running a position is idempotent
    for l in range(m.args.n_layers):                   # the KV cache rows behind those logits, every layer, first / middle / last position
        for pos in (0, 7, 19):
            for which, nm in ((0, "key"), (1, "value")):
                assert_bit_equal(m.kv_row(which, l, pos), orc.kv_row(which, l, pos), f"{cfg} {nm} row, layer {l}, pos {pos}")


@pytest.mark.parametrize("cfg,q", [("mini-llama", S.Q4_0), ("mini-gemma", S.Q4_0), ("mini-gemma", S.Q8_0), ("mini-gemma9b", S.Q8_0), ("tiny-wide-att", S.Q4_0)])
def test_mini_q4_and_gemma_logits_bit_exact(L, cfg, q):
    """Q4_0 weights/activations and the Gemma variant (GELU, four norms, soft-caps, head 256): BASELINE configs[2] shapes.
    mini-gemma9b / tiny-wide-att: n_heads * head_size > dim (Gemma-2-9B, transformer.rs:497-499) on the generic kernels."""
    img = S.build_image(cfg, q, seed=12)
    m = L.Transformer(img); orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, 4, 12)
    tok = None
    for pos in range(12):
        t = int(prompt[pos]) if pos < len(prompt) else tok
        lo = orc.forward(t, pos)
        assert_bit_equal(m.forward(t, pos), lo, f"{cfg} q{q} logits at pos {pos}")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))


@pytest.mark.parametrize("i", range(18))
def test_random_geometries_single_token_and_batched(L, i):
    """Geometries no kernel was tuned or instantiated for (random family, heads, kv heads, head size, hidden, depth and a vocabulary
    that is not a multiple of 4), in each weight format: single-token steps, then a batched prefill and decode steps on its cache."""
    rng = np.random.default_rng(2000 + i)
    cfg = S.random_cfg(rng, i, max_pos=128)
    q = [S.Q8_0, S.Q4_0, S.Q_NONE][i % 3]
    img = S.build_image(cfg, q, seed=70 + i, threads=1)
    m = L.Transformer(img); orc = O.Oracle(img)
    tok = int(rng.integers(0, cfg.vocab_size))
    for pos in range(5):
        lo = orc.forward(tok, pos)
        assert_bit_equal(m.forward(tok, pos), lo, f"{cfg} q{q} logits at pos {pos}")
        nxt = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
        assert m.forward_argmax(tok, pos) == nxt
        tok = nxt
    n_tok = int(rng.integers(18, 90))
    toks = S.prompt_tokens(cfg, n_tok, 70 + i)
    a = m.get_embeddings(toks); b = orc.get_embeddings(toks)
    assert m.fill_kv_cache(a, 5) == orc.fill_kv_cache(b, 5) == 5 + n_tok
    assert_bit_equal(a, b, f"{cfg} q{q}: residual stream after {n_tok} batched tokens")
    for pos in range(5 + n_tok, 5 + n_tok + 2):
        lo = orc.forward(tok, pos)
        assert_bit_equal(m.forward(tok, pos), lo, f"{cfg} q{q}: decode at {pos} on the prefilled cache")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))


@pytest.mark.parametrize("n,o,sl", [(256, 64, 5), (2048, 3072, 70), (8192, 2048, 33), (3072, 48, 129), (256, 16384, 130), (512, 16400, 257)])
def test_matmul_q8_token_batch_on_matrix_cores(L, n, o, sl):
    """matmul_q8 over sl tokens (functional.rs:173-214 with sl > 1): the int8-MFMA kernel of the batched forward_layer.
    Integer group sums are exact whatever their order; the float combine keeps the reference's group order per element:
    bit-equal to the CPU path.  Ragged token counts (not multiples of 16 / 64) and several token blocks; the last two shapes
    have enough 128 x 128 tiles for the LDS-tiled kernel (ragged rows and tokens there too)."""
    rng = np.random.default_rng(n + o + sl)
    wq, ws = _rand_q8(rng, o, n)
    x = (rng.standard_normal(sl * n) * rng.uniform(0.1, 4.0, sl).repeat(n)).astype(np.float32)
    xq, xs = O.quantize(x)
    got = L.matmul_q8(xq, xs, wq, ws, n, o, sl=sl)
    ref = O.matmul_q8(xq, xs, wq, ws, n, o, sl=sl)
    assert_bit_equal(got, ref, f"gemm {n}x{o}, {sl} tokens")


@pytest.mark.parametrize("n,o,sl,gemma", [(256, 16384, 130, False), (512, 16384, 500, False), (256, 18432, 300, True), (1024, 8192, 512, False)])
def test_w13_with_the_quantiser_of_h_in_its_epilogue(L, n, o, sl, gemma):
    """From a few hundred tokens on fill_kv_cache never forms the hidden vector in f32: the w1/w3 GEMM's 256-row tiles (gate / up rows
    interleaved) hold one quantisation group of h per token, and the epilogue applies SiLU(gate) * up (Gemma: GELU) and the reference's
    quantize (transformer.rs:588-630, quantization.rs:44-67).  Against the oracle's three steps - matmul_q8 over the batch, the
    activation, quantize - the int8 values and the scales are bit-equal; both tile forms (256 x 128, and 256 x 64 where the cost model
    asks for 128 x 128), ragged token counts, groups whose values span six orders of magnitude, an all-zero group."""
    rng = np.random.default_rng(n + o + sl)
    wq, ws = _rand_q8(rng, o, n)
    ws = ws.reshape(o, -1)
    ws[:512] *= np.float32(1e-3); ws[512:514] = 0.0; ws[1024:1280:2] *= np.float32(300.0)    # tiny groups of h, one exact zero inside a group, large gates
    ws[768:1024] = 0.0                                                                        # ... and one group of h (values 384..511 of every token) that is all zeros: scale 0, NaN quotients -> 0
    ws = ws.reshape(-1)
    x = (rng.standard_normal(sl * n) * rng.uniform(0.1, 4.0, sl).repeat(n)).astype(np.float32)
    xq, xs = O.quantize(x)
    hq, hs = L.w13_quant(xq, xs, wq, ws, n, o, sl, gemma=gemma)
    out = O.matmul_q8(xq, xs, wq, ws, n, o, sl=sl).reshape(sl, o)
    h = O.glu(out[:, 0::2], out[:, 1::2], gemma=gemma)
    rq, rs = O.quantize(h.reshape(-1))
    assert_bit_equal(hs, rs, f"scales of h, {n}x{o}, {sl} tokens")
    assert np.array_equal(hq, rq), f"quantised h, {n}x{o}, {sl} tokens: {(hq != rq).sum()} of {hq.size} differ"


@pytest.mark.parametrize("cfg,q,n_tok,pos0", [("mini-llama", S.Q8_0, 70, 5), ("mini-llama3b", S.Q8_0, 33, 0), ("mini-phi", S.Q8_0, 140, 2),
                                              ("mini-llama-long", S.Q8_0, 600, 3), ("mini-llama", S.Q4_0, 70, 5), ("mini-gemma", S.Q8_0, 50, 3),
                                              ("mini-llama-long", S.Q8_0, 530, 0),     # a 512-token pass on the ring kernels (transposed scales), then 18 tokens on the direct ones (row-major)
                                              ("mini-gemma", S.Q4_0, 75, 0),
                                              # 48 <= tokens < 64: the LDS-DMA ring GEMM's smallest batches (one ragged token tile)
                                              ("mini-llama", S.Q8_0, 48, 1), ("mini-llama3b", S.Q8_0, 57, 0), ("mini-phi", S.Q8_0, 63, 2)])
def test_fill_kv_cache_batched_prefill(L, cfg, q, n_tok, pos0):
    """forward_layer(sl = n) as GEMMs over the token batch (more than one 64-token block; 600 tokens: more than one 512-token chunk):
    the mutated embeddings, and the decode steps that continue on the prefilled KV cache, are bit-identical to the CPU path.
    Q4_0: packed weights unpacked into MFMA fragments, Q4 activations; Gemma: folded norm+add rows, GELU, soft-capped scores and
    the window quirk of a batched call."""
    img = S.build_image(cfg, q, seed=21)
    m = L.Transformer(img); orc = O.Oracle(img)
    toks = S.prompt_tokens(cfg, n_tok, 21)
    a = m.get_embeddings(toks); b = orc.get_embeddings(toks)
    assert m.fill_kv_cache(a, pos0) == orc.fill_kv_cache(b, pos0) == pos0 + n_tok
    assert_bit_equal(a, b, "residual stream after the batched layers")
    t = 7
    for pos in range(pos0 + n_tok, pos0 + n_tok + 3):
        lo = orc.forward(t, pos)
        assert_bit_equal(m.forward(t, pos), lo, f"decode at {pos} on the prefilled cache")
        t = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))


@pytest.mark.parametrize("q,n_tok,pos0", [(S.Q8_0, 150, 7), (S.Q4_0, 90, 0)])
def test_gemma_batched_attention_in_its_long_batch_forms(L, monkeypatch, q, n_tok, pos0):
    """Gemma-2's block attention picks its forms by how full the chip is: up to a few hundred tokens 32-key score chunks on 8 waves and
    32-dim value slices (what every other Gemma test here runs), beyond that 64-key chunks on 4 waves and 64-dim slices.
    LMRS_ATT_LONG_BATCH_FORMS forces the latter at a length the CPU path finishes quickly: the same bit-equality, RoPE in the staging,
    soft-cap and window term included, with and without keys from an earlier call in the cache."""
    monkeypatch.setenv("LMRS_ATT_LONG_BATCH_FORMS", "1")
    img = S.build_image("mini-gemma", q, seed=29)
    m = L.Transformer(img); orc = O.Oracle(img)
    if pos0:
        warm = S.prompt_tokens("mini-gemma", pos0, 5)
        a0 = m.get_embeddings(warm); b0 = orc.get_embeddings(warm)
        assert m.fill_kv_cache(a0, 0) == orc.fill_kv_cache(b0, 0) == pos0
    toks = S.prompt_tokens("mini-gemma", n_tok, 29)
    a = m.get_embeddings(toks); b = orc.get_embeddings(toks)
    assert m.fill_kv_cache(a, pos0) == orc.fill_kv_cache(b, pos0) == pos0 + n_tok
    assert_bit_equal(a, b, "residual stream after the batched layers (long-batch attention forms)")
    lo = orc.forward(3, pos0 + n_tok)
    assert_bit_equal(m.forward(3, pos0 + n_tok), lo, "decode on the prefilled cache")


@pytest.mark.parametrize("cfg,n_tok,pos0", [("mini-llama", 150, 9), ("mini-llama3b", 70, 0)])
def test_batched_attention_with_memory_resident_scores(L, monkeypatch, cfg, n_tok, pos0):
    """Beyond 2048 keys the block softmax keeps its scores in the slab instead of LDS; LMRS_ATT_LDS_KEYS forces that variant at a
    length the CPU path finishes quickly.  Same bit-equality as the LDS variant."""
    monkeypatch.setenv("LMRS_ATT_LDS_KEYS", "32")
    img = S.build_image(cfg, S.Q8_0, seed=23)
    m = L.Transformer(img); orc = O.Oracle(img)
    if pos0:
        warm = S.prompt_tokens(cfg, pos0, 5)
        a0 = m.get_embeddings(warm); b0 = orc.get_embeddings(warm)
        assert m.fill_kv_cache(a0, 0) == orc.fill_kv_cache(b0, 0) == pos0
    toks = S.prompt_tokens(cfg, n_tok, 23)
    a = m.get_embeddings(toks); b = orc.get_embeddings(toks)
    assert m.fill_kv_cache(a, pos0) == orc.fill_kv_cache(b, pos0) == pos0 + n_tok
    assert_bit_equal(a, b, "residual stream after the batched layers (scores in memory)")
    lo = orc.forward(3, pos0 + n_tok)
    assert_bit_equal(m.forward(3, pos0 + n_tok), lo, "decode on the prefilled cache")


@pytest.mark.parametrize("cfg,n_prompt", [("mini-llama", 200), ("mini-phi", 90)])
def test_generate_greedy_with_a_long_prompt(L, cfg, n_prompt):
    """generate_greedy feeds all prompt tokens but the last through the batched forward_layer (only their K/V rows matter,
    chat.rs:188-193 discards those logits): same token ids as the token-by-token CPU path, and as the per-token device path."""
    img = S.build_image(cfg, S.Q8_0, seed=31)
    prompt = S.prompt_tokens(cfg, n_prompt, 31)
    m = L.Transformer(img)
    got = m.generate_greedy(prompt, 20)
    ref = O.Oracle(img).generate_greedy(prompt, 20)
    assert (got == ref).all(), f"first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    # continuing from the same cache with the per-token API gives the oracle's logits
    orc = O.Oracle(img); orc.generate_greedy(prompt, 20)
    pos = n_prompt + 19
    assert_bit_equal(m.forward(int(got[-1]), pos), orc.forward(int(ref[-1]), pos), "decode after prompt + 20 tokens")


@pytest.mark.parametrize("cfg", ["mini-llama", "mini-gemma", "mini-phi", "mini-llama8b"])
def test_unquantised_models_logits_bit_exact(L, cfg):
    """q_type None (f32 weights, the reference's `matmul`, functional.rs:142-171: chunk sums through wide's tree, added to the row in
    chunk order): logits bit-equal at every step at the real head geometries, Gemma's separate norm + add launches, GELU and soft-cap,
    Phi's lm_head; get_embeddings / fill_kv_cache (token by token) / greedy generation on top."""
    img = S.build_image(cfg, S.Q_NONE, seed=17)
    m = L.Transformer(img); orc = O.Oracle(img)
    assert m.bytes_consumed == orc.bytes_consumed == img.size
    prompt = S.prompt_tokens(cfg, 4, 17)
    tok = None
    for pos in range(8):
        t = int(prompt[pos]) if pos < len(prompt) else tok
        lo = orc.forward(t, pos)
        assert_bit_equal(m.forward(t, pos), lo, f"{cfg} f32 logits at pos {pos}")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
    e_dev = m.get_embeddings(prompt); e_ref = orc.get_embeddings(prompt)
    assert_bit_equal(e_dev, e_ref, "get_embeddings (f32 table)")
    a = e_dev.copy(); b = e_ref.copy()
    assert m.fill_kv_cache(a, 8) == orc.fill_kv_cache(b, 8) == 12
    assert_bit_equal(a, b, "fill_kv_cache (f32)")
    m2 = L.Transformer(img); o2 = O.Oracle(img)
    assert (m2.generate_greedy(prompt, 12) == o2.generate_greedy(prompt, 12)).all()


@pytest.mark.parametrize("cfg,q", [("mini-phi", S.Q8_0), ("mini-gemma", S.Q4_0)])
def test_get_embeddings_and_fill_kv_cache(L, cfg, q):
    """mini-gemma: the folded residual form (x += rmsnorm(branch) inside the next GEMV's prologue) must hand the FINISHED
    residual stream back to fill_kv_cache's caller."""
    img = S.build_image(cfg, q, seed=13)
    m = L.Transformer(img); orc = O.Oracle(img)
    toks = S.prompt_tokens(cfg, 9, 13)
    e_dev = m.get_embeddings(toks); e_ref = orc.get_embeddings(toks)
    assert_bit_equal(e_dev, e_ref, "get_embeddings")
    # empty input is legal in the reference (returns an empty Vec)
    assert m.get_embeddings(np.zeros(0, np.uint32)).size == 0
    a = e_dev.copy(); b = e_ref.copy()
    assert m.fill_kv_cache(a, 3) == orc.fill_kv_cache(b, 3) == 3 + len(toks)
    assert_bit_equal(a, b, "fill_kv_cache mutates the embeddings identically")
    # decode continues on top of the batched prefill and stays bit-identical
    assert_bit_equal(m.forward(5, 12), orc.forward(5, 12), "decode after fill_kv_cache")


# ------------------------------------------------------------------ sample_argmax edge cases (sampler.rs:29-41)
def _cls_case(rows_scale, rows_k, n=256):
    """One activation spike in group 0 -> logit[r] = ((127 * k_r) as f32 * ws[r][0]) * xs[0] (+ an exact zero from group 1): any
    float - NaN, +-inf, ties - can be placed at any row through ws."""
    o = len(rows_scale)
    x = np.zeros(n, np.float32); x[0] = 3.0
    w = np.ones(n, np.float32)
    wq = np.zeros((o, n), np.int8); wq[:, 0] = rows_k
    ws = np.ones((o, n // 128), np.float32); ws[:, 0] = rows_scale
    return x, w, wq, ws


@pytest.mark.parametrize("case", ["tie_first_wins", "nan_at_zero", "nan_elsewhere", "all_minus_inf", "max_in_last_row", "tie_across_workgroups", "plus_inf_tie"])
def test_argmax_edge_cases(L, case):
    """Transformer::forward's classifier + Sampler::sample_argmax as the decode step runs them (EPI_CLS partials + argmax_final):
    the reference starts at index 0 and moves on a strict `>` - first index of the maximum, a NaN at index 0 is never displaced,
    NaNs elsewhere never win, a row of -inf answers 0."""
    o = 4096
    rng = np.random.default_rng(3)
    sc = rng.uniform(0.5, 1.5, o).astype(np.float32); k = rng.integers(-100, 100, o).astype(np.int8)
    if case == "tie_first_wins": sc[:] = 1.0; k[:] = 5; k[[77, 900, 3000]] = 120
    elif case == "nan_at_zero": sc[0] = np.nan; k[0] = 1
    elif case == "nan_elsewhere": sc[[5, 2047, 4095]] = np.nan; k[[5, 2047, 4095]] = 1; sc[1234] = 9.0; k[1234] = 127
    elif case == "all_minus_inf": sc[:] = -np.inf; k[:] = 1
    elif case == "max_in_last_row": sc[o - 1] = 50.0; k[o - 1] = 127
    elif case == "tie_across_workgroups": sc[:] = 1.0; k[:] = -3; k[[2500, 40]] = 99
    elif case == "plus_inf_tie": sc[[3000, 17]] = np.inf; k[[3000, 17]] = 1
    x, w, wq, ws = _cls_case(sc, k)
    tok, lg = L.classifier_argmax(x, w, wq, ws, 1e-5)
    xq, xs = O.quantize(O.rmsnorm(x, w, 1e-5))
    ref = O.matmul_q8(xq, xs, wq.reshape(-1), ws.reshape(-1), x.size, o)
    assert_bit_equal(lg, ref, f"{case}: logits")
    want = int(O.lib().lmrs_ref_argmax(ref.ctypes.data, ref.size))
    assert tok == want, f"{case}: device {tok}, reference {want} (logit[tok] = {ref[tok]}, logit[want] = {ref[want]})"
    expect = {"tie_first_wins": 77, "nan_at_zero": 0, "nan_elsewhere": 1234, "all_minus_inf": 0, "max_in_last_row": o - 1,
              "tie_across_workgroups": 40, "plus_inf_tie": 17}[case]
    assert want == expect, f"{case}: the oracle itself answered {want}, the reference's rule gives {expect}"


# ------------------------------------------------------------------ BASELINE config 2 at full size
def test_llama_1b_q8_greedy_token_ids(L):
    """BASELINE.json configs[0]/[1]: Llama-3.2-1B Q8_0, greedy, 16-token prompt + 128 generated tokens;
    token IDs must be identical to the CPU path."""
    cfg = "llama-3.2-1b"
    img = S.build_image(cfg, S.Q8_0, seed=1234)
    prompt = S.prompt_tokens(cfg, 16, 1234)
    m = L.Transformer(img)
    got, sec = m.generate_greedy(prompt, 128, timing=True)
    orc = O.Oracle(img)
    ref = orc.generate_greedy(prompt, 128)
    assert (got == ref).all(), f"first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    # logits of one more step, bit for bit
    lg = m.forward(int(ref[-1]), 16 + 127)
    assert_bit_equal(lg, orc.forward(int(ref[-1]), 16 + 127), "1B logits at pos 143")
    print(f"\n1B greedy 143 steps: {sec*1e3:.1f} ms on device = {143/sec:.0f} tok/s")


def test_llama_1b_q8_long_prompt_at_full_size(L):
    """Llama-3.2-1B Q8_0 at full size with a 600-token prompt: two chunks of the batched prefill (int8 matrix-core GEMMs,
    block attention), then 40 decode steps on the split-attention graph - token ids identical to the token-by-token CPU path,
    logits of one more step bit-equal.  (Checked once at 1100 tokens as well: three chunks, the 1024..2047 bucket.)"""
    cfg = "llama-3.2-1b"
    img = S.build_image(cfg, S.Q8_0, seed=1234)
    prompt = S.prompt_tokens(cfg, 600, 99)
    m = L.Transformer(img)
    got = m.generate_greedy(prompt, 40)
    orc = O.Oracle(img)
    ref = orc.generate_greedy(prompt, 40)
    assert (got == ref).all(), f"first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    pos = 600 + 39
    assert_bit_equal(m.forward(int(ref[-1]), pos), orc.forward(int(ref[-1]), pos), f"1B logits at pos {pos}")


@pytest.mark.parametrize("cfg", ["llama-3.2-3b", "phi-3.5"])
def test_full_size_3b_and_phi_q8_greedy_token_ids(L, cfg):
    """BASELINE.json configs[3] (Llama-3.2-3B, 28 layers, on one GPU) and configs[4]'s text model (Phi-3.5, 32 layers, LongRoPE
    factors, separate lm_head) at FULL size, Q8_0: 16 prompt tokens + 32 greedy tokens identical to the CPU path, the logits of
    one more step bit-equal, and KV rows of the first / last layer."""
    img = S.build_image(cfg, S.Q8_0, seed=2024)
    prompt = S.prompt_tokens(cfg, 16, 2024)
    m = L.Transformer(img)
    got, sec = m.generate_greedy(prompt, 32, timing=True)
    orc = O.Oracle(img)
    ref = orc.generate_greedy(prompt, 32)
    assert (got == ref).all(), f"{cfg}: first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    pos = 16 + 31
    assert_bit_equal(m.forward(int(ref[-1]), pos), orc.forward(int(ref[-1]), pos), f"{cfg} logits at pos {pos}")
    for l in (0, m.args.n_layers - 1):
        for which in (0, 1):
            assert_bit_equal(m.kv_row(which, l, pos), orc.kv_row(which, l, pos), f"{cfg} kv[{which}] layer {l} pos {pos}")
    print(f"\n{cfg} q8_0 full size: {47 / sec:.0f} tok/s")


def test_gemma_2b_q4_greedy_token_ids(L):
    """BASELINE.json configs[2]: Gemma-2-2B Q4_0 at full size, greedy token ids identical to the CPU path."""
    cfg = "gemma-2-2b"
    img = S.build_image(cfg, S.Q4_0, seed=77)
    prompt = S.prompt_tokens(cfg, 8, 77)
    m = L.Transformer(img)
    got, sec = m.generate_greedy(prompt, 24, timing=True)
    ref = O.Oracle(img).generate_greedy(prompt, 24)
    assert (got == ref).all(), f"first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    print(f"\ngemma-2-2b q4_0: {31/sec:.0f} tok/s")


def test_gemma_2b_q4_batched_prefill_at_full_size(L):
    """Gemma-2-2B Q4_0 at full size through fill_kv_cache (forward_layer over a batch, transformer.rs:672-684): 300 embeddings - the
    packed-nibble weights through the ring kernel's 256 x 128 (gate / up), 128 x 128 and 64 x 64 tiles, the folded norm + add rows, GELU,
    soft-capped attention with the window - the residual stream that comes back and KV rows bit-equal to the CPU path's."""
    cfg = "gemma-2-2b"
    img = S.build_image(cfg, S.Q4_0, seed=77)
    toks = S.prompt_tokens(cfg, 300, 78)
    m = L.Transformer(img); orc = O.Oracle(img)
    emb = m.get_embeddings(toks)
    assert_bit_equal(emb, orc.get_embeddings(toks), "embeddings")
    a = emb.copy(); b = emb.copy()
    assert m.fill_kv_cache(a, 0) == 300 and orc.fill_kv_cache(b, 0) == 300
    assert_bit_equal(a, b, "residual stream after 300 batched tokens")
    for l in (0, m.args.n_layers - 1):
        for which in (0, 1):
            for pos in (0, 150, 299):
                assert_bit_equal(m.kv_row(which, l, pos), orc.kv_row(which, l, pos), f"kv[{which}] layer {l} pos {pos}")


@pytest.mark.parametrize("cfg", ["mini-llama", "mini-gemma"])
def test_fill_kv_cache_on_q4_files_is_the_decode_form_not_the_references_q9(L, cfg):
    """The one documented deviation (INTEGRATION.md "Deviations"): the reference's batched matmul_q4 multiplies token j with token 2j's
    nibbles.  The library returns what the oracle's DEFAULT mode returns - every token through the decode form - bit for bit, and NOT what the
    oracle's reference-faithful mode returns (tests/test_oracle.py pins what that is); token 0, whose offset is 0 either way, agrees with both."""
    img = S.build_image(cfg, S.Q4_0, seed=23)
    toks = S.prompt_tokens(cfg, 40, 23)
    m = L.Transformer(img); o = O.Oracle(img)
    a = m.get_embeddings(toks); b = o.get_embeddings(toks)
    assert m.fill_kv_cache(a, 0) == o.fill_kv_cache(b, 0) == 40
    assert_bit_equal(a, b, f"{cfg} Q4_0: batched fill_kv_cache against the oracle's default (decode-form) mode")
    with O.faithful_q9():
        f = O.Oracle(img).get_embeddings(toks)
        of = O.Oracle(img)
        assert of.fill_kv_cache(f, 0) == 40
    dim = m.args.dim
    assert_bit_equal(a[:dim], f[:dim], "token 0 is the same arithmetic in both modes")
    assert (bits(a[dim:]) != bits(f[dim:])).mean() > 0.9, "the library must NOT reproduce the reference's Q9 offsets"


@pytest.mark.parametrize("cfg,n_steps", [("mini-llama-long", 700), ("mini-phi-long", 400)])
def test_long_context_multi_chunk_attention(L, cfg, n_steps):
    """Positions beyond one LDS chunk of K/V rows (256 timesteps at head 64, 160 at head 96): the chunked score and
    value loops, the serial softmax sum over hundreds of terms.  Token ids over the whole run + bit-equal logits at the end."""
    img = S.build_image(cfg, S.Q8_0, seed=51)
    prompt = S.prompt_tokens(cfg, 8, 51)
    m = L.Transformer(img); orc = O.Oracle(img)
    got = m.generate_greedy(prompt, n_steps)
    ref = orc.generate_greedy(prompt, n_steps)
    assert (got == ref).all(), f"first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    pos = 8 + n_steps - 1
    assert_bit_equal(m.forward(int(ref[-1]), pos), orc.forward(int(ref[-1]), pos), f"{cfg} logits at pos {pos}")


@pytest.mark.parametrize("cfg,q,n_steps,split_pos", [("mini-llama-long", S.Q8_0, 1100, 40), ("mini-phi-long", S.Q8_0, 300, 20), ("mini-llama3b", S.Q8_0, 200, 16),
                                                     ("mini-gemma", S.Q4_0, 200, 16)])
def test_split_attention_for_long_contexts(L, monkeypatch, cfg, q, n_steps, split_pos):
    """From LMRS_ATT_SPLIT_POS on (default 384) a decode step computes its scores by (head, 256-key chunk) and the softmax + V
    chains by (head, quarter of the head dims) - two launches instead of one workgroup per head.  Forced early here: the switch
    between the two step graphs, both context buckets of the split graph (below / above 1024 positions), every head size and
    the Gemma score path.  Token ids over the whole run + bit-equal logits at the end."""
    monkeypatch.setenv("LMRS_ATT_SPLIT_POS", str(split_pos))
    img = S.build_image(cfg, q, seed=52)
    prompt = S.prompt_tokens(cfg, 8, 52)
    m = L.Transformer(img); orc = O.Oracle(img)
    got = m.generate_greedy(prompt, n_steps)
    ref = orc.generate_greedy(prompt, n_steps)
    assert (got == ref).all(), f"first mismatch at {int(np.flatnonzero(got != ref)[0])}"
    pos = 8 + n_steps - 1
    assert_bit_equal(m.forward(int(ref[-1]), pos), orc.forward(int(ref[-1]), pos), f"{cfg} logits at pos {pos}")


# ------------------------------------------------------------------ row sharding
@pytest.mark.parametrize("cfg,q,world,mode", [
    ("mini-llama", S.Q8_0, 2, "int8"), ("mini-llama", S.Q8_0, 8, "int8"), ("mini-llama3b", S.Q8_0, 4, "int8"), ("mini-phi", S.Q8_0, 8, "int8"),
    ("mini-llama", S.Q8_0, 4, "f32"), ("mini-llama", S.Q8_0, 4, "split"), ("mini-llama3b", S.Q8_0, 8, "split"),
    ("mini-gemma", S.Q8_0, 2, "int8"), ("mini-gemma", S.Q8_0, 4, "split"), ("mini-gemma", S.Q4_0, 4, "int8"), ("mini-llama", S.Q4_0, 4, "int8"),
    ("mini-llama", S.Q8_0, 2, "p2p"), ("mini-llama3b", S.Q8_0, 2, "p2p"), ("mini-gemma", S.Q8_0, 2, "p2p-split"),
    ("mini-llama-v4102", S.Q8_0, 2, "int8"),
    ("mini-llama", S.Q8_0, 2, "cls"), ("mini-llama", S.Q8_0, 8, "cls"), ("mini-gemma", S.Q4_0, 4, "cls"), ("mini-phi", S.Q8_0, 2, "cls-p2p"), ("mini-llama-v4102", S.Q8_0, 2, "cls")])
def test_row_sharding_is_bit_identical(L, monkeypatch, cfg, q, world, mode):
    """`world` logical shards on ONE device - same kernels, same partition as the multi-GPU path: every logit must equal the unsharded
    CPU path bit for bit.  int8: att_out / h travel quantised by their producers (Q8_0; Q4_0 models fall back to f32 slices);
    f32: LMRS_SHARD_F32_PAYLOAD=1; split: wo / w2 row-split as well (four exchanges per layer); Gemma-2: the norm + add steps
    around the exchanges; p2p: the shards run CONCURRENTLY on their own streams and exchange through the push kernel (stores into
    the peers' arenas + flags) instead of lock-step copies (two shards only: a process has four hardware queues by default, and
    with more shards in ONE process two of them share a queue - the second's kernels would sit behind the first's waiting exchange
    kernel; separate processes, the real launch shape, each have their own: test_peer_to_peer_shards_in_separate_processes)."""
    # cls: the layers whole on every shard, only the classifier's rows split (the plan the library picks by itself for models this small;
    # every other mode pins the row-split plan)
    monkeypatch.setenv("LMRS_SHARD_PLAN", "cls" if mode.startswith("cls") else "tp")
    if mode == "f32": monkeypatch.setenv("LMRS_SHARD_F32_PAYLOAD", "1")
    if "split" in mode: monkeypatch.setenv("LMRS_SHARD_SPLIT_OUT", "1")
    if "p2p" in mode: monkeypatch.setenv("LMRS_GROUP_P2P", "1")
    img = S.build_image(cfg, q, seed=31)
    grp = L.ShardGroup(img, world)
    orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, 4, 31)
    tok = None
    for pos in range(10):
        t = int(prompt[pos]) if pos < len(prompt) else tok
        lg, nxt = grp.forward(t, pos)
        lo = orc.forward(t, pos)
        assert_bit_equal(lg, lo, f"{cfg} world={world} {mode} logits at pos {pos}")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
        assert nxt == tok
    grp.close()


def test_row_sharding_full_size_3b(L, monkeypatch):
    """BASELINE.json configs[3] in its sharded form at FULL size: Llama-3.2-3B Q8_0 (28 layers, 24 query / 8 kv heads of 128, 128 256-row
    vocabulary), every weight matrix row-split over 8 shards (plan `tp`: 3 query heads + 1 kv head, 1024 gate/up pairs and 16 032
    classifier rows per shard; quantised int8 payloads), the 8 shards run in lock step on one device.  4 prompt + 8 greedy steps:
    logits bit-equal to the CPU path at every step - the static / generic kernel choice for the shard shapes and the 16 032-row
    classifier shard included."""
    monkeypatch.setenv("LMRS_SHARD_PLAN", "tp")
    cfg = "llama-3.2-3b"
    img = S.build_image(cfg, S.Q8_0, seed=2024)
    grp = L.ShardGroup(img, 8)
    orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, 4, 2024)
    tok = None
    for pos in range(12):
        t = int(prompt[pos]) if pos < len(prompt) else tok
        lg, nxt = grp.forward(t, pos)
        lo = orc.forward(t, pos)
        assert_bit_equal(lg, lo, f"3B tp8 logits at pos {pos}")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
        assert nxt == tok
    grp.close()


@pytest.mark.parametrize("i", range(8))
def test_row_sharding_on_random_geometries(L, monkeypatch, i):
    """Two logical shards of geometries nothing was written against (slices below one quantisation group fall back to f32 payloads;
    vocabularies with an unwritten tail; every family), full split on the odd cases."""
    import dataclasses
    rng = np.random.default_rng(3000 + i)
    while True:
        cfg = S.random_cfg(rng, i, max_pos=64)
        if cfg.n_kv_heads == 2:
            break
    cfg = dataclasses.replace(cfg, vocab_size=cfg.vocab_size + cfg.vocab_size % 2)           # the world must divide the vocabulary
    monkeypatch.setenv("LMRS_SHARD_PLAN", "tp" if i < 6 else "cls")
    if i % 2: monkeypatch.setenv("LMRS_SHARD_SPLIT_OUT", "1")
    q = [S.Q8_0, S.Q4_0][(i // 2) % 2]
    img = S.build_image(cfg, q, seed=90 + i, threads=1)
    grp = L.ShardGroup(img, 2); orc = O.Oracle(img)
    tok = int(rng.integers(0, cfg.vocab_size))
    for pos in range(8):
        lg, nxt = grp.forward(tok, pos)
        lo = orc.forward(tok, pos)
        assert_bit_equal(lg, lo, f"{cfg} q{q} two shards, logits at pos {pos}")
        tok = int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
        assert nxt == tok
    grp.close()


def _p2p_rank(rank, world, img_path, cfg, env, q_in, q_out, n_fill=6, n_prompt=3):
    """One process of the peer-to-peer test: its shard of the model on device 0, handles exchanged through the parent."""
    try:
        import os, sys
        os.environ.update(env)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
        import numpy as np
        import lmrs_amd
        from tools import synth_lmrs as S
        img = np.fromfile(img_path, np.uint8)
        m = lmrs_amd.Transformer(img, device=0, rank=rank, world=world)          # no communicator id: peer-to-peer transport
        q_out.put((rank, "handle", m.p2p_handle()))
        m.p2p_connect(q_in.get(timeout=120))
        prompt = S.prompt_tokens(cfg, n_fill, 44)
        emb = m.get_embeddings(prompt)
        newp = m.fill_kv_cache(emb, 0)                                           # sharded fill_kv_cache: batched on row shards where built, else layer segments token by token
        try: m.last_fill_ms(); batched = True                                     # (only the batched forward_layer records its device time)
        except Exception: batched = False
        toks = m.generate_greedy(S.prompt_tokens(cfg, n_prompt, 45), 20, start_pos=newp)   # crosses LMRS_ATT_SPLIT_POS: split attention on shards
        lg = m.forward(int(toks[-1]), newp + n_prompt + 19).copy()
        q_out.put((rank, "done", (emb, newp, toks, lg, m.shard_uses_graph(), batched)))
        m.close()
    except Exception:
        import traceback
        q_out.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("cfg,world,plan,n_fill,n_prompt,q", [
    ("mini-llama", 2, "tp", 6, 3, S.Q8_0), ("mini-gemma", 2, "tp", 6, 3, S.Q8_0), ("mini-llama3b", 4, "tp", 6, 3, S.Q8_0), ("mini-llama", 4, "cls", 6, 3, S.Q8_0),
    ("mini-gemma", 2, "cls", 6, 3, S.Q8_0),
    ("mini-llama", 2, "tp-tokenwise", 6, 3, S.Q8_0),          # LMRS_NO_BATCHED_PREFILL: the row shards' token-by-token fill_kv_cache
    ("mini-llama", 2, "tp", 70, 12, S.Q8_0), ("mini-llama3b", 4, "tp", 70, 12, S.Q8_0), ("mini-phi", 2, "tp", 40, 9, S.Q8_0),     # batched forward_layer on row shards, block attention, batched prompt
    ("mini-gemma", 2, "tp", 70, 12, S.Q8_0), ("mini-llama", 2, "tp", 70, 12, S.Q4_0), ("mini-gemma", 2, "tp", 70, 12, S.Q4_0),    # round 6: Gemma-2 and Q4_0 on row shards
    ("mini-llama", 2, "tp-splitout", 70, 12, S.Q8_0), ("mini-llama3b", 4, "tp-splitout", 70, 12, S.Q8_0), ("mini-gemma", 2, "tp-splitout", 70, 12, S.Q4_0),    # ... and wo / w2 split too
    # 8 shards (a full node), 520 classifier rows each = 65 argmax partials per shard: an ODD count, whose 520-byte block the push transport (16 bytes per lane)
    # refused until round 6 - as it did Llama-3.2-1B's 501 partials on 8 shards (the block is padded to 16 bytes now: lmrs_ctx::part_stride)
    (S.ModelCfg("mini-llama-v4160", 2048, 8192, 2, 32, 64, 8, 4160, 256, 1e-5, 500000.0, S.LLAMA), 8, "cls", 6, 3, S.Q8_0)])
def test_peer_to_peer_shards_in_separate_processes(L, tmp_path, cfg, world, plan, n_fill, n_prompt, q):
    """The multi-GPU launch shape on a one-GPU box: `world` PROCESSES, one shard each (here all on device 0), exchange arenas opened
    through IPC handles, every exchange a push kernel that really waits for the other process's flag.  fill_kv_cache on the shards,
    greedy decoding across the split-attention switch, full logits on every rank - all bit-equal to the CPU path.  plan "tp": the
    layers' matrices row-split (exchanges inside every layer); "cls": whole layers on every shard, the classifier split (one exchange
    per token).  Plan "tp" (Q8_0 and Q4_0; Llama / Phi shapes and Gemma-2) runs fill_kv_cache and the prompt of generate_greedy as
    forward_layer over the token batch on every shard (prefill_layers: two all-gathers of quantised token-batch blocks per layer)."""
    import multiprocessing as mp
    img = S.build_image(cfg, q, seed=43)
    path = str(tmp_path / "m.lmrs"); img.tofile(path)
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue(); q_in = [ctx.Queue() for _ in range(world)]
    env = {"LMRS_ATT_SPLIT_POS": "16", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "LMRS_P2P_TIMEOUT_MS": "1500", "LMRS_SHARD_PLAN": plan.split("-")[0]}
    if plan.endswith("tokenwise"): env["LMRS_NO_BATCHED_PREFILL"] = "1"
    if plan.endswith("splitout"): env["LMRS_SHARD_SPLIT_OUT"] = "1"
    procs = [ctx.Process(target=_p2p_rank, args=(r, world, path, cfg, env, q_in[r], q_out, n_fill, n_prompt)) for r in range(world)]
    for p in procs: p.start()
    try:
        handles = {}
        while len(handles) < world:
            r, kind, val = q_out.get(timeout=90)
            assert kind == "handle", val
            handles[r] = val
        for r in range(world): q_in[r].put([handles[i] for i in range(world)])
        res = {}
        while len(res) < world:
            r, kind, val = q_out.get(timeout=90)
            assert kind == "done", val
            res[r] = val
    finally:
        for p in procs: p.join(30)
        for p in procs:
            if p.is_alive(): p.kill()
    orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, n_fill, 44)
    e_ref = orc.get_embeddings(prompt); p_ref = orc.fill_kv_cache(e_ref, 0)
    t_ref = orc.generate_greedy(S.prompt_tokens(cfg, n_prompt, 45), 20, start_pos=p_ref)
    l_ref = orc.forward(int(t_ref[-1]), p_ref + n_prompt + 19)
    for r in range(world):
        emb, newp, toks, lg, graph, batched = res[r]
        assert newp == p_ref
        assert batched == (plan in ("cls", "tp", "tp-splitout")), f"rank {r}: fill_kv_cache took the {'batched' if batched else 'token-by-token'} path"
        assert_bit_equal(emb, e_ref, f"rank {r}: embeddings after the sharded fill_kv_cache")
        assert (toks == t_ref).all(), (r, toks, t_ref)
        assert_bit_equal(lg, l_ref, f"rank {r}: logits")
        print(f"rank {r}: step graph captured = {graph}")


def _p2p_stall_rank(rank, world, img_path, cfg, env, stall_us, q_in, q_out):
    """One process of the stalled-peer test: plan "cls", steps enqueued eagerly; rank 1 spins on the device after every partials exchange."""
    try:
        import os, sys
        os.environ.update(env)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
        import numpy as np
        import lmrs_amd
        from tools import synth_lmrs as S
        img = np.fromfile(img_path, np.uint8)
        m = lmrs_amd.Transformer(img, device=0, rank=rank, world=world)
        q_out.put((rank, "handle", m.p2p_handle()))
        m.p2p_connect(q_in.get(timeout=120))
        m.debug_inject(1, 4 * m.args.n_layers, stall_us if rank == 1 else 0)     # eager steps on every rank; the stall on rank 1 only
        toks = m.generate_greedy(S.prompt_tokens(cfg, 4, 46), 40)
        q_out.put((rank, "done", toks))
        m.close()
    except Exception:
        import traceback
        q_out.put((rank, "error", traceback.format_exc()))


def test_push_exchange_with_a_stalled_peer(L, tmp_path):
    """Plan "cls" has ONE exchange per token (the argmax partials), so nothing but the exchange itself orders a shard that runs ahead
    against a peer that has not yet consumed the previous block: rank 1 stalls 2 ms on the device between every partials exchange and
    its argmax while rank 0 runs a whole step ahead and pushes the next partials.  The gathered partials are double-buffered by the
    parity of the exchange (ArgmaxArgs::part_par), so rank 1 still reads step s while step s + 1 lands in the other half: token ids
    identical to the CPU path on both ranks."""
    import multiprocessing as mp
    cfg, world = "mini-llama", 2
    img = S.build_image(cfg, S.Q8_0, seed=47)
    path = str(tmp_path / "m.lmrs"); img.tofile(path)
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue(); q_in = [ctx.Queue() for _ in range(world)]
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "LMRS_P2P_TIMEOUT_MS": "3000", "LMRS_SHARD_PLAN": "cls"}
    procs = [ctx.Process(target=_p2p_stall_rank, args=(r, world, path, cfg, env, 2000, q_in[r], q_out)) for r in range(world)]
    for p in procs: p.start()
    try:
        handles = {}
        while len(handles) < world:
            r, kind, val = q_out.get(timeout=90)
            assert kind == "handle", val
            handles[r] = val
        for r in range(world): q_in[r].put([handles[i] for i in range(world)])
        res = {}
        while len(res) < world:
            r, kind, val = q_out.get(timeout=120)
            assert kind == "done", val
            res[r] = val
    finally:
        for p in procs: p.join(30)
        for p in procs:
            if p.is_alive(): p.kill()
    ref = O.Oracle(img).generate_greedy(S.prompt_tokens(cfg, 4, 46), 40)
    for r in range(world):
        assert (res[r] == ref).all(), (r, res[r], ref)


def test_bench_fallback_is_taken_by_all_ranks_together(tmp_path):
    """The driver's multi-GPU launch (torch.distributed.run, one process per rank; here both ranks on ONE GPU over gloo) with a
    peer-to-peer connect that fails on rank 1 only (LMRS_BENCH_FAIL_RANK -> lmrs_debug_inject): rank 0, whose own connect succeeded, must drop its context
    too, both ranks must take the fallback branch together, and the run must complete with token parity.  (On one device the
    fallback is a second peer-to-peer attempt - RCCL refuses two ranks on one GPU; on a node it is the RCCL communicator.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LMRS_BENCH_ONE_DEVICE="1", LMRS_BENCH_FAIL_RANK="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29597",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--cpu-steps", "6"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    d = json.loads(lines[0])
    assert "second attempt" in d["roofline"]["transport"], d["roofline"]["transport"]
    assert "injected failure" in r.stderr                                  # rank 1 reported why it could not connect
    assert d["parity"]["tokens_equal"] and d["n_gpus"] == 2


@pytest.mark.parametrize("cfg,q,split_out", [("mini-llama", S.Q8_0, False), ("mini-gemma", S.Q4_0, False), ("mini-llama", S.Q8_0, True), ("mini-gemma", S.Q8_0, True)])
def test_rccl_path_with_one_rank(L, monkeypatch, cfg, q, split_out):
    """The RCCL code path (communicator, all-gathers between the segments, graph capture) with world = 1; split_out: the plan whose wo / w2
    rows are split too (with one rank the "slice" is all rows, but the blocks, the gathers and the scatter-add are the code that runs)."""
    if split_out: monkeypatch.setenv("LMRS_SHARD_SPLIT_OUT", "1")
    img = S.build_image(cfg, q, seed=32)
    m = L.Transformer(img, rank=0, world=1, unique_id=L.comm_unique_id())
    orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, 4, 32)
    assert (m.generate_greedy(prompt, 8) == orc.generate_greedy(prompt, 8)).all()
    o2 = O.Oracle(img)
    for pos, t in enumerate(prompt):
        assert_bit_equal(m.forward(int(t), pos), o2.forward(int(t), pos), f"rccl world=1 logits at pos {pos}")
    # the batched forward_layer of a row-sharded context over RCCL (prefill_layers_tp: quantised token-batch blocks, ncclAllGather in place, gather
    # kernel) - one rank is all a one-GPU box can give RCCL, but it is the same code: fill_kv_cache of 70 tokens, then a 12-token prompt
    toks = S.prompt_tokens(cfg, 70, 33)
    a = m.get_embeddings(toks); b = o2.get_embeddings(toks)
    assert m.fill_kv_cache(a, 4) == o2.fill_kv_cache(b, 4) == 74
    assert m.last_fill_ms() > 0                                             # (only the batched path records its device time)
    assert_bit_equal(a, b, "rccl world=1: residual stream after the batched layers")
    p2 = S.prompt_tokens(cfg, 12, 34)
    assert (m.generate_greedy(p2, 8, start_pos=74) == o2.generate_greedy(p2, 8, start_pos=74)).all()


@pytest.mark.parametrize("cfg,q", [("mini-llama", S.Q8_0), ("mini-llama3b", S.Q8_0), ("mini-phi", S.Q8_0), ("mini-gemma", S.Q8_0),
                                   ("mini-gemma", S.Q4_0), ("mini-llama", S.Q4_0)])
def test_merged_qkv_attention_launch_and_classifier_tail(L, monkeypatch, cfg, q):
    """The decode step's in-launch hand-offs: qkv + attention as ONE launch (the attention workgroups poll the {value, tag}
    granules the GEMV workgroups of the same launch write; one wave per head at the shortest contexts, one workgroup per head
    after that) and the final argmax folded into the classifier launch (workgroup 0 sweeps the tagged partials).  All forms -
    default, workgroup form only, everything as separate launches - must give the CPU path's token ids over a run that crosses
    the wave -> workgroup switch, and bit-equal logits when positions are re-run out of order (the tags come from a step counter
    that never repeats, not from the position)."""
    img = S.build_image(cfg, q, seed=41)
    orc = O.Oracle(img)
    prompt = S.prompt_tokens(cfg, 6, 41)
    n_new = 150
    ref = orc.generate_greedy(prompt, n_new)
    m = L.Transformer(img)
    nl = m.args.n_layers
    merged = m.step_info(0)[0] == 4 * nl + 1 and m.step_info(200)[0] == 4 * nl + 1
    assert merged, f"{cfg}: the merged launch is not in use ({m.step_info(0)[0]} / {m.step_info(200)[0]} launches per step)"
    assert (m.generate_greedy(prompt, n_new) == ref).all(), f"{cfg}: default form"
    monkeypatch.setenv("LMRS_QKV_ATT", "1")                  # one workgroup per head from position 0
    m1 = L.Transformer(img)
    assert m1.step_info(0)[0] == 4 * nl + 1
    assert (m1.generate_greedy(prompt, n_new) == ref).all(), f"{cfg}: workgroup form"
    monkeypatch.setenv("LMRS_QKV_ATT", "0"); monkeypatch.setenv("LMRS_CLS_TAIL", "0")
    m0 = L.Transformer(img)
    assert m0.step_info(0)[0] == 5 * nl + 2
    assert (m0.generate_greedy(prompt, n_new) == ref).all(), f"{cfg}: separate launches"
    # positions re-run out of order on the default form (cache rows of the run above are in place): logits bit for bit
    toks = np.concatenate([prompt, ref]).astype(np.uint32)
    for pos in (3, 70, 3, 64, 63, 127, 128, 5, 140):
        assert_bit_equal(m.forward(int(toks[pos]), pos), orc.forward(int(toks[pos]), pos), f"{cfg} logits at re-run pos {pos}")
        assert m.forward_argmax(int(toks[pos]), pos) == int(toks[pos + 1]) or pos < len(prompt) - 1


def test_steps_enqueued_without_a_graph_give_the_same_tokens(L, monkeypatch):
    """LMRS_NO_GRAPH=1 (the profiling aid: rocprofv3 1.1 dies on the graph launches of most models) enqueues a step's launches one by one -
    same kernels, same order: token ids and logits as the CPU path's, across the wave -> workgroup switch of the merged launch."""
    img = S.build_image("mini-llama", S.Q8_0, seed=43)
    prompt = S.prompt_tokens("mini-llama", 6, 43)
    orc = O.Oracle(img)
    ref = orc.generate_greedy(prompt, 140)
    monkeypatch.setenv("LMRS_NO_GRAPH", "1")
    m = L.Transformer(img)
    assert (m.generate_greedy(prompt, 140) == ref).all()
    assert_bit_equal(m.forward(int(ref[-1]), 6 + 139), orc.forward(int(ref[-1]), 6 + 139), "logits, eager step")


def test_classifier_tags_survive_2047_layers_only_steps(L, monkeypatch):
    """The folded argmax tags its packed partials with 11 bits of a counter.  That counter used to be the step counter, which the
    layers-only steps of a token-by-token fill bump as well: exactly 2047 of them between two decode steps made the previous step's
    partials look current.  The tags now count classifier launches only: decode, 2047 fill steps (32 fills over the same positions),
    decode - the token and the logits of the second decode step are the CPU path's."""
    monkeypatch.setenv("LMRS_NO_BATCHED_PREFILL", "1")
    cfg = "mini-llama"
    img = S.build_image(cfg, S.Q8_0, seed=91)
    m = L.Transformer(img); orc = O.Oracle(img)
    t0 = 17
    first = m.forward_argmax(t0, 0)
    lo0 = orc.forward(t0, 0)
    assert first == int(O.lib().lmrs_ref_argmax(lo0.ctypes.data, lo0.size))
    toks = S.prompt_tokens(cfg, 64, 92)
    emb = m.get_embeddings(toks)
    done = 0
    for k in range(32):                                       # 31 x 64 + 63 = 2047 layers-only steps
        n = 64 if k < 31 else 63
        assert m.fill_kv_cache(emb[: n * m.args.dim].copy(), 1) == 1 + n
        done += n
    assert done == 2047
    e2 = orc.get_embeddings(toks[:63])
    assert orc.fill_kv_cache(e2, 1) == 64                     # (the fills rewrite the same rows with the same values: one suffices here)
    lo = orc.forward(int(first), 64)
    assert m.forward_argmax(int(first), 64) == int(O.lib().lmrs_ref_argmax(lo.ctypes.data, lo.size))
    assert_bit_equal(m.forward(int(first), 64), lo, "logits after the 2047 layers-only steps")


# ------------------------------------------------------------------ error behaviour (reference: panics)
def test_errors(L):
    img = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "tiny_llama_q8.lmrs"), np.uint8)
    bad = img.copy(); bad[0] = 0
    with pytest.raises(L.LmrsError, match="lm.rs format"):
        L.Transformer(bad)
    with pytest.raises(L.LmrsError, match="truncated"):
        L.Transformer(img[: img.size // 2])
    m = L.Transformer(img)
    with pytest.raises(L.LmrsError):
        m.forward(m.args.vocab_size, 0)
    with pytest.raises(L.LmrsError):
        m.forward(0, m.args.seq_len)
    with pytest.raises(L.LmrsError):
        m.generate_greedy(np.zeros(0, np.uint32), 4)
    # the peer-to-peer transport's per-peer tables hold 8 shards (one node): a larger world without a communicator id is refused at create
    # (tiny_llama's vocabulary of 256 rows divides by 16, so the "cls" plan itself would have accepted it)
    with pytest.raises(L.LmrsError, match="at most 8 shards"):
        L.Transformer(img, rank=1, world=16)
    # Phi with a head size above 96: the reference indexes past its 48 LongRoPE factors and panics (transformer.rs:473-475)
    phi128 = S.build_image(S.ModelCfg("phi-128", 128, 128, 1, 2, 128, 2, 256, 32, 1e-5, 10000.0, S.PHI), S.Q8_0, seed=3, threads=1)
    with pytest.raises(L.LmrsError, match="LongRoPE"):
        L.Transformer(phi128)
    with pytest.raises(Exception, match="LongRoPE"):
        O.Oracle(phi128)


# ------------------------------------------------------------------ CLIP image tower (BASELINE configs[4])
@pytest.mark.parametrize("n_layers,num_crops", [(2, 1), (3, 2)])
def test_vision_tower_matches_the_cpu_path(L, n_layers, num_crops):
    """VisionTransformer::forward (vision.rs:244-577) at the real CLIP ViT-L/14-336 geometry (the reference hard-codes 577
    positions), a few layers deep: patch embedding with the matmul_rest tail quirk, f32x8-structured layernorm and attention
    sums, Q8_0 projections as int8 matrix-core GEMMs with bias / QuickGELU / residual epilogues - bit-equal to the CPU path."""
    from tools import synth_vision as V
    cfg = V.VisionCfg(n_layers=n_layers)
    sec = V.build_vision_section(cfg, seed=5 + n_layers)
    dev = L.VisionTransformer(sec); orc = O.VisionOracle(sec)
    assert dev.bytes_consumed == orc.bytes_consumed == sec.size
    pv = V.pixel_values(cfg, num_crops, seed=3)
    got = dev.forward(pv, num_crops); ref = orc.forward(pv, num_crops)
    assert_bit_equal(got.reshape(-1), ref.reshape(-1), f"vision tower, {n_layers - 1} layer(s), {num_crops} crop(s)")


def test_vision_tower_with_the_last_query_in_a_block_of_its_own(L, monkeypatch):
    """577 = 9 x 64 + 1: query 576 is normally taken by workgroups of its own inside the softmax launch (vis_att_stray: threads = keys, then (lane sum, dim)
    pairs).  LMRS_VIS_NO_STRAY=1 runs it as a tenth block with one live lane through the blocked phases instead: both forms equal the CPU path bit for bit."""
    from tools import synth_vision as V
    cfg = V.VisionCfg(n_layers=2)
    sec = V.build_vision_section(cfg, seed=77)
    dev = L.VisionTransformer(sec); orc = O.VisionOracle(sec)
    pv = V.pixel_values(cfg, 2, seed=6)
    ref = orc.forward(pv, 2)
    assert_bit_equal(dev.forward(pv, 2).reshape(-1), ref.reshape(-1), "vision tower, stray query in its own workgroups")
    monkeypatch.setenv("LMRS_VIS_NO_STRAY", "1")                 # read once, when the tower is created
    dev2 = L.VisionTransformer(sec)
    assert_bit_equal(dev2.forward(pv, 2).reshape(-1), ref.reshape(-1), "vision tower, last query as a tenth block")


@pytest.mark.parametrize("q,n_layers,num_crops", [(S.Q4_0, 3, 2), (S.Q_NONE, 2, 1), (S.Q_NONE, 3, 2)])
def test_vision_tower_with_q4_and_unquantised_sections(L, q, n_layers, num_crops):
    """The other two section types export.py writes for the tower (vision.rs:110-243): Q4_0 - quantize_q4 of every row, matmul_q4 on
    the matrix cores from packed nibbles - and q_type None - the f32 `matmul` (chunk sums through wide's tree, added in order)."""
    from tools import synth_vision as V
    cfg = V.VisionCfg(n_layers=n_layers)
    sec = V.build_vision_section(cfg, seed=15 + n_layers, q_type=q)
    dev = L.VisionTransformer(sec); orc = O.VisionOracle(sec)
    assert dev.bytes_consumed == orc.bytes_consumed == sec.size
    pv = V.pixel_values(cfg, num_crops, seed=4)
    got = dev.forward(pv, num_crops); ref = orc.forward(pv, num_crops)
    assert_bit_equal(got.reshape(-1), ref.reshape(-1), f"vision tower q_type {q}, {n_layers - 1} layer(s), {num_crops} crop(s)")


def test_image_path_with_five_crops(L):
    """A 2 x 2 grid of sub-images + the global crop (num_crops = 5, the shape chat.rs produces for a large square image): tower (a few
    layers) -> projector, (2*12) * (2*12 + 1) + 12 * 13 + 1 = 757 embeddings, bit-equal to the CPU path."""
    from tools import synth_vision as V
    cfg = V.VisionCfg(n_layers=3)
    vsec = V.build_vision_section(cfg, seed=51); psec = V.build_processor_section(seed=52)
    vis = L.VisionTransformer(vsec); vis_o = O.VisionOracle(vsec)
    proc = L.PHI3VProcessor(psec); proc_o = O.ProcessorOracle(psec)
    pv = V.pixel_values(cfg, 5, seed=9)
    feats = vis.forward(pv, 5); feats_o = vis_o.forward(pv, 5)
    assert_bit_equal(feats.reshape(-1), feats_o.reshape(-1), "vision tower, 5 crops")
    got = proc.forward(feats, 576 * 1024, 12, 2, 2); ref = proc_o.forward(feats_o, 576 * 1024, 12, 2, 2)
    assert got.shape == ref.shape == (757, 3072)
    assert_bit_equal(got.reshape(-1), ref.reshape(-1), "image projector, 2 x 2 sub-images")


def test_contexts_release_their_device_memory(L):
    """Twenty rounds of create / run / destroy of a text model, a tower and a projector leave the device's free memory where it was
    (the arena, the prefill buffers, the split-attention scratch, the vision work buffers all go with their owner)."""
    import ctypes
    from tools import synth_vision as V
    hip = ctypes.CDLL("libamdhip64.so")
    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value
    img = S.build_image("mini-llama", S.Q8_0, seed=5)
    vsec = V.build_vision_section(V.VisionCfg(n_layers=2), seed=6); psec = V.build_processor_section(seed=7)
    toks = S.prompt_tokens("mini-llama", 40, 5)
    def one_round():
        m = L.Transformer(img)
        e = m.get_embeddings(toks); m.fill_kv_cache(e, 0); m.generate_greedy(toks[:4], 4, start_pos=40)
        v = L.VisionTransformer(vsec); p = L.PHI3VProcessor(psec)
        f = v.forward(V.pixel_values(V.VisionCfg(n_layers=2), 1, seed=1), 1)
        p.forward(np.concatenate([f, f]), 576 * 1024, 12, 1, 1)
        m.close(); v.close(); p.close()
    one_round(); one_round()                       # (the first rounds may grow runtime-internal pools)
    before = free_bytes()
    for _ in range(20):
        one_round()
    after = free_bytes()
    assert before - after < 64 << 20, f"device memory shrank by {(before - after) >> 20} MiB over 20 create / destroy rounds"


def test_vision_tower_at_full_depth(L):
    """The tower as configs[4] runs it: 24 layers in the file, 23 executed (vision.rs:303), global crop + one sub-image."""
    from tools import synth_vision as V
    cfg = V.VisionCfg(n_layers=24)
    sec = V.build_vision_section(cfg, seed=41)
    dev = L.VisionTransformer(sec); orc = O.VisionOracle(sec)
    pv = V.pixel_values(cfg, 2, seed=6)
    assert_bit_equal(dev.forward(pv, 2).reshape(-1), orc.forward(pv, 2).reshape(-1), "vision tower, 23 layers, 2 crops")


def _multimodal_file(text_cfg, vis_layers, seed, q=S.Q8_0):
    from tools import synth_vision as V
    text = S.build_image(text_cfg, q, seed=seed, multimodal=1)
    vcfg = V.VisionCfg(n_layers=vis_layers)
    return np.concatenate([text, V.build_vision_section(vcfg, seed=seed + 1, q_type=q), V.build_processor_section(seed=seed + 2, q_type=q)]), vcfg


def _image_prefill(model, vision, processor, pv, num_crops, w_crop, h_crop, bos_image, bos_text):
    """chat.rs:84-121: vision.forward -> processor.forward -> [prefix embeddings | image features | suffix embeddings] -> fill_kv_cache."""
    feats = vision.forward(pv, num_crops)
    img = processor.forward(feats, 576 * 1024, 336 // 14 // 2, w_crop, h_crop)
    pre = model.get_embeddings(np.asarray(bos_image, np.uint32)); suf = model.get_embeddings(np.asarray(bos_text, np.uint32))
    emb = np.ascontiguousarray(np.concatenate([pre.reshape(-1), img.reshape(-1), suf.reshape(-1)]), np.float32)
    return emb, model.fill_kv_cache(emb, 0), img


@pytest.mark.parametrize("text_cfg,vis_layers,q", [("mini-phi-long", 3, S.Q8_0), ("phi-3.5", 24, S.Q8_0), ("mini-phi-long", 3, S.Q4_0), ("mini-phi-long", 2, S.Q_NONE)])
def test_multimodal_prefill_end_to_end(L, text_cfg, vis_layers, q):
    """BASELINE.json configs[4] in the reference's call order (src/bin/chat.rs:84-121): Transformer::new, VisionTransformer::new at
    the offset it returned, PHI3VProcessor::new after the vision section; vision.forward -> processor.forward -> the 313 image
    features spliced between two get_embeddings blocks (4 + 313 + 3 = 320 embeddings) -> fill_kv_cache(320) -> 8 greedy decode
    steps on the prefilled cache.  Every stage bit-equal to the CPU path: image features, the mutated embeddings, the token ids,
    the logits of one more step and KV rows inside the image span.  ("phi-3.5", 24): everything at full size; Q4_0 / q_type None:
    the whole file - text, tower, projector - in the exporter's other two formats."""
    from tools import synth_vision as V
    data, vcfg = _multimodal_file(text_cfg, vis_layers, seed=31, q=q)
    m = L.Transformer(data); orc = O.Oracle(data)
    assert m.args.multimodal == 1 and m.bytes_consumed == orc.bytes_consumed
    off = m.bytes_consumed
    vis = L.VisionTransformer(data[off:]); vis_o = O.VisionOracle(data[off:])
    proc = L.PHI3VProcessor(data[off + vis.bytes_consumed:]); proc_o = O.ProcessorOracle(data[off + vis_o.bytes_consumed:])
    assert off + vis.bytes_consumed + proc.bytes_consumed == data.size
    pv = V.pixel_values(vcfg, 2, seed=8)
    V_ = m.args.vocab_size
    bos_image = [1, 32010 % V_, 29871 % V_, 13]; bos_text = [1, 29871 % V_, 13]          # chat.rs:110-112 (ids folded into the mini vocabulary)
    e_dev, p_dev, f_dev = _image_prefill(m, vis, proc, pv, 2, 1, 1, bos_image, bos_text)
    e_ref, p_ref, f_ref = _image_prefill(orc, vis_o, proc_o, pv, 2, 1, 1, bos_image, bos_text)
    assert f_dev.shape == (313, 3072) and p_dev == p_ref == 320
    assert_bit_equal(f_dev.reshape(-1), f_ref.reshape(-1), "image features")
    assert_bit_equal(e_dev, e_ref, "embeddings after fill_kv_cache")
    prompt = S.prompt_tokens(text_cfg, 5, 77)
    got = m.generate_greedy(prompt, 8, start_pos=320); ref = orc.generate_greedy(prompt, 8, start_pos=320)
    assert (got == ref).all(), (got, ref)
    pos = 320 + 5 + 7
    assert_bit_equal(m.forward(int(ref[-1]), pos), orc.forward(int(ref[-1]), pos), f"logits at pos {pos}")
    for l in (0, m.args.n_layers - 1):
        for p in (3, 160, 319, pos):
            for which in (0, 1):
                assert_bit_equal(m.kv_row(which, l, p), orc.kv_row(which, l, p), f"kv[{which}] layer {l} pos {p}")


def test_image_prefill_host_program_runs_on_the_gpu(L, tmp_path):
    """hostcpp/image_prefill.cpp - the same call order through the C++ mirrors of Transformer / VisionTransformer / PHI3VProcessor -
    built with g++ against the shared library and RUN: its token ids equal the Python path's (and therefore the CPU path's)."""
    import subprocess
    from tools import synth_vision as V
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "lm.rs_amd", "hostcpp")
    exe = str(tmp_path / "image_prefill")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(host, "image_prefill.cpp"), "-I", os.path.join(root, "include"),
                    "-L", os.path.join(root, "lm.rs_amd"), "-llmrs_hip", f"-Wl,-rpath,{os.path.join(root, 'lm.rs_amd')}", "-o", exe], check=True)
    # the program feeds the reference's literal prefix / suffix ids (chat.rs:110-112): needs the real vocabulary size, few layers
    cfg = S.ModelCfg("phi-2layer", 3072, 8192, 2, 32, 96, 32, 32064, 131072, 1e-5, 10000.0, S.PHI)
    text = S.build_image(cfg, S.Q8_0, seed=57, multimodal=1)
    vcfg = V.VisionCfg(n_layers=3)
    data = np.concatenate([text, V.build_vision_section(vcfg, seed=58), V.build_processor_section(seed=59)])
    pv = V.pixel_values(vcfg, 2, seed=9)
    data.tofile(tmp_path / "model.lmrs"); np.ascontiguousarray(pv, np.float32).tofile(tmp_path / "patches.f32")
    prompt = [int(t) for t in S.prompt_tokens(cfg, 5, 12)]
    out = subprocess.run([exe, str(tmp_path / "model.lmrs"), str(tmp_path / "patches.f32"), "2", "1", "1", "8"] + [str(t) for t in prompt],
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert "313 embeddings" in out[0] and "position 320" in out[0], out
    ids_cpp = np.array([int(t) for t in out[1].split()], np.uint32)
    m = L.Transformer(data); off = m.bytes_consumed
    vis = L.VisionTransformer(data[off:]); proc = L.PHI3VProcessor(data[off + vis.bytes_consumed:])
    _image_prefill(m, vis, proc, pv, 2, 1, 1, [1, 32010, 29871, 13], [1, 29871, 13])
    ids_py = m.generate_greedy(np.asarray(prompt, np.uint32), 8, start_pos=320)
    assert (ids_cpp == ids_py).all(), (ids_cpp, ids_py)
    orc = O.Oracle(data); vo = O.VisionOracle(data[off:]); po = O.ProcessorOracle(data[off + vo.bytes_consumed:])
    _image_prefill(orc, vo, po, pv, 2, 1, 1, [1, 32010, 29871, 13], [1, 29871, 13])
    assert (orc.generate_greedy(np.asarray(prompt, np.uint32), 8, start_pos=320) == ids_cpp).all()


@pytest.mark.parametrize("w_crop,h_crop", [(1, 1), (2, 1)])
def test_image_projector_matches_the_cpu_path(L, w_crop, h_crop):
    """PHI3VProcessor::forward (processor.rs:234-342): the 2x2 HD merge, the row separators (sub_GN), glb_GN between the
    sub-image and global features, then quantise -> proj0 + bias -> tanh-GELU (f64 tanh) -> quantise -> proj1 + bias for every
    embedding, as two int8 matrix-core GEMMs - bit-equal to the CPU path."""
    from tools import synth_vision as V
    sec = V.build_processor_section(seed=11)
    dev = L.PHI3VProcessor(sec); orc = O.ProcessorOracle(sec)
    assert dev.bytes_consumed == orc.bytes_consumed == sec.size
    n_crops = 1 + w_crop * h_crop
    rng = np.random.default_rng(17 + w_crop)
    feats = (rng.standard_normal((n_crops, 576, 1024)) * 1.5).astype(np.float32)
    got = dev.forward(feats, 576 * 1024, 12, w_crop, h_crop); ref = orc.forward(feats, 576 * 1024, 12, w_crop, h_crop)
    assert got.shape == ref.shape == ((h_crop * 12) * (w_crop * 12 + 1) + 12 * 13 + 1, 3072)
    assert_bit_equal(got.reshape(-1), ref.reshape(-1), f"image projector, {w_crop}x{h_crop} sub-images")


@pytest.mark.parametrize("q,text_dim", [(S.Q4_0, 3072), (S.Q_NONE, 3072), (S.Q4_0, 2048)])
def test_image_projector_with_q4_and_unquantised_sections(L, q, text_dim):
    from tools import synth_vision as V
    sec = V.build_processor_section(text_dim=text_dim, seed=12, q_type=q)
    dev = L.PHI3VProcessor(sec); orc = O.ProcessorOracle(sec)
    assert dev.bytes_consumed == orc.bytes_consumed == sec.size
    feats = (np.random.default_rng(19).standard_normal((3, 576, 1024)) * 1.5).astype(np.float32)
    got = dev.forward(feats, 576 * 1024, 12, 2, 1); ref = orc.forward(feats, 576 * 1024, 12, 2, 1)
    assert got.shape == ref.shape
    assert_bit_equal(got.reshape(-1), ref.reshape(-1), f"image projector q_type {q}, text_dim {text_dim}")


def test_image_projector_rejects_bad_geometry(L):
    from tools import synth_vision as V
    sec = V.build_processor_section(seed=11)
    dev = L.PHI3VProcessor(sec)
    with pytest.raises(L.LmrsError):
        dev.forward(np.zeros((2, 576, 1024), np.float32), 576 * 1024, 12, 2, 1)      # 3 crops promised, 2 given
    with pytest.raises(L.LmrsError):
        L.PHI3VProcessor(sec[: sec.size // 2])


# ------------------------------------------------------------------ the chat-compatible harness, text in / text out
def _chat_fixture(tmp_path):
    import struct
    cfg = S.ModelCfg("llama-2layer", 2048, 8192, 2, 32, 64, 8, 128256, 131072, 1e-5, 500000.0, S.LLAMA)
    img = S.build_image(cfg, S.Q8_0, seed=91)
    img.tofile(tmp_path / "model.lmrs")
    # a tokenizer.bin in the layout tokenizer.rs:24-64 reads, full Llama vocabulary size (the chat template ids must exist)
    toks = [("<unk>", 0.0), ("<s>", 0.0), ("</s>", 0.0)] + [("<0x%02X>" % b, 0.0) for b in range(256)]
    toks += [(ch, -1.0 - i) for i, ch in enumerate(" abcdefghijklmnopqrstuvwxyzSO0123456789")]
    toks += [(m, 5.0 - 0.1 * i) for i, m in enumerate(["he", "ll", "hell", "hello", " w", "or", "ld", " world", "wor", "Se", "ep", "Sep", "20", "24", "2024", "23"])]
    toks += [("<fill_%d>" % i, 0.0) for i in range(cfg.vocab_size - len(toks))]
    blob = struct.pack("IIII", len(toks), 16, 128000, 128009)
    for s_, sc in toks:
        b = s_.encode(); blob += struct.pack("fI", sc, len(b)) + b
    (tmp_path / "tokenizer.bin").write_bytes(blob)
    return cfg, img, blob


@pytest.mark.parametrize("temperature", [0.0, 0.7])
def test_chat_program_text_in_text_out(L, tmp_path, temperature):
    """hostcpp/chat.cpp - the reference's chat loop (src/bin/chat.rs:148-227) over the C++ mirrors of Tokenizer, Transformer and
    Sampler - built and RUN on the GPU with a text prompt: what it prints is what the CPU path (the oracle's forward, the second
    transcription of tokenizer and sampler) produces for the same model, tokenizer, date, temperature, top-p and seed."""
    import subprocess
    import text_ref as R
    cfg, img, blob = _chat_fixture(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "chat")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(root, "lm.rs_amd", "hostcpp", "chat.cpp"), "-I", os.path.join(root, "include"),
                    "-L", os.path.join(root, "lm.rs_amd"), "-llmrs_hip", f"-Wl,-rpath,{os.path.join(root, 'lm.rs_amd')}", "-o", exe], check=True)
    n_new, seed, top_p = 6, 4242, 0.9
    out = subprocess.run([exe, "--model", str(tmp_path / "model.lmrs"), "--tokenizer", str(tmp_path / "tokenizer.bin"), "--temperature", str(temperature),
                          "--top-p", str(top_p), "--seed", str(seed), "--date", "23 Sep 2024", "--max-tokens", str(n_new)],
                         input="  hello world  \n", capture_output=True, text=True, check=True).stdout
    assert out.startswith("You: Assistant:\n"), out
    printed = out[len("You: Assistant:\n"):]
    # the CPU path: second transcription of the tokenizer, the oracle's forward, second transcription of the sampler
    tk = R.Tokenizer(blob)
    prompt = [128000, 128006, 9125, 128007, 271, 38766, 1303, 33025, 2696, 25, 6790, 220, 2366, 18, 198, 15724, 2696, 25, 220]
    prompt += tk.encode("23 Sep 2024", False, False, False, 1) + [271, 128009] + tk.encode("hello world", False, False, True, 1)
    assert (L.Tokenizer(blob).encode("hello world", False, False, True, 1) == np.array(tk.encode("hello world", False, False, True, 1), np.uint32)).all()
    orc = O.Oracle(img); smp = R.Sampler(cfg.vocab_size, temperature, top_p, seed)
    pos, nxt, pieces = 0, 0, []
    for i in range(len(prompt) + n_new - 1):
        token = prompt[i] if i < len(prompt) else nxt
        lg = orc.forward(int(token), pos); pos += 1
        if temperature == 0.0:
            nxt = int(O.lib().lmrs_ref_argmax(lg.ctypes.data, lg.size))
        else:                                                   # every call sorts the candidate vector (stale entries stay): run them all, as chat.rs does
            nxt = smp.sample([np.float32(v) for v in lg])
        if i >= len(prompt) - 1 and nxt != tk.eos:
            pieces.append(tk.decode(nxt))
    assert printed == "".join(pieces) + "\n", (printed, pieces)
