"""Independent numpy transcription of the reference's forward pass (TEST INFRASTRUCTURE).

Written from the Rust sources (reference src/transformer.rs:316-657, src/functional.rs:48-250,
src/quantization.rs:25-95) separately from oracle/lmrs_oracle.c, in a different style (vectorised
numpy with explicit float32 rounding points), to catch transcription slips in the C oracle.  Only
small models: it is slow.  exp/cos/sin/pow go through the C library (ctypes libm), as Rust's do.
"""
import ctypes
import ctypes.util
import math
import struct

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m"))
for _n in ("expf", "cosf", "sinf", "logf", "sqrtf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.tanh.restype = ctypes.c_double
_libm.tanh.argtypes = [ctypes.c_double]


def expf(x):
    return F(_libm.expf(float(x)))


def reduce_add8(v):          # wide f32x8::reduce_add, AVX path
    return F(F(F(v[0] + v[4]) + F(v[2] + v[6])) + F(F(v[1] + v[5]) + F(v[3] + v[7])))


def rmsnorm(x, w, eps, add_unit):
    n = x.size
    acc = np.zeros(8, F)
    xs = x.reshape(-1, 8)
    for j in range(n // 8):
        acc = (acc + (xs[j] * xs[j]).astype(F)).astype(F)
    ss = reduce_add8(acc)
    ss = F(ss / F(n)); ss = F(ss + F(eps)); ss = F(F(1.0) / F(_libm.sqrtf(float(ss))))
    t = (ss * x).astype(F)
    return ((F(1.0) + w).astype(F) * t).astype(F) if add_unit else (w * t).astype(F)


def round_half_away(v):
    return np.where(v >= 0, np.floor(v + F(0.5)), np.ceil(v - F(0.5))).astype(F)


def quantize_q8(x, gs=128):
    g = x.reshape(-1, gs)
    wmax = np.abs(g).max(axis=1).astype(F)
    scale = (wmax / F(127.0)).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (g / scale[:, None]).astype(F)
    # f32::round is exact on the f32 value; v+0.5 in f32 could double-round, so do it in float64
    r = np.where(q >= 0, np.floor(q.astype(np.float64) + 0.5), np.ceil(q.astype(np.float64) - 0.5))
    r = np.nan_to_num(r, nan=0.0)
    return np.clip(r, -128, 127).astype(np.int8).reshape(-1), scale


def quantize_q4(x, gs=128):
    g = x.reshape(-1, gs)
    wmax = np.abs(g).max(axis=1).astype(F)
    scale = (wmax / F(-8.0)).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = ((g / scale[:, None]).astype(F) + F(8.0)).astype(F)
    r = np.where(q >= 0, np.floor(q.astype(np.float64) + 0.5), np.ceil(q.astype(np.float64) - 0.5))
    r = np.clip(np.nan_to_num(r, nan=0.0), 0, 15).astype(np.uint8)
    r = r.reshape(-1, 2)
    return (r[:, 0] | (r[:, 1] << 4)).astype(np.uint8), scale


def unpack_q4(b):
    b = b.astype(np.int32)
    lo = (b & 0x0F) - 8
    hi = ((b >> 4) & 0x0F) - 8
    return np.stack([lo, hi], axis=-1).reshape(*b.shape[:-1], -1)


def matmul_q(xq, xs, wq, ws, n, o, gs, q4):
    """matmul_q8 / matmul_q4 (decode form): per row, groups ascending, ((ival as f32) * ws) * xs."""
    if q4:
        W = unpack_q4(wq.reshape(o, n // 2)); X = unpack_q4(xq.reshape(1, n // 2))[0]
    else:
        W = wq.reshape(o, n).astype(np.int32); X = xq.astype(np.int32)
    G = n // gs
    isum = (W.reshape(o, G, gs) * X.reshape(1, G, gs)).sum(axis=2).astype(np.int32)
    WS = ws.reshape(o, G)
    out = np.zeros(o, F)
    for g in range(G):
        p = (isum[:, g].astype(F) * WS[:, g]).astype(F)
        p = (p * xs[g]).astype(F)
        out = (out + p).astype(F)
    if not q4:
        out[(o // 4) * 4:] = 0
    return out


def matmul_f32(x, w, n, o):
    """matmul (functional.rs:142-171), one token: per row, chunks of 8 in order, xout += (x_vec * w_vec).reduce_add() with wide's AVX
    tree ((p0+p4)+(p2+p6)) + ((p1+p5)+(p3+p7)); rows in groups of 4 (par_chunks_exact_mut(4): a shorter last group is never written)."""
    nc = n // 8
    P = (w.reshape(o, n)[:, :nc * 8].reshape(o, nc, 8) * x[:nc * 8].reshape(1, nc, 8)).astype(F)
    q0 = (P[..., 0] + P[..., 4]).astype(F); q1 = (P[..., 1] + P[..., 5]).astype(F)
    q2 = (P[..., 2] + P[..., 6]).astype(F); q3 = (P[..., 3] + P[..., 7]).astype(F)
    S = ((q0 + q2).astype(F) + (q1 + q3).astype(F)).astype(F)                    # [o, nc] chunk sums
    out = np.zeros(o, F)
    for j in range(nc):
        out = (out + S[:, j]).astype(F)
    out[(o // 4) * 4:] = 0
    return out


PHI_SHORT = [1.08, 1.1, 1.1300000000000001, 1.2800000000000002, 1.3100000000000003, 1.4500000000000004, 1.4500000000000004, 1.9500000000000008,
             2.030000000000001, 2.4299999999999926, 2.5699999999999896, 2.9499999999999815, 3.729999999999965, 3.869999999999962, 4.189999999999955,
             4.43999999999995, 4.6399999999999455, 4.979999999999938, 5.159999999999934, 5.279999999999932, 5.759999999999922, 5.889999999999919,
             5.889999999999919, 5.969999999999917, 6.089999999999915, 6.2799999999999105, 6.7699999999999, 6.8899999999998975, 7.109999999999893,
             7.129999999999892, 7.179999999999891, 7.289999999999889, 7.339999999999888, 7.559999999999883, 7.619999999999882, 7.69999999999988,
             7.879999999999876, 7.879999999999876, 7.879999999999876, 7.939999999999875, 7.949999999999875, 7.979999999999874, 8.19999999999987,
             8.439999999999864, 8.469999999999864, 8.589999999999861, 8.809999999999857, 8.999999999999853]


class NumpyModel:
    def __init__(self, image: np.ndarray):
        d = image.tobytes()
        assert d[:4] == b"lmrs"
        (self.dim, self.hidden, self.L, self.n_heads, self.hs, self.n_kv, self.vocab, self.seq_len, self.eps, self.theta) = struct.unpack("IIIIIIIIff", d[8:48])
        self.q_type, self.model_type = d[48], d[49]
        self.gs = struct.unpack("I", d[50:54])[0]
        self.seq_len = min(self.seq_len, 8192)
        self.att = self.n_heads * self.hs; self.kv = self.n_kv * self.hs
        off = [256]
        img = image

        def f32(cnt):
            a = img[off[0]:off[0] + cnt * 4].view(F); off[0] += cnt * 4; return a

        def quant(n_t, each):
            out = []
            if self.q_type == 0:                                   # q_type None: plain f32 tensors, no scales (init_param, transformer.rs:16-22)
                return [(f32(each), None) for _ in range(n_t)]
            for _ in range(n_t):
                qb = each // 2 if self.q_type == 2 else each
                q = img[off[0]:off[0] + qb]; off[0] += qb
                s = img[off[0]:off[0] + each // self.gs * 4].view(F); off[0] += each // self.gs * 4
                out.append((q.view(np.uint8) if self.q_type == 2 else q.view(np.int8), s))
            return out
        dim, L, att, kv, hid, V = self.dim, self.L, self.att, self.kv, self.hidden, self.vocab
        gem = self.model_type == 0
        self.emb = quant(1, V * dim)[0]
        self.rms_att = f32(L * dim).reshape(L, dim)
        self.wq = quant(L, dim * att); self.wk = quant(L, dim * kv); self.wv = quant(L, dim * kv); self.wo = quant(L, dim * att)
        self.rms_post = f32(L * dim).reshape(L, dim)
        if gem: self.rms_pre_ffn = f32(L * dim).reshape(L, dim)
        self.w1 = quant(L, dim * hid); self.w2 = quant(L, dim * hid); self.w3 = quant(L, dim * hid)
        if gem: self.rms_post_ffn = f32(L * dim).reshape(L, dim)
        self.rms_final = f32(dim)
        self.lm_head = quant(1, dim * V)[0] if self.model_type == 2 else self.emb
        self.end = off[0]
        self.kc = np.zeros((L, self.seq_len, kv), F); self.vc = np.zeros((L, self.seq_len, kv), F)

    def _q(self, x):
        if self.q_type == 0:
            return x, None                                         # unquantised: the activation goes into matmul as it is
        return quantize_q4(x, self.gs) if self.q_type == 2 else quantize_q8(x, self.gs)

    def _mm(self, xq, xs, w, n, o):
        if self.q_type == 0:
            return matmul_f32(xq, w[0], n, o)
        return matmul_q(xq, xs, w[0], w[1], n, o, self.gs, self.q_type == 2)

    def embed(self, token):
        q, s = self.emb
        if self.q_type == 0:
            return q[token * self.dim:(token + 1) * self.dim].copy()
        if self.q_type == 2:
            vals = unpack_q4(q[token * self.dim // 2:(token + 1) * self.dim // 2].reshape(1, -1))[0].astype(F)
        else:
            vals = q[token * self.dim:(token + 1) * self.dim].astype(F)
        idx = (token * self.dim + np.arange(self.dim)) // self.gs
        return (vals * s[idx]).astype(F)

    def rope(self, pos, j):
        freq = F(F(1.0) / F(_libm.powf(float(F(self.theta)), float(F(F(2 * j) / F(self.hs))))))
        sf = F(1.0)
        if self.model_type == 1:
            wavelen = F(F(F(2.0) * F(math.pi)) / freq)
            factor, lo, hi, old = F(32.0), F(1.0), F(4.0), F(8192.0)
            if wavelen > F(old / lo):
                freq = F(freq / factor)
            elif F(old / hi) <= wavelen <= F(old / lo):
                sm = F(F(F(old / wavelen) - lo) / F(hi - lo))
                freq = F(F(F(F(F(1.0) - sm) * freq) / factor) + F(sm * freq))
        if self.model_type == 2:
            freq = F(freq * F(1.0 / PHI_SHORT[j]))
            sf = F(_libm.sqrtf(float(F(F(1.0) + F(F(_libm.logf(32.0)) / F(_libm.logf(4096.0)))))))
        val = F(F(pos) * freq)
        return F(F(_libm.cosf(float(val))) * sf), F(F(_libm.sinf(float(val))) * sf)

    def layer(self, x, l, pos, wpos=None):
        """forward_layer for one token at `pos`; wpos: the `pos` argument of the forward_layer CALL, which the Gemma window test uses
        for every token of a batch (transformer.rs:525, u32 arithmetic) - None: a single-token call."""
        wpos = pos if wpos is None else wpos
        gem = self.model_type == 0
        dim, hs, att, kv = self.dim, self.hs, self.att, self.kv
        xn = rmsnorm(x, self.rms_att[l], self.eps, gem)
        xq, xs = self._q(xn)
        q = self._mm(xq, xs, self.wq[l], dim, att)
        k = self._mm(xq, xs, self.wk[l], dim, kv)
        v = self._mm(xq, xs, self.wv[l], dim, kv)
        half = hs // 2
        for i in range(self.n_heads):
            for j in range(half):
                c, s = self.rope(pos, j)
                for vec, ok in ((q, True), (k, i * hs + j + half < kv)):
                    if ok:
                        v0, v1 = vec[i * hs + j], vec[i * hs + j + half]
                        vec[i * hs + j] = F(F(v0 * c) - F(v1 * s))
                        vec[i * hs + j + half] = F(F(v0 * s) + F(v1 * c))
        self.kc[l, pos] = k; self.vc[l, pos] = v
        out = np.zeros(att, F)
        kvm = self.n_heads // self.n_kv
        for h in range(self.n_heads):
            qh = q[h * hs:(h + 1) * hs]
            sc = np.zeros(pos + 1, F)
            for t in range(pos + 1):
                kk = self.kc[l, t, (h // kvm) * hs:(h // kvm + 1) * hs]
                s_ = F(0.0)
                for d_ in range(hs):
                    s_ = F(s_ + F(qh[d_] * kk[d_]))
                s_ = F(s_ / F(_libm.sqrtf(float(hs))))
                if gem:
                    s_ = F(s_ / F(50.0)); s_ = F(_libm.tanh(float(s_))); s_ = F(s_ * F(50.0))
                    s_ = F(s_ + (F(0.0) if ((wpos - t) & 0xFFFFFFFF) <= 4096 else F(-2.3819763e38)))
                sc[t] = s_
            mx = sc.max()
            sm = F(0.0)
            for t in range(pos + 1):
                sc[t] = expf(F(sc[t] - mx)); sm = F(sm + sc[t])
            sc = (sc / sm).astype(F)
            o = np.zeros(hs, F)
            for t in range(pos + 1):
                o = (o + (sc[t] * self.vc[l, t, (h // kvm) * hs:(h // kvm + 1) * hs]).astype(F)).astype(F)
            out[h * hs:(h + 1) * hs] = o
        aq, as_ = self._q(out)
        wo = self._mm(aq, as_, self.wo[l], att, dim)
        if gem:
            x = (x + rmsnorm(wo, self.rms_post[l], self.eps, True)).astype(F)
            e = rmsnorm(x, self.rms_pre_ffn[l], self.eps, True)
        else:
            x = (x + wo).astype(F)
            e = rmsnorm(x, self.rms_post[l], self.eps, False)
        eq, es = self._q(e)
        g = self._mm(eq, es, self.w1[l], dim, self.hidden)
        u = self._mm(eq, es, self.w3[l], dim, self.hidden)
        for i in range(self.hidden):
            val = g[i]
            if gem:
                cube = F(F(F(F(0.044715) * val) * val) * val)
                th = _libm.tanh(0.7978845608028654 * float(F(val + cube)))
                val = F(val * F(F(0.5) * F(F(1.0) + F(th))))
            else:
                val = F(val * F(F(1.0) / F(F(1.0) + expf(F(-val)))))
            g[i] = F(val * u[i])
        hq, hs_ = self._q(g)
        ff = self._mm(hq, hs_, self.w2[l], self.hidden, dim)
        if gem:
            return (x + rmsnorm(ff, self.rms_post_ffn[l], self.eps, True)).astype(F)
        return (x + ff).astype(F)

    def fill_kv_cache(self, embeddings, curr_pos):
        """Transformer::fill_kv_cache (transformer.rs:672-684): forward_layer over all n tokens for each layer in turn (token i sits
        at curr_pos + i and attends to keys 0 .. curr_pos + i, :507,:533); mutates `embeddings` [n, dim] in place, returns
        curr_pos + n.  Per token the arithmetic is that of a single-token call (quantisation groups never straddle tokens),
        except Gemma's window test, which uses the call's `pos`."""
        n = embeddings.shape[0]
        for l in range(self.L):
            for i in range(n):
                embeddings[i] = self.layer(embeddings[i].copy(), l, curr_pos + i, wpos=curr_pos)
        return curr_pos + n

    def forward(self, token, pos):
        x = self.embed(token)
        if self.model_type == 0:
            x = (x * F(_libm.sqrtf(float(self.dim)))).astype(F)
        for l in range(self.L):
            x = self.layer(x, l, pos)
        x = rmsnorm(x, self.rms_final, self.eps, self.model_type == 0)
        xq, xs = self._q(x)
        logits = self._mm(xq, xs, self.lm_head, self.dim, self.vocab)
        if self.model_type == 0:
            for d_ in range(self.dim):
                v = F(logits[d_] / F(30.0)); v = F(_libm.tanh(float(v))); logits[d_] = F(v * F(30.0))
        return logits


# ------------------------------------------------------------------ image projector (reference src/processor.rs:234-342)
def hd_transform(feats, h_crop, w_crop, sep):
    """reshape_hd_patches_2x2merge (:377-418) + add_image_newline (:480-484), as one reshape/transpose: [n, 576, C] ->
    [h_crop*12 * (w_crop*12 + 1), 4C]."""
    n, L, Cc = feats.shape
    H = int(math.isqrt(L)); ni = n // (h_crop * w_crop)
    assert ni == 1
    t = feats.reshape(ni, h_crop, w_crop, H // 2, 2, H // 2, 2, Cc)           # img, hc, wc, i, di, j, dj, C
    t = t.transpose(0, 1, 3, 2, 5, 4, 6, 7).reshape(ni * h_crop * (H // 2), w_crop * (H // 2), 4 * Cc)
    nl = np.broadcast_to(sep.reshape(1, 1, -1), (t.shape[0], 1, 4 * Cc))
    return np.concatenate([t, nl], axis=1).reshape(-1, 4 * Cc)


def gelu_tanh(h):
    cube = (F(0.044715) * h).astype(F); cube = (cube * h).astype(F); cube = (cube * h).astype(F)
    inner = (h + cube).astype(F).astype(np.float64) * 0.7978845608028654
    th = np.array([_libm.tanh(float(v)) for v in inner], np.float64).astype(F)
    return (h * (F(0.5) * (F(1.0) + th).astype(F)).astype(F)).astype(F)


def processor_forward(section, feats, w_crop, h_crop, rows=None):
    hidden, text = struct.unpack_from("II", section, 0)
    qt = section[8]
    gs = struct.unpack_from("I", section, 9)[0]
    off = 128
    def take(dt, cnt):
        nonlocal off
        a = np.frombuffer(section, dt, cnt, off); off += a.nbytes
        return a
    def take_w(cnt):
        if qt == 0: return take(F, cnt), None
        if qt == 2: return take(np.uint8, cnt // 2), take(F, cnt // gs)
        return take(np.int8, cnt), take(F, cnt // gs)
    glb, sub = take(F, hidden), take(F, hidden)
    p0, s0 = take_w(text * hidden)
    p1, s1 = take_w(text * text)
    b0, b1 = take(F, text), take(F, text)
    emb = np.concatenate([hd_transform(feats[1:], h_crop, w_crop, sub), glb.reshape(1, -1), hd_transform(feats[:1], 1, 1, sub)])
    pick = range(emb.shape[0]) if rows is None else rows
    out = {}
    def mm(x, w, ws, n, o):
        if qt == 0: return matmul_f32(x, w, n, o)
        q, sc = quantize_q4(x, gs) if qt == 2 else quantize_q8(x, gs)
        return matmul_q(q, sc, w, ws, n, o, gs, qt == 2)
    for r in pick:
        h = (mm(emb[r], p0, s0, hidden, text) + b0).astype(F)
        h = gelu_tanh(h)
        out[r] = (mm(h, p1, s1, text, text) + b1).astype(F)
    return emb.shape[0], out


# ------------------------------------------------------------------ CLIP tower (reference src/vision.rs:99-577): Q8_0, Q4_0 or f32 sections
def _expf_arr(a):
    f = _libm.expf
    return np.array([f(float(v)) for v in a.reshape(-1)], F).reshape(a.shape)


def lane_dot(xa, wa):
    """matmul_rest / matmul inner loop (functional.rs:252-280) for n % 8 == 0: xa [..., n] . wa [..., n] with the f32x8
    accumulator (8 lane sums over chunks ascending) and wide's reduce_add; broadcasts over leading axes."""
    n = xa.shape[-1]
    acc = None
    for j in range(n // 8):
        p = (wa[..., j * 8:j * 8 + 8] * xa[..., j * 8:j * 8 + 8]).astype(F)
        acc = p if acc is None else (acc + p).astype(F)          # 0 + p == p exactly
    a = [acc[..., i] for i in range(8)]
    return ((a[0] + a[4]).astype(F) + (a[2] + a[6]).astype(F)).astype(F) + ((a[1] + a[5]).astype(F) + (a[3] + a[7]).astype(F)).astype(F)


def layernorm_rows(x, w, b, eps):
    """functional.rs:80-114 for every row of x [T, n]."""
    T, n = x.shape
    xs = x.reshape(T, n // 8, 8)
    m = np.zeros((T, 8), F)
    for j in range(n // 8):
        m = (m + xs[:, j]).astype(F)
    mean = (reduce_add8(m.T) / F(n)).astype(F)
    v = np.zeros((T, 8), F)
    for j in range(n // 8):
        d = (xs[:, j] - mean[:, None]).astype(F)
        v = (v + (d * d).astype(F)).astype(F)
    var = ((reduce_add8(v.T) / F(n)).astype(F) + F(eps)).astype(F)
    inv = (F(1.0) / np.sqrt(var).astype(F)).astype(F)
    nrm = ((x - mean[:, None]).astype(F) * inv[:, None]).astype(F)
    return ((nrm * w[None, :]).astype(F) + b[None, :]).astype(F)


def quant_matmul_rows(x, wq, ws, n, o, gs, qt=1):
    """The section's quantiser (quantization.rs:44-95) + matmul_q8 / matmul_q4, or the plain matmul (functional.rs:142-250), for
    every row of x [T, n] -> [T, o].  qt: 1 Q8_0, 2 Q4_0 (packed nibbles on both sides), 0 unquantised f32."""
    T = x.shape[0]
    if qt == 0:
        w = wq.reshape(o, n)
        out = np.zeros((T, o), F)
        for j in range(n // 8):                                        # chunk sums through wide's tree, added to the row in order
            pr = (x[:, None, j * 8:j * 8 + 8] * w[None, :, j * 8:j * 8 + 8]).astype(F)
            out = (out + reduce_add8(np.moveaxis(pr, -1, 0))).astype(F)
        return out
    G = n // gs
    sc = np.empty((T, G), F)
    if qt == 2:
        q = np.empty((T, n), np.int32)
        for t in range(T):
            pk, sc[t] = quantize_q4(x[t], gs)
            q[t] = unpack_q4(pk.reshape(1, n // 2))[0]
        wq = unpack_q4(wq.reshape(o, n // 2))
    else:
        q = np.empty((T, n), np.int8)
        for t in range(T):
            q[t], sc[t] = quantize_q8(x[t], gs)
    Wf = wq.reshape(o, G, gs).astype(np.float64); Xf = q.reshape(T, G, gs).astype(np.float64)
    WS = ws.reshape(o, G)
    out = np.zeros((T, o), F)
    for g in range(G):
        isum = (Xf[:, g] @ Wf[:, g].T)                             # exact: |sum| < 2^53
        pgrp = (isum.astype(F) * WS[None, :, g]).astype(F)
        pgrp = (pgrp * sc[:, g:g + 1]).astype(F)
        out = (out + pgrp).astype(F)
    return out


def vision_forward(section, pixel_values):
    """VisionTransformer::forward for ONE crop.  pixel_values [576, 588] (patch-major, as process() lays them out)."""
    dim, hid, n_layers, n_heads, hs = struct.unpack_from("5I", section, 0)
    eps = struct.unpack_from("f", section, 20)[0]
    patch, image = struct.unpack_from("2I", section, 24)
    qt = section[32]
    gs = struct.unpack_from("I", section, 33)[0]
    off = 128
    def take(dt, *shape):
        nonlocal off
        a = np.frombuffer(section, dt, int(np.prod(shape)), off).reshape(shape); off += a.nbytes
        return a
    def take_q(o, n):
        qs, ss = [], []
        for _ in range(n_layers):
            if qt == 0: qs.append(take(F, o, n)); ss.append(None)
            elif qt == 2: qs.append(take(np.uint8, o, n // 2)); ss.append(take(F, o * n // gs))
            else: qs.append(take(np.int8, o, n)); ss.append(take(F, o * n // gs))
        return qs, ss
    kd = 3 * patch * patch
    cls = take(F, dim); pe = take(F, dim, kd); pos = take(F, 577, dim)
    ln1w, ln1b, ln2w, ln2b = take(F, n_layers, dim), take(F, n_layers, dim), take(F, n_layers, dim), take(F, n_layers, dim)
    wq, wqb = take_q(dim, dim), take(F, n_layers, dim)
    wk, wkb = take_q(dim, dim), take(F, n_layers, dim)
    wv, wvb = take_q(dim, dim), take(F, n_layers, dim)
    wo, wob = take_q(dim, dim), take(F, n_layers, dim)
    w1, w1b = take_q(hid, dim), take(F, n_layers, hid)
    w2, w2b = take_q(dim, hid), take(F, n_layers, dim)
    prew, preb = take(F, dim), take(F, dim)
    end = off

    npch = (image // patch) ** 2
    # conv as matmul_rest(out, x = kernel, w = pixels, n = 588, o = 576): out[d][p]; the tail reads x[r], i.e. kernel row 0
    nb = kd // 8 * 8
    body = lane_dot(pe[:, None, :nb], pixel_values[None, :, :nb])          # [dim, npch]
    for r in range(nb, kd):
        body = (body + (pixel_values[None, :, r] * pe[0, r]).astype(F)).astype(F)
    emb = np.concatenate([cls[None, :], body.T.copy()], axis=0)            # [577, dim]
    emb = (emb + pos).astype(F)
    T = npch + 1
    nrm = layernorm_rows(emb, prew, preb, eps)
    scale = F(np.sqrt(F(hs)))
    for l in range(n_layers - 1):
        x = nrm.copy()
        e = layernorm_rows(nrm, ln1w[l], ln1b[l], eps)
        # the three projections share one quantisation of the row
        q = ((quant_matmul_rows(e, wq[0][l], wq[1][l], dim, dim, gs, qt) + wqb[l]).astype(F) / scale).astype(F)
        k = (quant_matmul_rows(e, wk[0][l], wk[1][l], dim, dim, gs, qt) + wkb[l]).astype(F)
        v = (quant_matmul_rows(e, wv[0][l], wv[1][l], dim, dim, gs, qt) + wvb[l]).astype(F)
        ao = np.empty((T, dim), F)
        for h in range(n_heads):
            qh, kh, vh = q[:, h * hs:(h + 1) * hs], k[:, h * hs:(h + 1) * hs], v[:, h * hs:(h + 1) * hs]
            att = lane_dot(qh[:, None, :], kh[None, :, :])                 # [Tq, Tk]
            att = _expf_arr((att - att.max(axis=1, keepdims=True)).astype(F))
            ssum = np.zeros(T, F)
            for j in range(T):
                ssum = (ssum + att[:, j]).astype(F)
            att = (att / ssum[:, None]).astype(F)
            nb2 = T // 8 * 8
            o_ = lane_dot(att[:, None, :nb2], vh.T[None, :, :nb2])          # [Tq, hs]
            for r in range(nb2, T):
                o_ = (o_ + (vh[r][None, :] * att[:, r:r + 1]).astype(F)).astype(F)
            ao[:, h * hs:(h + 1) * hs] = o_
        e = (quant_matmul_rows(ao, wo[0][l], wo[1][l], dim, dim, gs, qt) + wob[l]).astype(F)
        e = (e + x).astype(F)
        x = e.copy()
        nrm = layernorm_rows(e, ln2w[l], ln2b[l], eps)
        hdn = (quant_matmul_rows(nrm, w1[0][l], w1[1][l], dim, hid, gs, qt) + w1b[l]).astype(F)
        sg = (F(1.0) / (F(1.0) + _expf_arr(-(F(1.702) * hdn).astype(F))).astype(F)).astype(F)
        hdn = (hdn * sg).astype(F)
        e = (quant_matmul_rows(hdn, w2[0][l], w2[1][l], hid, dim, gs, qt) + w2b[l]).astype(F)
        nrm = (e + x).astype(F)
    return end, nrm[1:]
