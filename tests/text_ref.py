"""Second transcription (pure Python, written from the Rust sources, not from lmrs_text.cpp) of the reference's tokenizer and sampler:
src/tokenizer.rs:24-163, src/sampler.rs:19-129, src/functional.rs:34-44 (random_u32 / random_f32) and :122-140 (softmax).
TEST INFRASTRUCTURE.  Floats are numpy float32 scalars (every operation rounds to f32); f32::exp is the host libm's expf."""
import bisect
import ctypes
import struct

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]
f32 = np.float32


class Tokenizer:
    def __init__(self, data: bytes, bsearch_flavour: int = 0):
        self.vocab_size, _max_len, self.bos, self.eos = struct.unpack_from("IIII", data, 0)
        self.vocab, self.scores = [], []
        off = 16
        for _ in range(self.vocab_size):
            score, n = struct.unpack_from("fI", data, off); off += 8
            self.vocab.append(data[off:off + n].decode("utf-8")); off += n          # String::from_utf8(...).expect(...)
            self.scores.append(score)
        self.sorted = None
        self.flavour = bsearch_flavour

    def _find(self, text: str):
        """binary_search_by(|t| t.text.cmp(text)) over sorted_vocab; Rust compares Strings byte-wise."""
        key = text.encode("utf-8")
        keys = self._keys
        if self.flavour == 0:                      # std of Rust 1.52 .. 1.81
            left, right = 0, len(keys); size = right
            while left < right:
                mid = left + size // 2
                if keys[mid] < key: left = mid + 1
                elif keys[mid] > key: right = mid
                else: return mid
                size = right - left
            return None
        base, size = 0, len(keys)                  # later std: branch-free halving
        if size == 0:
            return None
        while size > 1:
            half = size // 2; mid = base + half
            if not keys[mid] > key: base = mid
            size -= half
        return base if keys[base] == key else None

    def encode(self, text: str, bos: bool, eos: bool, chat_format: bool, model_type: int):
        assert text, "Text to encode should not be empty"
        if self.sorted is None:
            order = sorted(range(self.vocab_size), key=lambda i: self.vocab[i].encode("utf-8"))   # sort_by is stable; so is sorted()
            self.sorted = order
            self._keys = [self.vocab[i].encode("utf-8") for i in order]
        tokens = []
        if bos:
            tokens.append(self.bos)
        if chat_format:
            tokens += {0: [self.bos, 106, 1645, 108], 1: [128006, 882, 128007, 271], 2: [self.bos, 32010, 29871, 13]}[model_type]
        for ch in text:
            idx = self._find(ch)
            if idx is not None:
                tokens.append(self.sorted[idx])
            else:
                tokens += [b + 3 for b in ch.encode("utf-8")]
        while True:
            best_score, best_id, best_idx = f32(-1e10), 0, -1
            for i in range(len(tokens) - 1):
                idx = self._find(self.vocab[tokens[i]] + self.vocab[tokens[i + 1]])
                if idx is not None:
                    tid = self.sorted[idx]
                    if f32(self.scores[tid]) > best_score:
                        best_score, best_id, best_idx = f32(self.scores[tid]), tid, i
            if best_idx == -1:
                break
            tokens[best_idx] = best_id
            del tokens[best_idx + 1]
        if chat_format:
            tokens += {0: [107, 108, 106, 2516, 108], 1: [128009, 128006, 78191, 128007, 271], 2: [32007, 29871, 13, 32001, 29871, 13]}[model_type]
        if eos:
            tokens.append(self.eos)
        return tokens

    def decode(self, token: int) -> str:
        piece = self.vocab[token]
        if piece.startswith("<0x") and piece.endswith(">") and len(piece.encode("utf-8")) == 6:
            try:
                return chr(int(piece[3:5], 16))
            except ValueError:
                pass
        return piece


def random_u32(state: int) -> int:
    m = (1 << 64) - 1
    state ^= state >> 12
    state ^= (state << 25) & m
    state ^= state >> 27
    return ((state * 0x2545F4914F6CDD1D) & m) >> 32


def random_f32(state: int):
    return f32(f32(random_u32(state) >> 8) / f32(16777216.0))


def softmax(x):
    s, mx = f32(0.0), x[0]
    for v in x:
        if v > mx:
            mx = v
    for i in range(len(x)):
        x[i] = f32(_libm.expf(float(f32(x[i] - mx))))
        s = f32(s + x[i])
    for i in range(len(x)):
        x[i] = f32(x[i] / s)


class Sampler:
    def __init__(self, vocab_size, temperature, top_p, seed):
        self.vocab_size, self.temperature, self.top_p, self.seed = vocab_size, f32(temperature), f32(top_p), seed
        self.probindex = [(f32(0.0), 0)] * vocab_size          # (prob, index)

    def sample(self, logits):
        """logits: list of np.float32, modified in place like the reference's slice."""
        n = self.vocab_size
        if self.temperature == f32(0.0):
            mi, mp = 0, logits[0]
            for i in range(1, n):
                if logits[i] > mp:
                    mi, mp = i, logits[i]
            return mi
        for q in range(n):
            logits[q] = f32(logits[q] / self.temperature)
        softmax(logits)
        rand = random_f32(self.seed)                             # the seed is never advanced (sampler.rs:119)
        if self.top_p <= f32(0.0) or self.top_p >= f32(1.0):
            cdf = f32(0.0)
            for i in range(n):
                cdf = f32(cdf + logits[i])
                if rand < cdf:
                    return i
            return n - 1
        n0 = 0
        cutoff = f32(f32(f32(1.0) - self.top_p) / f32(n - 1))
        for i in range(n):
            if logits[i] >= cutoff:
                self.probindex[n0] = (logits[i], i); n0 += 1
        self.probindex.sort(key=lambda pi: -float(pi[0]))         # stable, descending by prob, the WHOLE vector (sampler.rs:81)
        cum, last = f32(0.0), n0 - 1
        for i in range(n0):
            cum = f32(cum + self.probindex[i][0])
            if cum > self.top_p:
                last = i
                break
        r = f32(rand * cum)
        cdf = f32(0.0)
        for i in range(last + 1):
            cdf = f32(cdf + self.probindex[i][0])
            if r < cdf:
                return self.probindex[i][1]
        return self.probindex[last][1]
