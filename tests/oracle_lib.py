"""ctypes binding of the CPU oracle (oracle/liblmrs_oracle.so).  TEST INFRASTRUCTURE: imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblmrs_oracle.so")


class Args(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("dim", "hidden_dim", "n_layers", "n_heads", "head_size", "n_kv_heads", "vocab_size", "seq_len")] + [
        ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float), ("q_type", C.c_uint8), ("model_type", C.c_uint8),
        ("multimodal", C.c_uint8), ("_pad", C.c_uint8), ("group_size", C.c_uint32)]


def build_oracle(force: bool = False):
    src = os.path.join(ORACLE_DIR, "lmrs_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "liblmrs_oracle.so"], check=True)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        vp, u32, sz, f32p = C.c_void_p, C.c_uint32, C.c_size_t, C.POINTER(C.c_float)
        L.lmrs_ref_last_error.restype = C.c_char_p
        L.lmrs_ref_create.argtypes = [vp, sz, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.lmrs_ref_destroy.argtypes = [vp]
        L.lmrs_ref_get_args.argtypes = [vp]; L.lmrs_ref_get_args.restype = C.POINTER(Args)
        L.lmrs_ref_forward.argtypes = [vp, u32, u32, C.POINTER(f32p)]
        L.lmrs_ref_forward_argmax.argtypes = [vp, u32, u32, C.POINTER(u32)]
        L.lmrs_ref_get_embeddings.argtypes = [vp, vp, sz, vp]
        L.lmrs_ref_fill_kv_cache.argtypes = [vp, vp, u32, u32, C.POINTER(u32)]
        L.lmrs_ref_generate_greedy.argtypes = [vp, vp, sz, u32, u32, vp, C.POINTER(C.c_double)]
        L.lmrs_ref_argmax.argtypes = [vp, sz]; L.lmrs_ref_argmax.restype = u32
        L.lmrs_ref_threads.restype = C.c_int
        L.lmrs_ref_set_threads.argtypes = [C.c_int]; L.lmrs_ref_set_threads.restype = None
        L.lmrs_ref_kv.argtypes = [vp, C.c_int, u32, u32]; L.lmrs_ref_kv.restype = f32p
        L.lmrs_ref_op_rmsnorm.argtypes = [vp, vp, vp, sz, C.c_float, C.c_int]; L.lmrs_ref_op_rmsnorm.restype = None
        L.lmrs_ref_op_softmax.argtypes = [vp, sz]; L.lmrs_ref_op_softmax.restype = None
        L.lmrs_ref_op_matmul_q8.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz]; L.lmrs_ref_op_matmul_q8.restype = None
        L.lmrs_ref_op_matmul_q4.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz]; L.lmrs_ref_op_matmul_q4.restype = None
        L.lmrs_ref_op_matmul_q4_batched_faithful.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz]; L.lmrs_ref_op_matmul_q4_batched_faithful.restype = None
        L.lmrs_ref_set_faithful_q9.argtypes = [C.c_int]; L.lmrs_ref_set_faithful_q9.restype = None
        L.lmrs_ref_get_faithful_q9.restype = C.c_int
        L.lmrs_ref_op_quantize.argtypes = [vp, vp, vp, sz, sz]; L.lmrs_ref_op_quantize.restype = None
        L.lmrs_ref_op_quantize_q4.argtypes = [vp, vp, vp, sz, sz]; L.lmrs_ref_op_quantize_q4.restype = None
        L.lmrs_ref_op_expf.argtypes = [C.c_float]; L.lmrs_ref_op_expf.restype = C.c_float
        L.lmrs_ref_op_glu.argtypes = [vp, vp, sz, C.c_int]; L.lmrs_ref_op_glu.restype = None
        L.lmrs_ref_op_tanh_cast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double]; L.lmrs_ref_op_tanh_cast.restype = None
        L.lmrs_ref_rope_terms.argtypes = [C.POINTER(Args), u32, u32, f32p, f32p]; L.lmrs_ref_rope_terms.restype = None
        L.lmrs_ref_random_u32.argtypes = [C.c_uint64]; L.lmrs_ref_random_u32.restype = u32
        L.lmrs_ref_random_f32.argtypes = [C.c_uint64]; L.lmrs_ref_random_f32.restype = C.c_float
        L.lmrs_ref_sampler_new.argtypes = [u32, C.c_float, C.c_float, C.c_uint64]; L.lmrs_ref_sampler_new.restype = vp
        L.lmrs_ref_sampler_free.argtypes = [vp]; L.lmrs_ref_sampler_free.restype = None
        L.lmrs_ref_sampler_sample.argtypes = [vp, vp]; L.lmrs_ref_sampler_sample.restype = C.c_int64
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Mirror of lmrs::transformer::Transformer over the oracle."""

    def __init__(self, image: np.ndarray):
        image = np.ascontiguousarray(image, np.uint8)
        h, used = C.c_void_p(), C.c_size_t()
        if lib().lmrs_ref_create(_p(image), image.size, 0, C.byref(h), C.byref(used)):
            raise RuntimeError(lib().lmrs_ref_last_error().decode())
        self.h, self.bytes_consumed = h, used.value
        self.args = lib().lmrs_ref_get_args(h).contents

    def __del__(self):
        if getattr(self, "h", None):
            lib().lmrs_ref_destroy(self.h); self.h = None

    def _chk(self, rc):
        if rc:
            raise RuntimeError(lib().lmrs_ref_last_error().decode())

    def forward(self, token: int, pos: int) -> np.ndarray:
        p = C.POINTER(C.c_float)()
        self._chk(lib().lmrs_ref_forward(self.h, token, pos, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(self.args.vocab_size,))

    def forward_argmax(self, token: int, pos: int) -> int:
        n = C.c_uint32()
        self._chk(lib().lmrs_ref_forward_argmax(self.h, token, pos, C.byref(n)))
        return n.value

    def get_embeddings(self, tokens) -> np.ndarray:
        t = np.ascontiguousarray(tokens, np.uint32)
        out = np.empty(t.size * self.args.dim, np.float32)
        self._chk(lib().lmrs_ref_get_embeddings(self.h, _p(t), t.size, _p(out)))
        return out

    def fill_kv_cache(self, embeddings: np.ndarray, curr_pos: int) -> int:
        assert embeddings.dtype == np.float32 and embeddings.flags.c_contiguous
        n = embeddings.size // self.args.dim
        newp = C.c_uint32()
        self._chk(lib().lmrs_ref_fill_kv_cache(self.h, _p(embeddings), n, curr_pos, C.byref(newp)))
        return newp.value

    def generate_greedy(self, prompt, n_new: int, start_pos: int = 0, timing: bool = False):
        pr = np.ascontiguousarray(prompt, np.uint32)
        out = np.zeros(n_new, np.uint32)
        sec = C.c_double()
        self._chk(lib().lmrs_ref_generate_greedy(self.h, _p(pr), pr.size, n_new, start_pos, _p(out), C.byref(sec)))
        return (out, sec.value) if timing else out

    def kv_row(self, which: int, layer: int, pos: int) -> np.ndarray:
        kv = self.args.n_kv_heads * self.args.head_size
        return np.ctypeslib.as_array(lib().lmrs_ref_kv(self.h, which, layer, pos), shape=(kv,)).copy()


# ---- free functions (functional.rs / quantization.rs)
def rmsnorm(x, w, eps, add_unit_offset=False):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    o = np.empty_like(x)
    lib().lmrs_ref_op_rmsnorm(_p(o), _p(x), _p(w), x.size, eps, int(add_unit_offset))
    return o


def softmax(x):
    x = np.array(x, np.float32, copy=True)
    lib().lmrs_ref_op_softmax(_p(x), x.size)
    return x


def quantize(x, gs=128):
    x = np.ascontiguousarray(x, np.float32)
    q = np.empty(x.size, np.int8); s = np.empty(x.size // gs, np.float32)
    lib().lmrs_ref_op_quantize(_p(q), _p(s), _p(x), x.size, gs)
    return q, s


def quantize_q4(x, gs=128):
    x = np.ascontiguousarray(x, np.float32)
    q = np.empty(x.size // 2, np.uint8); s = np.empty(x.size // gs, np.float32)
    lib().lmrs_ref_op_quantize_q4(_p(q), _p(s), _p(x), x.size, gs)
    return q, s


def matmul_q8(xq, xs, wq, ws, n, o, gs=128, sl=1):
    out = np.zeros(sl * o, np.float32)
    lib().lmrs_ref_op_matmul_q8(_p(out), _p(np.ascontiguousarray(xq, np.int8)), _p(np.ascontiguousarray(xs, np.float32)),
                                _p(np.ascontiguousarray(wq, np.int8)), _p(np.ascontiguousarray(ws, np.float32)), n, o, gs, sl)
    return out


def matmul_q4(xq, xs, wq, ws, n, o, gs=128):
    out = np.zeros(o, np.float32)
    lib().lmrs_ref_op_matmul_q4(_p(out), _p(np.ascontiguousarray(xq, np.uint8)), _p(np.ascontiguousarray(xs, np.float32)),
                                _p(np.ascontiguousarray(wq, np.uint8)), _p(np.ascontiguousarray(ws, np.float32)), n, o, gs)
    return out


def glu(gate, up, gemma=False) -> np.ndarray:
    """act(gate) * up elementwise (transformer.rs:607-624): SiLU, or Gemma's tanh-GELU."""
    h = np.ascontiguousarray(gate, np.float32).copy(); u = np.ascontiguousarray(up, np.float32)
    lib().lmrs_ref_op_glu(_p(h), _p(u), h.size, int(bool(gemma)))
    return h


def expf(x: float) -> float:
    return lib().lmrs_ref_op_expf(float(x))


def tanh_cast(x, c=1.0) -> np.ndarray:
    """(float)tanh(c * (double)x) with the host libm (numpy's own tanh is a SIMD implementation, not libm's)"""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().lmrs_ref_op_tanh_cast(_p(x), _p(y), x.size, float(c))
    return y


def random_u32(state: int) -> int:
    return int(lib().lmrs_ref_random_u32(state & ((1 << 64) - 1)))


def random_f32(state: int) -> float:
    return float(lib().lmrs_ref_random_f32(state & ((1 << 64) - 1)))


class Sampler:
    """Mirror of lmrs::sampler::Sampler (src/sampler.rs:10-129) over the oracle: the candidate vector persists across calls."""

    def __init__(self, vocab_size: int, temperature: float, top_p: float, seed: int):
        self.h = lib().lmrs_ref_sampler_new(vocab_size, temperature, top_p, seed & ((1 << 64) - 1))
        if not self.h:
            raise RuntimeError("lmrs_ref_sampler_new failed")
        self.vocab_size = vocab_size

    def __del__(self):
        if getattr(self, "h", None):
            lib().lmrs_ref_sampler_free(self.h); self.h = None

    def sample(self, logits: np.ndarray) -> int:
        """logits: float32[vocab_size], scaled and soft-maxed IN PLACE when temperature != 0 (sampler.rs:115-117)."""
        assert logits.dtype == np.float32 and logits.size == self.vocab_size and logits.flags.c_contiguous
        t = int(lib().lmrs_ref_sampler_sample(self.h, _p(logits)))
        if t < 0:
            raise RuntimeError("sample_topp: no candidate above the cutoff (the reference panics)")
        return t


def threads() -> int:
    return lib().lmrs_ref_threads()


def set_threads(n: int):
    lib().lmrs_ref_set_threads(n)


class VisionOracle:
    """CPU restatement of the reference's CLIP tower (src/vision.rs), Q8_0."""

    def __init__(self, section: np.ndarray):
        L = lib()
        L.lmrs_ref_vision_create.restype = C.c_int
        L.lmrs_ref_vision_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.lmrs_ref_vision_forward.restype = C.c_int
        L.lmrs_ref_vision_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        L.lmrs_ref_vision_destroy.argtypes = [C.c_void_p]
        L.lmrs_ref_vision_args.argtypes = [C.c_void_p, C.c_void_p]
        sec = np.ascontiguousarray(section, np.uint8)
        h = C.c_void_p(); used = C.c_size_t()
        if L.lmrs_ref_vision_create(sec.ctypes.data, sec.size, C.byref(h), C.byref(used)):
            raise RuntimeError(L.lmrs_ref_last_error().decode())
        self._h, self.bytes_consumed = h, used.value
        a = np.zeros(8, np.uint32); L.lmrs_ref_vision_args(h, a.ctypes.data)
        self.dim, self.n_layers, self.image_size, self.patch_size = int(a[0]), int(a[2]), int(a[6]), int(a[5])

    def forward(self, pixel_values: np.ndarray, num_crops: int) -> np.ndarray:
        pv = np.ascontiguousarray(pixel_values, np.float32).reshape(-1)
        n = (self.image_size // self.patch_size) ** 2
        out = np.zeros(num_crops * n * self.dim, np.float32); ns = C.c_uint32()
        if lib().lmrs_ref_vision_forward(self._h, pv.ctypes.data, num_crops, out.ctypes.data, C.byref(ns)):
            raise RuntimeError(lib().lmrs_ref_last_error().decode())
        assert ns.value == n * self.dim
        return out.reshape(num_crops, n, self.dim)

    def __del__(self):
        try:
            lib().lmrs_ref_vision_destroy(self._h)
        except Exception:
            pass


class ProcessorOracle:
    """CPU restatement of the reference's PHI3VProcessor (src/processor.rs:168-342), Q8_0."""

    def __init__(self, section: np.ndarray):
        L = lib()
        L.lmrs_ref_processor_create.restype = C.c_int
        L.lmrs_ref_processor_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.lmrs_ref_processor_forward.restype = C.c_int
        L.lmrs_ref_processor_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.POINTER(C.c_uint32)]
        L.lmrs_ref_processor_destroy.argtypes = [C.c_void_p]
        sec = np.ascontiguousarray(section, np.uint8)
        h = C.c_void_p(); used = C.c_size_t()
        if L.lmrs_ref_processor_create(sec.ctypes.data, sec.size, C.byref(h), C.byref(used)):
            raise RuntimeError(L.lmrs_ref_last_error().decode())
        self._h, self.bytes_consumed = h, used.value
        self.text_dim = int(np.frombuffer(sec[4:8].tobytes(), np.uint32)[0])

    def forward(self, out_patches: np.ndarray, new_shape: int, patch_side: int, w_crop: int, h_crop: int) -> np.ndarray:
        op = np.ascontiguousarray(out_patches, np.float32).reshape(-1)
        ne = (h_crop * patch_side) * (w_crop * patch_side + 1) + patch_side * (patch_side + 1) + 1
        out = np.zeros(ne * self.text_dim, np.float32); n = C.c_uint32()
        if lib().lmrs_ref_processor_forward(self._h, op.ctypes.data, op.size, new_shape, patch_side, w_crop, h_crop, out.ctypes.data, C.byref(n)):
            raise RuntimeError(lib().lmrs_ref_last_error().decode())
        assert n.value == ne
        return out.reshape(ne, self.text_dim)

    def __del__(self):
        try:
            lib().lmrs_ref_processor_destroy(self._h)
        except Exception:
            pass


class faithful_q9:
    """with faithful_q9(): batched forward_layer on Q4_0 files computes what the REFERENCE computes (token j's activations
    taken at byte j*n of the packed tensor), not the token-by-token form the library implements.  Process-wide switch, restored on exit."""

    def __enter__(self):
        self.prev = lib().lmrs_ref_get_faithful_q9()
        lib().lmrs_ref_set_faithful_q9(1)
        return self

    def __exit__(self, *exc):
        lib().lmrs_ref_set_faithful_q9(self.prev)
        return False
