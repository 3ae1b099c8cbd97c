"""lm.rs_amd — MI355X-native decode hot path for lm.rs models (host-side mirror, Python flavour).

`Transformer` mirrors the public surface of `lmrs::transformer::Transformer`
(reference src/transformer.rs:127-131, :134, :316, :659, :672) over the C ABI of
`liblmrs_hip.so` (include/lmrs_hip.h).  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, construction raises.

The directory name contains a dot, so import it through the repo-root shim:  `import lmrs_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblmrs_hip.so")
CSRC = os.path.join(_HERE, "csrc")

GEMMA, LLAMA, PHI = 0, 1, 2
Q_NONE, Q8_0, Q4_0 = 0, 1, 2

# every symbol include/lmrs_hip.h declares (tests/test_abi.py checks the header against this list)
EXPORTS = [
    "lmrs_create", "lmrs_create_sharded", "lmrs_comm_unique_id", "lmrs_destroy", "lmrs_get_args", "lmrs_forward",
    "lmrs_forward_argmax", "lmrs_get_embeddings", "lmrs_fill_kv_cache", "lmrs_generate_greedy", "lmrs_last_error",
    "lmrs_op_matmul_q8", "lmrs_op_matmul_q4", "lmrs_op_quantize", "lmrs_op_quantize_q4", "lmrs_op_rmsnorm",
    "lmrs_op_softmax", "lmrs_op_expf", "lmrs_op_tanh_cast", "lmrs_forward_sample", "lmrs_sampler_info", "lmrs_op_sample_mult", "lmrs_op_classifier_argmax", "lmrs_bench_gemv", "lmrs_bench_step", "lmrs_step_info", "lmrs_debug_timeline", "lmrs_debug_kv", "lmrs_debug_inject", "lmrs_last_fill_ms", "lmrs_debug_gemm_tile", "lmrs_debug_w13_quant",
    "lmrs_group_create", "lmrs_group_forward", "lmrs_shard_plan", "lmrs_shard_uses_graph", "lmrs_comm_ranks", "lmrs_p2p_handle", "lmrs_p2p_connect",
    "lmrs_vision_create", "lmrs_vision_destroy", "lmrs_vision_forward",
    "lmrs_processor_create", "lmrs_processor_destroy", "lmrs_processor_forward", "lmrs_processor_hd_transform", "lmrs_rope_terms",
    "lmrs_tokenizer_create", "lmrs_tokenizer_destroy", "lmrs_tokenizer_info", "lmrs_tokenizer_encode", "lmrs_tokenizer_decode",
    "lmrs_sampler_create", "lmrs_sampler_destroy", "lmrs_sampler_sample", "lmrs_sampler_sample_exps", "lmrs_sampler_topp_pairs",
    "lmrs_sampler_exps_prepare", "lmrs_sampler_exps_finish",
]


class TransformerArgs(C.Structure):
    """lmrs_args / TransformerArgs (src/transformer.rs:57-74)."""
    _fields_ = [(n, C.c_uint32) for n in ("dim", "hidden_dim", "n_layers", "n_heads", "head_size", "n_kv_heads", "vocab_size", "seq_len")] + [
        ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float), ("q_type", C.c_uint8), ("model_type", C.c_uint8),
        ("multimodal", C.c_uint8), ("_pad", C.c_uint8), ("group_size", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile liblmrs_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".inc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "lmrs_hip.h")); srcs.append(os.path.join(CSRC, "Makefile"))     # (the flags are part of the build)
    def stale():
        return not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale():
        # one builder at a time: the ranks of a multi-GPU launch all come through here (bench.py), and two `make`s writing the same
        # objects corrupt them; whoever waited finds the library fresh and skips the build
        import fcntl
        with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or stale():
                    subprocess.run(["make", "-s", "-j4", "-C", CSRC], check=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("LMRS_LIB", LIB_PATH)          # (A/B builds of the same sources: tools/ab_bench.sh)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is mandatory; there is no CPU fallback)")
        L = C.CDLL(path)
        vp, u32, sz, f32p = C.c_void_p, C.c_uint32, C.c_size_t, C.POINTER(C.c_float)
        L.lmrs_last_error.restype = C.c_char_p
        L.lmrs_create.argtypes = [vp, sz, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.lmrs_create_sharded.argtypes = [vp, sz, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp), C.POINTER(sz)]
        L.lmrs_comm_unique_id.argtypes = [vp]
        L.lmrs_destroy.argtypes = [vp]; L.lmrs_destroy.restype = None
        L.lmrs_get_args.argtypes = [vp]; L.lmrs_get_args.restype = C.POINTER(TransformerArgs)
        L.lmrs_forward.argtypes = [vp, u32, u32, C.POINTER(f32p)]
        L.lmrs_forward_argmax.argtypes = [vp, u32, u32, C.POINTER(u32)]
        L.lmrs_get_embeddings.argtypes = [vp, vp, sz, vp]
        L.lmrs_fill_kv_cache.argtypes = [vp, vp, u32, u32, C.POINTER(u32)]
        L.lmrs_generate_greedy.argtypes = [vp, vp, sz, u32, u32, vp, C.POINTER(C.c_double)]
        L.lmrs_op_matmul_q8.argtypes = [C.c_int, vp, vp, vp, vp, vp, sz, sz, sz, sz]
        L.lmrs_op_matmul_q4.argtypes = [C.c_int, vp, vp, vp, vp, vp, sz, sz, sz]
        L.lmrs_op_quantize.argtypes = [C.c_int, vp, vp, vp, sz, sz]
        L.lmrs_op_quantize_q4.argtypes = [C.c_int, vp, vp, vp, sz, sz]
        L.lmrs_op_rmsnorm.argtypes = [C.c_int, vp, vp, vp, sz, C.c_float, C.c_int]
        L.lmrs_op_softmax.argtypes = [C.c_int, vp, sz]
        L.lmrs_op_expf.argtypes = [C.c_int, vp, vp, sz]
        L.lmrs_op_tanh_cast.argtypes = [C.c_int, vp, vp, sz, C.c_double]
        L.lmrs_bench_gemv.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.lmrs_step_info.argtypes = [vp, u32, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.lmrs_debug_timeline.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.lmrs_shard_plan.argtypes = [C.POINTER(TransformerArgs), C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.lmrs_shard_uses_graph.argtypes = [vp]
        L.lmrs_comm_ranks.argtypes = [vp]
        L.lmrs_group_create.argtypes = [vp, sz, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.lmrs_group_forward.argtypes = [C.POINTER(vp), C.c_int, u32, u32, C.POINTER(f32p), C.POINTER(u32)]
        L.lmrs_vision_create.argtypes = [vp, sz, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.lmrs_vision_destroy.argtypes = [vp]
        L.lmrs_vision_destroy.restype = None
        L.lmrs_vision_forward.argtypes = [vp, vp, u32, vp, C.POINTER(u32)]
        L.lmrs_processor_create.argtypes = [vp, sz, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.lmrs_processor_destroy.argtypes = [vp]
        L.lmrs_processor_destroy.restype = None
        L.lmrs_processor_forward.argtypes = [vp, vp, u32, u32, u32, u32, u32, vp, C.POINTER(u32)]
        L.lmrs_processor_hd_transform.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, C.POINTER(u32)]
        L.lmrs_rope_terms.argtypes = [C.POINTER(TransformerArgs), u32, u32, f32p, f32p]
        L.lmrs_tokenizer_create.argtypes = [vp, sz, C.POINTER(vp)]
        L.lmrs_tokenizer_destroy.argtypes = [vp]; L.lmrs_tokenizer_destroy.restype = None
        L.lmrs_tokenizer_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        L.lmrs_tokenizer_encode.argtypes = [vp, C.c_char_p, sz, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, C.POINTER(sz)]
        L.lmrs_tokenizer_decode.argtypes = [vp, u32, C.c_char_p, sz, C.POINTER(sz)]
        L.lmrs_sampler_create.argtypes = [u32, C.c_float, C.c_float, C.c_uint64, C.POINTER(vp)]
        L.lmrs_sampler_destroy.argtypes = [vp]; L.lmrs_sampler_destroy.restype = None
        L.lmrs_sampler_sample.argtypes = [vp, vp, C.POINTER(u32)]
        L.lmrs_sampler_topp_pairs.argtypes = [vp, vp, sz, C.POINTER(u32)]
        L.lmrs_sampler_sample_exps.argtypes = [vp, vp, C.POINTER(u32)]
        L.lmrs_sampler_exps_prepare.argtypes = [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_size_t)]
        L.lmrs_sampler_exps_finish.argtypes = [vp, vp, vp, C.POINTER(u32)]
        L.lmrs_sampler_info.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.lmrs_forward_sample.argtypes = [vp, u32, u32, vp, C.POINTER(u32)]
        L.lmrs_op_sample_mult.argtypes = [C.c_int, vp, sz, C.c_float, C.c_float, C.POINTER(u32)]
        L.lmrs_debug_kv.argtypes = [vp, C.c_int, u32, u32, vp]
        L.lmrs_debug_inject.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.lmrs_last_fill_ms.argtypes = [vp, C.POINTER(C.c_double)]
        L.lmrs_debug_gemm_tile.argtypes = [u32, u32, u32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lmrs_debug_w13_quant.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, sz, sz, sz, C.c_int]
        L.lmrs_p2p_handle.argtypes = [vp, vp]
        L.lmrs_p2p_connect.argtypes = [vp, vp]
        L.lmrs_bench_step.argtypes = [vp, u32, C.c_int, vp, vp, vp]
        L.lmrs_op_classifier_argmax.argtypes = [C.c_int, vp, vp, vp, vp, sz, sz, C.c_float, C.POINTER(u32), vp]
        _lib = L
    return _lib


class LmrsError(RuntimeError):
    """The reference panics (assert!/expect); the C ABI returns a status and we raise."""


def _chk(rc):
    if rc:
        raise LmrsError(lib().lmrs_last_error().decode())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Transformer:
    """Drop-in for lmrs::transformer::Transformer on one MI355X.

    Transformer(data) -> like Transformer::new(&mmap): `data` is the LMRS image (bytes-like /
    numpy uint8 / np.memmap).  `.bytes_consumed` is the second element of the reference's tuple.
    """

    def __init__(self, data, device: int = 0, rank: int = 0, world: int = 1, unique_id: bytes | None = None):
        image = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
        h, used = C.c_void_p(), C.c_size_t()
        if world == 1 and unique_id is None:
            _chk(lib().lmrs_create(_p(image), image.size, device, C.byref(h), C.byref(used)))
        else:
            uid = C.create_string_buffer(unique_id, 128) if unique_id else None
            _chk(lib().lmrs_create_sharded(_p(image), image.size, device, rank, world, uid, C.byref(h), C.byref(used)))
        self._h = h
        self.bytes_consumed = used.value
        self.args = lib().lmrs_get_args(h).contents

    def close(self):
        if getattr(self, "_h", None):
            lib().lmrs_destroy(self._h)
            self._h = None

    __del__ = close

    def forward(self, token: int, pos: int) -> np.ndarray:
        """-> logits view (vocab_size f32, pinned host memory owned by the model, valid until the next call)."""
        p = C.POINTER(C.c_float)()
        _chk(lib().lmrs_forward(self._h, token, pos, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(self.args.vocab_size,))

    def forward_argmax(self, token: int, pos: int) -> int:
        n = C.c_uint32()
        _chk(lib().lmrs_forward_argmax(self._h, token, pos, C.byref(n)))
        return n.value

    def forward_sample(self, token: int, pos: int, sampler: "Sampler") -> int:
        """forward + Sampler::sample: temperature 0 -> the fused argmax; else scaling / max / exp on the device, the sequential chains on the host, the sort of
        many top-p candidates on the device again (lmrs_forward_sample)"""
        n = C.c_uint32()
        _chk(lib().lmrs_forward_sample(self._h, token, pos, sampler._h, C.byref(n)))
        return n.value

    def get_embeddings(self, tokens) -> np.ndarray:
        t = np.ascontiguousarray(tokens, np.uint32)
        out = np.empty(t.size * self.args.dim, np.float32)
        _chk(lib().lmrs_get_embeddings(self._h, _p(t), t.size, _p(out)))
        return out

    def fill_kv_cache(self, embeddings: np.ndarray, curr_pos: int) -> int:
        if embeddings.dtype != np.float32 or not embeddings.flags.c_contiguous or embeddings.size % self.args.dim:
            raise LmrsError("embeddings must be a contiguous float32 array of n*dim elements")
        newp = C.c_uint32()
        _chk(lib().lmrs_fill_kv_cache(self._h, _p(embeddings), embeddings.size // self.args.dim, curr_pos, C.byref(newp)))
        return newp.value

    def generate_greedy(self, prompt, n_new: int, start_pos: int = 0, timing: bool = False):
        """chat.rs:188-222 on token IDs: returns the n_new greedy tokens (and device seconds if timing)."""
        pr = np.ascontiguousarray(prompt, np.uint32)
        out = np.zeros(n_new, np.uint32)
        sec = C.c_double()
        _chk(lib().lmrs_generate_greedy(self._h, _p(pr), pr.size, n_new, start_pos, _p(out), C.byref(sec)))
        return (out, sec.value) if timing else out

    def kv_row(self, which: int, layer: int, pos: int) -> np.ndarray:
        """Verification aid: one KV-cache row in the reference's layout (which: 0 key, 1 value)."""
        out = np.empty(self.args.n_kv_heads * self.args.head_size, np.float32)
        _chk(lib().lmrs_debug_kv(self._h, which, layer, pos, _p(out)))
        return out

    def last_fill_ms(self) -> float:
        """device milliseconds of the last batched fill_kv_cache without its host <-> device copies (lmrs_last_fill_ms)"""
        ms = C.c_double()
        _chk(lib().lmrs_last_fill_ms(self._h, C.byref(ms)))
        return ms.value

    def debug_inject(self, what: int, a: int = 0, b: int = 0) -> None:
        """Fault injection for the multi-GPU tests (include/lmrs_hip.h: lmrs_debug_inject)."""
        _chk(lib().lmrs_debug_inject(self._h, what, a, b))

    def p2p_handle(self) -> bytes:
        """Peer-to-peer sharded context: the 64-byte IPC handle of this rank's exchange arena (send it to every peer)."""
        buf = C.create_string_buffer(64)
        _chk(lib().lmrs_p2p_handle(self._h, buf))
        return buf.raw

    def p2p_connect(self, handles) -> None:
        """handles: the `world` handles in rank order."""
        blob = b"".join(handles)
        _chk(lib().lmrs_p2p_connect(self._h, C.create_string_buffer(blob, len(blob))))

    def comm_ranks(self) -> int:
        """ranks of the RCCL communicator (ncclCommCount); 0: none (one GPU / peer-to-peer transport)"""
        return lib().lmrs_comm_ranks(self._h)

    def shard_uses_graph(self) -> int:
        return lib().lmrs_shard_uses_graph(self._h)

    # ---- measurement hooks
    def bench_gemv(self, iters: int = 5):
        """-> {shape: (sum_us, sum_bytes, launches)} over `iters` GEMV-only passes of one decode step."""
        us, b, n = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_int * 5)()
        _chk(lib().lmrs_bench_gemv(self._h, iters, us, b, n))
        names = ["qkv", "wo", "w1w3", "w2", "classifier"]
        return {names[k]: (us[k], b[k], n[k]) for k in range(5)}

    def bench_step(self, pos: int, iters: int = 8):
        """Per-kernel durations of the real decode step (eager replay with events on every dispatch) -> {kind: (us, bytes, launches)}."""
        us = np.zeros(9, np.float64); b = np.zeros(9, np.float64); n = np.zeros(9, np.int32)
        _chk(lib().lmrs_bench_step(self._h, pos, iters, _p(us), _p(b), _p(n)))
        names = ("qkv", "attention", "wo", "w1w3", "w2", "classifier", "argmax", "glue", "exchange")
        return {k: (float(us[i]), float(b[i]), int(n[i])) for i, k in enumerate(names) if n[i]}

    def debug_timeline(self):
        """-> uint64[n_kernels, 8] wall-clock stamps (10 ns units) of the last decode step (needs LMRS_DEBUG_TIMELINE=1)."""
        buf = np.zeros((1024, 8), np.uint64); n = C.c_int()
        _chk(lib().lmrs_debug_timeline(self._h, _p(buf), 1024, C.byref(n)))
        return buf[: n.value]

    def step_info(self, pos: int):
        n, b = C.c_int(), C.c_double()
        _chk(lib().lmrs_step_info(self._h, pos, C.byref(n), C.byref(b)))
        return n.value, b.value


def shard_plan(args: TransformerArgs, rank: int, world: int) -> dict:
    """Row ranges (first, count) shard `rank` of `world` owns; host-only."""
    p = (C.c_int * 10)()
    _chk(lib().lmrs_shard_plan(C.byref(args), rank, world, p))
    k = ["q_heads", "kv_heads", "dim_rows", "hidden_pairs", "vocab_rows"]
    return {k[i]: (p[2 * i], p[2 * i + 1]) for i in range(5)}


def comm_unique_id() -> bytes:
    """128-byte ncclUniqueId for Transformer(..., world > 1): make it on rank 0, broadcast it to the other ranks."""
    buf = C.create_string_buffer(128)
    _chk(lib().lmrs_comm_unique_id(buf))
    return buf.raw


class ShardGroup:
    """`world` row shards of one model on ONE device (verification aid: the multi-GPU partitioning without RCCL)."""

    def __init__(self, data, world: int, device: int = 0):
        image = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
        self._arr = (C.c_void_p * world)()
        used = C.c_size_t()
        _chk(lib().lmrs_group_create(_p(image), image.size, device, world, self._arr, C.byref(used)))
        self.world = world
        self.args = lib().lmrs_get_args(self._arr[0]).contents

    def forward(self, token: int, pos: int):
        p, n = C.POINTER(C.c_float)(), C.c_uint32()
        _chk(lib().lmrs_group_forward(self._arr, self.world, token, pos, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(p, shape=(self.args.vocab_size,)), n.value

    def close(self):
        for i in range(self.world):
            if self._arr[i]:
                lib().lmrs_destroy(self._arr[i]); self._arr[i] = None

    __del__ = close


# ---- free functions (functional.rs / quantization.rs), each on the device kernels
def matmul_q8(xq, xs, wq, ws, n, o, gs=128, sl=1, device=0):
    out = np.zeros(sl * o, np.float32)
    _chk(lib().lmrs_op_matmul_q8(device, _p(out), _p(np.ascontiguousarray(xq, np.int8)), _p(np.ascontiguousarray(xs, np.float32)),
                                 _p(np.ascontiguousarray(wq, np.int8)), _p(np.ascontiguousarray(ws, np.float32)), n, o, gs, sl))
    return out


def w13_quant(xq, xs, wq, ws, n, o, n_tok, gemma=False, device=0):
    """The batched w1/w3 projection with the activation and the next matmul's quantiser in its epilogue (lmrs_debug_w13_quant):
    -> (n_tok x o/2 int8, n_tok x o/256 scales)."""
    hq = np.empty(n_tok * (o // 2), np.int8); hs = np.empty(n_tok * (o // 256), np.float32)
    _chk(lib().lmrs_debug_w13_quant(device, _p(hq), _p(hs), _p(np.ascontiguousarray(xq, np.int8)), _p(np.ascontiguousarray(xs, np.float32)),
                                    _p(np.ascontiguousarray(wq, np.int8)), _p(np.ascontiguousarray(ws, np.float32)), n, o, n_tok, int(bool(gemma))))
    return hq, hs


def matmul_q4(xq, xs, wq, ws, n, o, gs=128, device=0):
    out = np.zeros(o, np.float32)
    _chk(lib().lmrs_op_matmul_q4(device, _p(out), _p(np.ascontiguousarray(xq, np.uint8)), _p(np.ascontiguousarray(xs, np.float32)),
                                 _p(np.ascontiguousarray(wq, np.uint8)), _p(np.ascontiguousarray(ws, np.float32)), n, o, gs))
    return out


def quantize(x, gs=128, device=0):
    x = np.ascontiguousarray(x, np.float32)
    q = np.empty(x.size, np.int8); s = np.empty(x.size // gs, np.float32)
    _chk(lib().lmrs_op_quantize(device, _p(q), _p(s), _p(x), x.size, gs))
    return q, s


def quantize_q4(x, gs=128, device=0):
    x = np.ascontiguousarray(x, np.float32)
    q = np.empty(x.size // 2, np.uint8); s = np.empty(x.size // gs, np.float32)
    _chk(lib().lmrs_op_quantize_q4(device, _p(q), _p(s), _p(x), x.size, gs))
    return q, s


def rmsnorm(x, w, eps, add_unit_offset=False, device=0):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    o = np.empty_like(x)
    _chk(lib().lmrs_op_rmsnorm(device, _p(o), _p(x), _p(w), x.size, eps, int(add_unit_offset)))
    return o


def softmax(x, device=0):
    x = np.array(x, np.float32, copy=True)
    _chk(lib().lmrs_op_softmax(device, _p(x), x.size))
    return x


def classifier_argmax(x, rms_w, wq, ws, eps, device=0):
    """Final rmsnorm + quantize + matmul_q8 (transformer.rs:341-381) + sample_argmax (sampler.rs:29-41) -> (token, logits)."""
    x = np.ascontiguousarray(x, np.float32); rms_w = np.ascontiguousarray(rms_w, np.float32)
    wq = np.ascontiguousarray(wq, np.int8); ws = np.ascontiguousarray(ws, np.float32)
    n = x.size; o = wq.size // n
    tok = C.c_uint32(); logits = np.empty(o, np.float32)
    _chk(lib().lmrs_op_classifier_argmax(device, _p(x), _p(rms_w), _p(wq), _p(ws), n, o, eps, C.byref(tok), _p(logits)))
    return tok.value, logits


def expf(x, device=0):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    _chk(lib().lmrs_op_expf(device, _p(y), _p(x), x.size))
    return y


def sample_mult(logits, temperature: float, rnd: float, device=0):
    """Sampler::sample (temperature != 0, sample_mult) on the device -> (token, the probabilities the logits were turned into)"""
    lg = np.ascontiguousarray(logits, np.float32).copy()
    tok = C.c_uint32()
    _chk(lib().lmrs_op_sample_mult(device, _p(lg), lg.size, C.c_float(temperature), C.c_float(rnd), C.byref(tok)))
    return tok.value, lg


def tanh_cast(x, c=1.0, device=0):
    """(float)tanh(c * (double)x) as the kernels evaluate it (Gemma soft-caps: c = 1; tanh-GELU: c = 0.7978845608028654)"""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    _chk(lib().lmrs_op_tanh_cast(device, _p(y), _p(x), x.size, float(c)))
    return y


class VisionTransformer:
    """lmrs::vision::VisionTransformer (src/vision.rs): the CLIP image tower of the multimodal models, on the device."""

    def __init__(self, section: np.ndarray, device: int = 0):
        sec = np.ascontiguousarray(section, np.uint8)
        h = C.c_void_p(); used = C.c_size_t()
        _chk(lib().lmrs_vision_create(sec.ctypes.data, sec.size, device, C.byref(h), C.byref(used)))
        self._h, self.bytes_consumed = h, used.value

    def forward(self, pixel_values: np.ndarray, num_crops: int) -> np.ndarray:
        """-> float32 [num_crops, 576, dim] (vision.rs:244-577; the class token is dropped)."""
        pv = np.ascontiguousarray(pixel_values, np.float32).reshape(-1)
        if pv.size != num_crops * 3 * 336 * 336:
            raise LmrsError("pixel_values must hold num_crops * 3 * 336 * 336 floats")
        out = np.empty(num_crops * 576 * 1024, np.float32); ns = C.c_uint32()
        _chk(lib().lmrs_vision_forward(self._h, pv.ctypes.data, num_crops, out.ctypes.data, C.byref(ns)))
        return out.reshape(num_crops, 576, ns.value // 576)

    def close(self):
        if self._h:
            lib().lmrs_vision_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rope_terms(model_type: int, rope_theta: float, head_size: int, pos: int, j: int):
    """(cos, sin) of the RoPE table lmrs_create builds, for one position and pair index (host-only, no GPU)."""
    a = TransformerArgs(); a.model_type = model_type; a.rope_theta = rope_theta; a.head_size = head_size
    c, s_ = C.c_float(), C.c_float()
    _chk(lib().lmrs_rope_terms(C.byref(a), pos, j, C.byref(c), C.byref(s_)))
    return np.float32(c.value), np.float32(s_.value)


def gemm_tile(n: int, o: int, n_tok: int, q4: bool = False):
    """(weight rows, tokens, waves) of the workgroup tile the batched matmul_q8 / matmul_q4 runs this launch with (host arithmetic only, no GPU);
    (0, 0, 0) below 48 tokens: the direct kernels (lmrs_debug_gemm_tile)."""
    tm, tn, w = C.c_int(), C.c_int(), C.c_int()
    _chk(lib().lmrs_debug_gemm_tile(n, o, n_tok, int(bool(q4)), C.byref(tm), C.byref(tn), C.byref(w)))
    return tm.value, tn.value, w.value


def processor_hd_transform(out_patches: np.ndarray, w_crop: int, h_crop: int, glb_gn: np.ndarray, sub_gn: np.ndarray) -> np.ndarray:
    """Host-only: the projector's input rows (processor.rs:240-254, 377-418, 480-484) -> float32 [n_embeds, 4096]."""
    op = np.ascontiguousarray(out_patches, np.float32).reshape(-1)
    g = np.ascontiguousarray(glb_gn, np.float32).reshape(-1); s_ = np.ascontiguousarray(sub_gn, np.float32).reshape(-1)
    ne = (h_crop * 12) * (w_crop * 12 + 1) + 12 * 13 + 1
    out = np.empty(ne * 4096, np.float32); n = C.c_uint32()
    _chk(lib().lmrs_processor_hd_transform(op.ctypes.data, op.size, 576 * 1024, w_crop, h_crop, g.ctypes.data, s_.ctypes.data, out.ctypes.data, C.byref(n)))
    return out[: n.value * 4096].reshape(n.value, 4096)


class PHI3VProcessor:
    """lmrs::processor::PHI3VProcessor (src/processor.rs:168-342): HD transform + two-layer projector, on the device."""

    def __init__(self, section: np.ndarray, device: int = 0):
        sec = np.ascontiguousarray(section, np.uint8)
        h = C.c_void_p(); used = C.c_size_t()
        _chk(lib().lmrs_processor_create(sec.ctypes.data, sec.size, device, C.byref(h), C.byref(used)))
        self._h, self.bytes_consumed = h, used.value
        self.text_dim = int(np.frombuffer(sec[4:8].tobytes(), np.uint32)[0])

    def forward(self, out_patches: np.ndarray, new_shape: int, patch_side: int, w_crop: int, h_crop: int) -> np.ndarray:
        """-> float32 [num_embeds, text_dim] (processor.rs:234-342)."""
        op = np.ascontiguousarray(out_patches, np.float32).reshape(-1)
        ne = (h_crop * patch_side) * (w_crop * patch_side + 1) + patch_side * (patch_side + 1) + 1
        out = np.empty(ne * self.text_dim, np.float32); n = C.c_uint32()
        _chk(lib().lmrs_processor_forward(self._h, op.ctypes.data, op.size, new_shape, patch_side, w_crop, h_crop, out.ctypes.data, C.byref(n)))
        return out[: n.value * self.text_dim].reshape(n.value, self.text_dim)

    def close(self):
        if self._h:
            lib().lmrs_processor_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tokenizer:
    """lmrs::tokenizer::Tokenizer (reference src/tokenizer.rs): Tokenizer(path or bytes), .bos / .eos, encode(), decode().  Host code."""

    def __init__(self, src):
        data = open(src, "rb").read() if isinstance(src, str) else bytes(src)
        buf = np.frombuffer(data, np.uint8)
        h = C.c_void_p()
        _chk(lib().lmrs_tokenizer_create(_p(buf), buf.size, C.byref(h)))
        self._h = h
        v, b, e = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _chk(lib().lmrs_tokenizer_info(h, C.byref(v), C.byref(b), C.byref(e)))
        self.vocab_size, self.bos, self.eos = v.value, b.value, e.value

    def encode(self, text: str, bos: bool, eos: bool, chat_format: bool, model_type: int) -> np.ndarray:
        raw = text.encode("utf-8")
        out = np.empty(len(raw) + 32, np.uint32); n = C.c_size_t()
        _chk(lib().lmrs_tokenizer_encode(self._h, raw, len(raw), int(bos), int(eos), int(chat_format), int(model_type), _p(out), out.size, C.byref(n)))
        return out[: n.value].copy()

    def decode(self, token: int) -> str:
        buf = C.create_string_buffer(256); n = C.c_size_t()
        _chk(lib().lmrs_tokenizer_decode(self._h, token, buf, 256, C.byref(n)))
        return buf.raw[: n.value].decode("utf-8")

    def close(self):
        if getattr(self, "_h", None):
            lib().lmrs_tokenizer_destroy(self._h); self._h = None

    __del__ = close


class Sampler:
    """lmrs::sampler::Sampler (reference src/sampler.rs): Sampler(vocab_size, temperature, top_p, seed).sample(logits).  Host code;
    `logits` (float32, C-contiguous) is modified in place when temperature != 0, as in the reference."""

    def __init__(self, vocab_size: int, temperature: float, top_p: float, seed: int):
        h = C.c_void_p()
        _chk(lib().lmrs_sampler_create(vocab_size, temperature, top_p, seed, C.byref(h)))
        self._h, self.vocab_size = h, vocab_size

    def sample(self, logits: np.ndarray) -> int:
        assert logits.dtype == np.float32 and logits.flags.c_contiguous and logits.size >= self.vocab_size
        nxt = C.c_uint32()
        _chk(lib().lmrs_sampler_sample(self._h, _p(logits), C.byref(nxt)))
        return nxt.value

    def sample_exps(self, exps: np.ndarray) -> int:
        """Sampler::sample from the softmax's exponentials on (lmrs_sampler_sample_exps): exps = exp(logits / temperature - max), turned into the probabilities in place"""
        assert exps.dtype == np.float32 and exps.flags.c_contiguous and exps.size >= self.vocab_size
        nxt = C.c_uint32()
        _chk(lib().lmrs_sampler_sample_exps(self._h, _p(exps), C.byref(nxt)))
        return nxt.value

    def topp_pairs(self, prob: np.ndarray, index: np.ndarray) -> int:
        """sample_topp from its sort on: the candidates (prob >= cutoff, in index order) were filtered elsewhere (lmrs_sampler_topp_pairs)"""
        pairs = np.zeros(prob.size, dtype=[("prob", np.float32), ("index", np.uint32)])
        pairs["prob"] = prob; pairs["index"] = index
        nxt = C.c_uint32()
        _chk(lib().lmrs_sampler_topp_pairs(self._h, _p(pairs) if prob.size else None, prob.size, C.byref(nxt)))
        return nxt.value

    def info(self):
        """(vocab_size, temperature, top_p, the random number every call draws: random_f32(seed), sampler.rs:119)"""
        v = C.c_uint32(); t = C.c_float(); p = C.c_float(); r = C.c_float()
        _chk(lib().lmrs_sampler_info(self._h, C.byref(v), C.byref(t), C.byref(p), C.byref(r)))
        return v.value, t.value, p.value, r.value

    def close(self):
        if getattr(self, "_h", None):
            lib().lmrs_sampler_destroy(self._h); self._h = None

    __del__ = close
