"""The C-ABI library: loads without a GPU, exports every symbol include/lmrs_hip.h declares,
and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "lmrs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lmrs_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_the_whole_header():
    import lmrs_amd
    lmrs_amd.build()
    lib = ctypes.CDLL(lmrs_amd.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} is declared in include/lmrs_hip.h but not exported"
    assert sorted(lmrs_amd.EXPORTS) == syms


def test_oracle_implements_the_same_surface():
    import oracle_lib as O
    lib = O.lib()
    for s in ["create", "destroy", "get_args", "forward", "forward_argmax", "get_embeddings", "fill_kv_cache", "generate_greedy",
              "last_error", "op_matmul_q8", "op_matmul_q4", "op_quantize", "op_quantize_q4", "op_rmsnorm", "op_softmax", "op_expf"]:
        assert hasattr(lib, "lmrs_ref_" + s)


def _no_gpu():
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) != 0 or n.value == 0
    except OSError:
        return True


def test_format_errors_are_reported_before_any_device_work():
    import lmrs_amd
    from tools import synth_lmrs as S
    img = S.build_image("tiny-llama", S.Q8_0, 1)
    bad = img.copy(); bad[1] = 0
    with pytest.raises(lmrs_amd.LmrsError, match="lm.rs format"):
        lmrs_amd.Transformer(bad)
    with pytest.raises(lmrs_amd.LmrsError, match="truncated"):
        lmrs_amd.Transformer(img[: img.size - 1])


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_silent_cpu_fallback():
    import lmrs_amd
    from tools import synth_lmrs as S
    with pytest.raises(lmrs_amd.LmrsError, match="no HIP device"):
        lmrs_amd.Transformer(S.build_image("tiny-llama", S.Q8_0, 1))
    with pytest.raises(lmrs_amd.LmrsError, match="no HIP device"):
        lmrs_amd.quantize(np.zeros(128, np.float32))


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "lm.rs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f in ("lmrs_device_math.h",), f"{f} mentions the oracle"
                assert "lmrs_ref_" not in txt, f


def test_cpp_host_mirror_compiles_and_links_against_the_abi(tmp_path):
    """lm.rs_amd/hostcpp: the C++ mirrors of Transformer / VisionTransformer / PHI3VProcessor build with plain g++ against
    include/lmrs_hip.h and link with the shared library (no GPU needed to link)."""
    import subprocess
    import lmrs_amd
    lmrs_amd.build()
    src = tmp_path / "use.cpp"
    src.write_text('#include "lm.rs_amd/hostcpp/vision.hpp"\n'
                   'int main(int argc, char**) {\n'
                   '    if (argc > 99) {\n'
                   '        auto [m, used] = lmrs_host::Transformer::create(nullptr, 0);\n'
                   '        auto [v, vused] = lmrs_host::VisionTransformer::create(nullptr, 0);\n'
                   '        auto p = lmrs_host::PHI3VProcessor::create(nullptr, 0);\n'
                   '        auto [f, ns] = v.forward({}, 1); (void)p.forward(f, ns, 12, 1, 1); (void)m.forward_argmax(0, 0); (void)used; (void)vused;\n'
                   '    }\n'
                   '    return 0;\n}\n')
    lib_dir = os.path.dirname(lmrs_amd.LIB_PATH)
    exe = tmp_path / "use"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", ROOT, str(src), "-L", lib_dir, "-llmrs_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)],
                   check=True, capture_output=True)
    for example in ("chat_greedy.cpp", "image_prefill.cpp", "chat.cpp"):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", os.path.join(ROOT, "lm.rs_amd", "hostcpp", example)], check=True, capture_output=True)


@pytest.mark.parametrize("w_crop,h_crop", [(1, 1), (2, 1), (1, 2), (2, 2), (3, 2)])
def test_processor_hd_transform_on_the_host(w_crop, h_crop):
    """The host half of lmrs_processor_forward (2x2 HD merge, row separators, glb_GN; processor.rs:240-254, 377-418, 480-484)
    against the reshape / transpose statement of the same transform in tests/numpy_ref.py - every crop grid, no GPU involved."""
    import lmrs_amd
    import numpy_ref as NR
    rng = np.random.default_rng(w_crop * 10 + h_crop)
    feats = rng.standard_normal((1 + w_crop * h_crop, 576, 1024)).astype(np.float32)
    glb = rng.standard_normal(4096).astype(np.float32); sub = rng.standard_normal(4096).astype(np.float32)
    got = lmrs_amd.processor_hd_transform(feats, w_crop, h_crop, glb, sub)
    ref = np.concatenate([NR.hd_transform(feats[1:], h_crop, w_crop, sub), glb.reshape(1, -1), NR.hd_transform(feats[:1], 1, 1, sub)])
    assert got.shape == ref.shape == ((h_crop * 12) * (w_crop * 12 + 1) + 12 * 13 + 1, 4096)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("model_type,theta,hs", [(0, 10000.0, 256), (0, 10000.0, 128), (1, 500000.0, 64), (1, 500000.0, 128), (2, 10000.0, 96), (2, 10000.0, 64)])
def test_rope_table_terms_on_the_host(model_type, theta, hs):
    """The host function behind lmrs_create's RoPE table (transformer.rs:446-477: frequency, Llama-3 wavelength scaling with its three
    regimes, Phi's LongRoPE short factors and magnitude) against the numpy transcription's rope(), which shares no code with it: every
    pair index, positions across the whole 8192-position cache - bit for bit, no GPU involved."""
    import types
    import lmrs_amd
    import numpy_ref as NR
    ref = types.SimpleNamespace(theta=np.float32(theta), hs=hs, model_type=model_type)
    for pos in list(range(0, 40)) + [63, 64, 127, 255, 256, 1000, 4095, 4096, 8191]:
        for j in range(hs // 2):
            c, s_ = lmrs_amd.rope_terms(model_type, theta, hs, pos, j)
            rc, rs = NR.NumpyModel.rope(ref, pos, j)
            assert np.float32(c).view(np.uint32) == np.float32(rc).view(np.uint32) and np.float32(s_).view(np.uint32) == np.float32(rs).view(np.uint32), (pos, j, c, rc, s_, rs)


def test_gemm_tile_choice_for_the_baseline_shapes():
    """The tile the batched int8-MFMA GEMM takes per launch (DESIGN.md section 4.1's table; host arithmetic of the library's cost model,
    lmrs_debug_gemm_tile).  Pinned so that a change of the model shows up as a change of this table."""
    import lmrs_amd
    t = lmrs_amd.gemm_tile
    # Llama-3.2-1B: w1/w3 (16384 rows, K 2048), qkv (3072), wo (2048, K 2048), w2 (2048, K 8192)
    assert t(2048, 16384, 512) == (256, 128, 8) and t(2048, 16384, 256) == (128, 128, 8)
    assert t(2048, 3072, 512) == (96, 64, 4) and t(2048, 3072, 256) == (64, 64, 8) and t(2048, 3072, 128) == (64, 32, 4)
    assert t(2048, 2048, 512) == (64, 64, 8) and t(8192, 2048, 512) == (64, 64, 8)
    assert t(2048, 2048, 256) == (64, 32, 4) and t(8192, 2048, 256) == (64, 32, 4)
    assert t(2048, 2048, 128) == (32, 32, 4) and t(8192, 2048, 128) == (32, 32, 4)
    # the 3072-wide models: wo / w2 at 512 tokens in one round of 128 x 64 tiles; Phi-3.5's 320 embeddings fit one round of 64 x 64
    assert t(3072, 3072, 512) == (128, 64, 8) and t(8192, 3072, 512) == (128, 64, 8) and t(8192, 3072, 320) == (64, 64, 8)
    assert t(3072, 16384, 512) == (256, 128, 8)
    # CLIP tower, 2 crops x 577 tokens: qkv, out_proj, fc1, fc2
    assert t(1024, 3072, 1154) == (128, 128, 8) and t(1024, 1024, 1154) == (128, 64, 8)
    assert t(1024, 4096, 1154) == (256, 128, 8) and t(4096, 1024, 1154) == (128, 64, 8)
    # Gemma-2-2B Q4_0: under-filled wo / w2 on 64 x 32, the gate / up pairs on 256 x 128
    assert t(2048, 2304, 256, q4=True) == (64, 32, 4) and t(9216, 2304, 256, q4=True) == (64, 32, 4) and t(2304, 18432, 256, q4=True) == (256, 128, 8)
    # ... round 6: at 512 tokens everything of Gemma-2-2B on 64 x 64 paired-group tiles (288 tiles of 256 x 128 would take two turns; the narrow
    # projections' 128 x 128 tiles filled a quarter of the chip); Llama-3.2-1B Q4_0 keeps the big ring tiles for w1/w3 (exactly one turn)
    assert t(9216, 2304, 512, q4=True) == (64, 64, 4) and t(2304, 18432, 512, q4=True) == (64, 64, 4) and t(2304, 4096, 512, q4=True) == (64, 64, 4)
    assert t(2048, 16384, 512, q4=True) == (256, 128, 8) and t(2048, 16384, 256, q4=True) == (128, 128, 8) and t(8192, 2048, 512, q4=True) == (64, 64, 4)
    assert t(2048, 2048, 47) == (0, 0, 0)                                   # below 48 tokens: the direct kernels
    with pytest.raises(Exception): t(2000, 2048, 64)

