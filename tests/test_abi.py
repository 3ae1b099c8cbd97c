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
