"""Measurement tooling that the round's artefacts depend on (no GPU): the profile summaries bench.py quotes are the ones of the
committed sources, the summary scripts reproduce the committed summaries from the committed raw statistics, (keyed by the hash of the kernel sources)."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rocprof_summary_is_reproducible_from_the_committed_statistics(tmp_path):
    """tools/rocprof_summary.py over the newest profiles/rN_kernel_stats.csv gives the per-step figure recorded in the same round's
    profiles/rN_rocprof_llama1b_q8.json."""
    ref_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_llama1b_q8.json")))[-1]
    stats = ref_path.replace("_rocprof_llama1b_q8.json", "_kernel_stats.csv")
    ref = json.load(open(ref_path))
    out = tmp_path / "s.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), stats, str(out), "llama-3.2-1b", "q8_0"], check=True, cwd=ROOT)
    got = json.load(open(out))
    for k in ("model", "qtype"):
        assert got[k] == ref[k]
    num = [k for k, v in ref.items() if isinstance(v, (int, float)) and not isinstance(v, bool)]
    assert num, "the summary carries numeric fields"
    for k in num:
        assert got[k] == pytest.approx(ref[k], rel=1e-9), k


def test_bench_reads_the_profile_summaries_when_the_source_hash_matches():
    """bench.py reports roofline.traffic / frac_rocprof only from summaries stamped with the hash of the sources it runs; whatever the
    state of the tree, the lookup must be consistent with the stamps."""
    import bench
    h = bench.kernel_source_hash()
    newest_tr = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_llama1b_q8.json")))[-1]
    tr = json.load(open(newest_tr))
    rp = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_llama1b_q8.json")))[-1]))
    t = bench.pmc_traffic("llama-3.2-1b", "q8_0")
    r = bench.rocprof_family_us("llama-3.2-1b", "q8_0")
    if tr["kernel_source_hash"] == h:
        assert t is not None and t > 1e7
    if tr["kernel_source_hash"] != h:
        assert t is None or t > 0          # (an older summary with a matching hash may exist; never a mismatching one)
    if rp["kernel_source_hash"] == h:
        assert r is not None and 300 < r < 700
    assert bench.pmc_traffic("llama-3.2-1b", "q4_0") is None or bench.pmc_traffic("llama-3.2-1b", "q4_0") > 0


def test_hot_kernel_resources_have_not_moved():
    """The per-kernel regression gate (no GPU): registers, scratch and the waves per SIMD they allow, of every hot kernel class of the BUILT
    library (read from the gfx950 code objects: tools/kernel_resources.py), against the committed table tests/golden/kernel_resources.json.
    A flag, launch-bound or source change that spills a class or drops its occupancy fails HERE, not in an artefact run on the GPU
    (round 5: __launch_bounds__(512, 4) spilled Gemma's 9216-wide w2 class, 1139 -> 1088 tok/s, found by accident).  A deliberate change
    is recorded with `python tools/kernel_resources.py --write` and shows up in the diff of the table."""
    import lmrs_amd
    from tools import kernel_resources as KR
    lmrs_amd.build()                                            # (no-op when the library is fresh)
    want = json.load(open(KR.TABLE))
    got = KR.collect(hot_only=True)
    assert set(got) == set(want), f"kernel classes added / removed: {sorted(set(got) ^ set(want))[:6]} - run tools/kernel_resources.py --write and review the diff"
    bad = []
    for name, w in want.items():
        g = got[name]
        if g["scratch"] != w["scratch"] or g["vgpr_spill"] != w["vgpr_spill"]:
            bad.append(f"{name}: scratch {w['scratch']} -> {g['scratch']} bytes per lane, spilled VGPRs {w['vgpr_spill']} -> {g['vgpr_spill']}")
        elif g["waves_per_simd"] != w["waves_per_simd"]:
            bad.append(f"{name}: waves per SIMD {w['waves_per_simd']} -> {g['waves_per_simd']} (VGPRs {w['vgpr']}+{w['agpr']} -> {g['vgpr']}+{g['agpr']})")
        elif abs(g["vgpr"] + g["agpr"] - w["vgpr"] - w["agpr"]) > 16:
            bad.append(f"{name}: VGPRs {w['vgpr']}+{w['agpr']} -> {g['vgpr']}+{g['agpr']}")
    assert not bad, "kernel resources moved:\n  " + "\n  ".join(bad[:12])
    # the decode path's classes never touch scratch (only the unquantised f32 GEMV, off the hot path, does)
    assert all(r["scratch"] == 0 for n, r in got.items() if "gemv_static_kernel" in n or "qkv_attn_kernel" in n)
