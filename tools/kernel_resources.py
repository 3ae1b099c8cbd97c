"""Per-kernel resources of the BUILT library, read from the gfx950 code objects inside lm.rs_amd/csrc/*.o (no GPU, no recompile):
registers, scratch (= spills), static LDS and the waves per SIMD the registers allow.

    python tools/kernel_resources.py            # table of the hot kernel classes
    python tools/kernel_resources.py --all      # every kernel
    python tools/kernel_resources.py --write    # rewrite tests/golden/kernel_resources.json (the table tests/test_tooling.py pins)

Why it exists (round 5's lesson): a flag or launch-bound change that spills one class - Gemma's 9216-wide w2 under __launch_bounds__(512, 4),
1139 -> 1088 tok/s - or costs another 30 registers went unnoticed until an artefact run on the GPU.  The gate is a CPU test now."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lm.rs_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
OBJECTS = ["lmrs_kernels.o", "lmrs_api.o", "lmrs_vision_att.o"]
TABLE = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
# the kernel classes of the decode step, the batched prefill and the image tower (everything a bench line or a profile quotes)
HOT = re.compile(r"^(?:void )?lmrs::(gemv_static_kernel|qkv_attn_kernel|gemm_q8_dma_kernel|gemm_q4_pair_kernel|attention_split_values_kernel|attention_split_scores_kernel|"
                 r"attention_kernel|att_scores_kernel|att_scores_wide_kernel|att_softmax_kernel|att_values_kernel|vis_att_\w+|rows_\w+_kernel|sample_\w+_kernel)\b")


def _kernels_of(obj):
    """[(mangled name, metadata dict)] of the gfx950 code object bundled in a host object file."""
    import yaml
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    m = re.search(r"^\s*---\n(.*?)^\.\.\.", notes, re.S | re.M)
    meta = yaml.safe_load(m.group(1))
    return [(k[".name"], k) for k in meta["amdhsa.kernels"]]


def waves_per_simd(vgpr, agpr):
    alloc = -(-(vgpr + agpr) // 8) * 8                      # the hardware allocates VGPRs + AGPRs in granules of 8 out of 512 per SIMD lane
    return min(8, 512 // max(alloc, 8))


def collect(hot_only=True):
    rows = {}
    for o in OBJECTS:
        path = os.path.join(CSRC, o)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: build the library first (python -c 'import lmrs_amd; lmrs_amd.build()')")
        ks = _kernels_of(path)
        names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in ks), capture_output=True, text=True, check=True).stdout.splitlines()
        for (_, k), name in zip(ks, names):
            name = re.sub(r"\(.*$", "", name).replace("void ", "")          # template arguments identify the class; the parameter list does not
            if hot_only and not HOT.match(name):
                continue
            rows[name] = {"vgpr": k[".vgpr_count"], "agpr": k.get(".agpr_count", 0), "sgpr": k[".sgpr_count"],
                          "scratch": k[".private_segment_fixed_size"], "vgpr_spill": k[".vgpr_spill_count"], "sgpr_spill": k[".sgpr_spill_count"],
                          "lds_static": k[".group_segment_fixed_size"], "waves_per_simd": waves_per_simd(k[".vgpr_count"], k.get(".agpr_count", 0))}
    return rows


def main():
    rows = collect(hot_only="--all" not in sys.argv)
    if "--write" in sys.argv:
        with open(TABLE, "w") as f:
            json.dump(dict(sorted(rows.items())), f, indent=0, sort_keys=True)
            f.write("\n")
        print(f"{len(rows)} kernel classes -> {os.path.relpath(TABLE, ROOT)}")
        return
    for name, r in sorted(rows.items()):
        print(f'{name[:150]:150s} V {r["vgpr"]:3d} A {r["agpr"]:3d} S {r["sgpr"]:3d} scratch {r["scratch"]:4d} spills {r["vgpr_spill"]:3d}/{r["sgpr_spill"]:3d} LDS {r["lds_static"]:6d} waves/SIMD {r["waves_per_simd"]}')


if __name__ == "__main__":
    main()
