#!/usr/bin/env python3
"""Pin the oracle in one command (needs cargo; the build container and the GPU box have none):

    python oracle/ref_harness/dump_all.py [--reference /path/to/lm.rs]

Builds oracle/ref_harness against the UNMODIFIED reference checkout (RUSTFLAGS="-C target-cpu=native", release, as the reference's
README.md:77 asks), runs it on every committed golden file tests/golden/*.lmrs with the prompt the golden vectors were made with,
and writes oracle/_ref/<fixture>.{tokens.u32, logits.f32, fill.f32, fill_logits.f32}.  tests/test_oracle.py::
test_against_reference_dump then compares the C oracle with those dumps bit for bit (and skips, loudly, while they do not exist)."""
import argparse
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tools import synth_lmrs as S  # noqa: E402

FIXTURES = [("tiny_llama_q8", "tiny-llama", 7), ("tiny_llama_q4", "tiny-llama", 7), ("tiny_gemma_q8", "tiny-gemma", 8),
            ("tiny_gemma_q4", "tiny-gemma", 8), ("tiny_phi_q8", "tiny-phi", 9), ("tiny_llama_f32", "tiny-llama", 7)]
N_PROMPT, N_NEW = 5, 12          # as tests/golden/make_golden.py


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("LMRS_REFERENCE_DIR", "/root/reference"))
    args = ap.parse_args()
    if shutil.which("cargo") is None:
        sys.exit("cargo not found: this recipe needs a Rust toolchain (none in the build container / on the GPU box)")
    out_dir = os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(out_dir, exist_ok=True)
    # a build copy of the crate with the reference path filled in (the committed Cargo.toml names the container's default)
    build = os.path.join(out_dir, "ref_harness_build")
    shutil.rmtree(build, ignore_errors=True)
    shutil.copytree(HERE, build, ignore=shutil.ignore_patterns("target", "__pycache__"))
    toml = open(os.path.join(build, "Cargo.toml")).read().replace('path = "/root/reference"', f'path = "{os.path.abspath(args.reference)}"')
    open(os.path.join(build, "Cargo.toml"), "w").write(toml)
    env = dict(os.environ, RUSTFLAGS="-C target-cpu=native")
    subprocess.run(["cargo", "build", "--release"], cwd=build, env=env, check=True)
    exe = os.path.join(build, "target", "release", "lmrs-ref-dump")
    for name, cfg, seed in FIXTURES:
        prompt = S.prompt_tokens(cfg, N_PROMPT, seed).astype("<u4")
        pfile = os.path.join(out_dir, name + ".prompt.u32")
        prompt.tofile(pfile)
        subprocess.run([exe, os.path.join(ROOT, "tests", "golden", name + ".lmrs"), pfile, str(N_NEW), os.path.join(out_dir, name)], check=True)
    print("wrote", out_dir, "- now run: python -m pytest tests/test_oracle.py -k reference_dump")


if __name__ == "__main__":
    main()
