#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.  Run in the BUILD container only
(it needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

1. `<name>.lmrs` — written by the REFERENCE's own exporter (/root/reference/export.py, unmodified,
   run as a subprocess) from float32 safetensors holding the tensors tools/synth_lmrs.py draws for
   (cfg, seed).  These pin the LMRS v4 layout and the reference weight quantisers
   (utils/quantization.py): tests/test_format.py requires tools/synth_lmrs.build_image() to
   reproduce them byte for byte and the oracle / HIP loaders to parse them.
2. `<name>.tokens.npy`, `<name>.logits.npy` — greedy token IDs and last-step logits produced by the
   CPU ORACLE (oracle/liblmrs_oracle.so) on those files.  The forward arithmetic of the reference
   cannot be executed here (Rust, no toolchain) so these are regression vectors for the oracle and
   the HIP path, NOT reference outputs: forward parity remains "unpinned" (see DESIGN.md).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tools import synth_lmrs as S  # noqa: E402

REF = "/root/reference"
FIXTURES = [  # (fixture name, cfg, q_type, seed)
    ("tiny_llama_q8", "tiny-llama", S.Q8_0, 7),
    ("tiny_llama_q4", "tiny-llama", S.Q4_0, 7),
    ("tiny_gemma_q8", "tiny-gemma", S.Q8_0, 8),
    ("tiny_gemma_q4", "tiny-gemma", S.Q4_0, 8),
    ("tiny_phi_q8", "tiny-phi", S.Q8_0, 9),
    ("tiny_llama_f32", "tiny-llama", S.Q_NONE, 7),
]
N_PROMPT, N_NEW = 5, 12


def export_with_reference(cfg, q_type, seed, out_base):
    import torch
    from safetensors.torch import save_file
    with tempfile.TemporaryDirectory() as td:
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in S.hf_state_dict(cfg, seed).items()}
        save_file(sd, os.path.join(td, "model.safetensors"))
        with open(os.path.join(td, "config.json"), "w") as f:
            json.dump(S.hf_config(cfg), f)
        cmd = [sys.executable, os.path.join(REF, "export.py"), "--files", os.path.join(td, "model.safetensors"),
               "--config", os.path.join(td, "config.json"), "--save-path", out_base,
               "--type", ["GEMMA", "LLAMA", "PHI"][cfg.model_type]]
        if q_type != S.Q_NONE:
            cmd += ["--quantize", "--quantize-type", str(q_type)]
        subprocess.run(cmd, check=True, cwd=REF, stdout=subprocess.DEVNULL)


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    for name, cfg_name, q_type, seed in FIXTURES:
        cfg = S.CONFIGS[cfg_name]
        base = os.path.join(HERE, name)
        export_with_reference(cfg, q_type, seed, base)
        ref_bytes = np.fromfile(base + ".lmrs", np.uint8)
        mine = S.build_image(cfg, q_type, seed)
        same = ref_bytes.size == mine.size and bool((ref_bytes == mine).all())
        print(f"{name}: export.py wrote {ref_bytes.size} bytes; synth writer identical: {same}")
        assert same
        orc = Oracle(ref_bytes)
        prompt = S.prompt_tokens(cfg, N_PROMPT, seed)
        toks = orc.generate_greedy(prompt, N_NEW)
        orc2 = Oracle(ref_bytes)
        logits = None
        seq = list(prompt) + list(toks[:-1])
        for pos, t in enumerate(seq):
            logits = orc2.forward(int(t), pos)
        np.save(base + ".tokens.npy", np.asarray(toks, np.uint32))
        np.save(base + ".logits.npy", logits.copy())
        print("   tokens:", list(map(int, toks)))


if __name__ == "__main__":
    main()
