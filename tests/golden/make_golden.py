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
3. `tiny_phi_vision_q8.json` — sizes and SHA-256 of a multimodal file (text + CLIP tower + projector sections) written by the
   reference exporter with --vision-config: pins the vision / processor section layout tools/synth_vision.py restates.
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


MM_NAME, MM_TEXT_CFG, MM_SEED = "tiny_phi_vision_q8", "tiny-phi", 9
MM_VSEED, MM_PSEED = 31, 32


def mm_vision_cfg():
    from tools import synth_vision as V
    return V.VisionCfg(dim=128, hidden_dim=512, n_layers=2, n_heads=2, head_size=64)


def mm_expected_image(q_type=S.Q8_0):
    """What tools/synth_lmrs.py + tools/synth_vision.py write for the multimodal fixture: text image with the multimodal flag set,
    then the vision section, then the processor section."""
    from tools import synth_vision as V
    cfg = S.CONFIGS[MM_TEXT_CFG]
    text = S.build_image(cfg, q_type, MM_SEED).copy()
    text[54] = 1                                                      # header: multimodal flag (export.py:80)
    vcfg = mm_vision_cfg()
    vis = V.build_vision_section(vcfg, MM_VSEED, q_type=q_type)
    proc = V.build_processor_section(4 * vcfg.dim, cfg.dim, MM_PSEED, q_type=q_type)
    return text, vis, proc


MM_VARIANTS = {S.Q8_0: "tiny_phi_vision_q8", S.Q4_0: "tiny_phi_vision_q4", S.Q_NONE: "tiny_phi_vision_f32"}


def export_multimodal(q_type=S.Q8_0):
    """The reference exporter on a tiny Phi + CLIP + projector checkpoint (--vision-config): pins the layout of the vision and
    processor sections (export.py:126-170) that tools/synth_vision.py restates.  Only sizes and the SHA-256 are committed."""
    import hashlib
    import torch
    from safetensors.torch import save_file
    from tools import synth_vision as V
    cfg = S.CONFIGS[MM_TEXT_CFG]
    vcfg = mm_vision_cfg()
    assert vcfg.hidden_dim == 4 * vcfg.dim
    name = MM_VARIANTS[q_type]
    base = os.path.join(HERE, name)
    with tempfile.TemporaryDirectory() as td:
        sd = dict(S.hf_state_dict(cfg, MM_SEED))
        sd.update(V.hf_vision_state_dict(vcfg, 4 * vcfg.dim, cfg.dim, MM_VSEED, MM_PSEED))
        save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, os.path.join(td, "model.safetensors"))
        with open(os.path.join(td, "config.json"), "w") as f:
            json.dump(S.hf_config(cfg), f)
        with open(os.path.join(td, "vision.json"), "w") as f:
            json.dump(V.hf_vision_config(vcfg), f)
        out = os.path.join(td, "mm")
        subprocess.run([sys.executable, os.path.join(REF, "export.py"), "--files", os.path.join(td, "model.safetensors"), "--config",
                        os.path.join(td, "config.json"), "--vision-config", os.path.join(td, "vision.json"), "--save-path", out,
                        "--type", "PHI"] + (["--quantize", "--quantize-type", str(q_type)] if q_type != S.Q_NONE else []),
                       check=True, cwd=REF, stdout=subprocess.DEVNULL)
        ref = np.fromfile(out + ".lmrs", np.uint8)
    text, vis, proc = mm_expected_image(q_type)
    mine = np.concatenate([text, vis, proc])
    same = ref.size == mine.size and bool((ref == mine).all())
    print(f"{name}: export.py wrote {ref.size} bytes (text {text.size} + vision {vis.size} + processor {proc.size}); synth writers identical: {same}")
    if not same:
        n = min(ref.size, mine.size); d = np.flatnonzero(ref[:n] != mine[:n])
        print("   first difference at byte", int(d[0]) if d.size else n)
    assert same
    with open(base + ".json", "w") as f:
        json.dump({"bytes": int(ref.size), "text_bytes": int(text.size), "vision_bytes": int(vis.size), "processor_bytes": int(proc.size),
                   "sha256": hashlib.sha256(ref.tobytes()).hexdigest()}, f, indent=1)


def main():
    for q in MM_VARIANTS:
        export_multimodal(q)
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
