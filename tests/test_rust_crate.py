"""rust/lmrs-hip — the Rust host side of the C ABI (north star: "Rust host code calling HIP through a thin extern "C" layer").
No Rust toolchain exists in the build image, so the crate cannot be compiled here; what CAN be checked is that every `extern "C"`
declaration in it matches include/lmrs_hip.h: name, arity, each argument's type and the return type, and that the crate offers
the reference's public surface (the `pub fn` signatures of src/transformer.rs:134,316,659,672, src/vision.rs:99,244 and
src/processor.rs:169,234)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "lmrs-hip")

# C type -> Rust type, after whitespace normalisation
CMAP = {
    "int": "c_int", "void": "()", "size_t": "usize", "uint32_t": "u32", "float": "f32", "double": "f64",
    "const uint8_t*": "*const u8", "const int8_t*": "*const i8", "const uint32_t*": "*const u32", "const float*": "*const f32",
    "uint32_t*": "*mut u32", "float*": "*mut f32", "float**": "*mut *mut f32", "double*": "*mut f64", "size_t*": "*mut usize",
    "const char*": "*const c_char", "const void*": "*const c_void", "void*": "*mut c_void", "int*": "*mut c_int",
    "lmrs_ctx*": "*mut LmrsCtx", "const lmrs_ctx*": "*const LmrsCtx", "lmrs_ctx**": "*mut *mut LmrsCtx",
    "const lmrs_args*": "*const TransformerArgs",
    "lmrs_vision*": "*mut LmrsVision", "lmrs_vision**": "*mut *mut LmrsVision",
    "lmrs_processor*": "*mut LmrsProcessor", "lmrs_processor**": "*mut *mut LmrsProcessor",
    "uint64_t": "u64", "lmrs_sampler*": "*mut LmrsSampler", "const lmrs_sampler*": "*const LmrsSampler", "lmrs_sampler**": "*mut *mut LmrsSampler",
}


def c_prototypes():
    txt = open(os.path.join(ROOT, "include", "lmrs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(lmrs_\w+)\s*\(([^;{}]*?)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue

        def norm(t):
            t = re.sub(r"\s+", " ", t.strip())
            return re.sub(r"\s*\*\s*", "*", t)
        alist = []
        for a in [x for x in args.split(",") if x.strip() and x.strip() != "void"]:
            a = re.sub(r"\s+", " ", a.strip())
            mm = re.match(r"(.*?)(\w+)$", a)            # drop the parameter name
            typ = norm(mm.group(1)) if mm and mm.group(1).strip() else norm(a)
            alist.append(typ)
        protos[name] = (norm(ret), alist)
    return protos


def rust_externs():
    txt = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    txt = re.sub(r"//[^\n]*", " ", txt)
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', txt, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        alist = []
        for a in [x for x in re.split(r",(?![^<]*>)", args) if x.strip()]:
            alist.append(re.sub(r"\s+", " ", a.split(":", 1)[1].strip()))
        out[name] = (re.sub(r"\s+", " ", ret), alist)
    return out


def test_every_extern_declaration_matches_the_header():
    c, r = c_prototypes(), rust_externs()
    assert len(r) >= 21, sorted(r)
    for name, (rret, rargs) in r.items():
        assert name in c, f"{name} is declared in ffi.rs but not in include/lmrs_hip.h"
        cret, cargs = c[name]
        assert len(cargs) == len(rargs), f"{name}: {len(rargs)} arguments in Rust, {len(cargs)} in C ({cargs})"
        assert CMAP[cret] == rret, f"{name}: returns {rret} in Rust, {cret} in C"
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            assert ca in CMAP, f"{name}: no Rust mapping for C type '{ca}'"
            assert CMAP[ca] == ra, f"{name}: argument {i} is {ra} in Rust, {ca} in C"


def test_the_transformer_surface_of_the_reference_is_bound():
    """Everything a drop-in src/transformer.rs / vision.rs / processor.rs needs is declared."""
    r = rust_externs()
    for need in ("lmrs_create", "lmrs_destroy", "lmrs_get_args", "lmrs_forward", "lmrs_get_embeddings", "lmrs_fill_kv_cache", "lmrs_last_error",
                 "lmrs_vision_create", "lmrs_vision_forward", "lmrs_vision_destroy", "lmrs_processor_create", "lmrs_processor_forward",
                 "lmrs_processor_destroy", "lmrs_create_sharded", "lmrs_comm_unique_id", "lmrs_generate_greedy", "lmrs_forward_argmax",
                 "lmrs_sampler_create", "lmrs_sampler_destroy", "lmrs_sampler_sample", "lmrs_forward_sample"):
        assert need in r, need


def test_public_rust_api_has_the_reference_signatures():
    t = open(os.path.join(CRATE, "src", "transformer.rs")).read()
    v = open(os.path.join(CRATE, "src", "vision.rs")).read()
    p = open(os.path.join(CRATE, "src", "processor.rs")).read()
    sig = lambda s: re.sub(r"\s+", " ", s)
    # reference src/transformer.rs:134, :316, :659, :672
    for s in ("pub fn new(data: &'a Mmap) -> (Transformer<'a>, usize)", "pub fn forward(&mut self, token: u32, pos: u32) -> &mut [f32]",
              "pub fn get_embeddings(&self, tokens: &[u32]) -> Vec<f32>", "pub fn fill_kv_cache(&mut self, embeddings: &mut [f32], curr_pos: u32) -> u32",
              "impl<'a> Drop for Transformer<'a>", "pub args: TransformerArgs", "pub vocab_size: u32", "pub model_type: ModelType", "pub multimodal: bool"):
        assert s in sig(t), s
    # reference src/sampler.rs:19, :109 (+ the device draw)
    sm = open(os.path.join(CRATE, "src", "sampler.rs")).read()
    for s in ("pub fn new(vocab_size: u32, temperature: f32, top_p: f32, seed: u64) -> Sampler", "pub fn sample(&mut self, logits: &mut [f32]) -> u32",
              "impl Drop for Sampler", "pub fn forward_sample(&mut self, token: u32, pos: u32, sampler: &mut Sampler) -> u32"):
        assert s in sig(sm), s
    # reference src/vision.rs:99, :244 and the two public header fields chat.rs reads
    for s in ("pub fn new(data: &'a [u8]) -> (VisionTransformer<'a>, usize)", "pub fn forward(&mut self, pixel_values: &[f32], num_crops: u32) -> (Vec<f32>, u32)",
              "pub patch_size: u32", "pub image_size: u32", "impl<'a> Drop for VisionTransformer<'a>"):
        assert s in sig(v), s
    # reference src/processor.rs:169, :234
    for s in ("pub fn new(data: &'a [u8]) -> PHI3VProcessor<'a>",
              "pub fn forward(&self, out_patches: &[f32], new_shape: u32, patch_side: u32, w_crop: u32, h_crop: u32) -> Vec<f32>",
              "impl<'a> Drop for PHI3VProcessor<'a>"):
        assert s in sig(p), s


def test_transformer_args_layout_matches_lmrs_args():
    """#[repr(C)] TransformerArgs must list the fields of `lmrs_args` in the header's order with the same widths."""
    h = open(os.path.join(ROOT, "include", "lmrs_hip.h")).read()
    body = re.search(r"typedef struct lmrs_args\s*\{(.*?)\}\s*lmrs_args;", h, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    cfields = []
    for decl in [d.strip() for d in body.split(";") if d.strip()]:
        typ, names = decl.split(None, 1)
        for n in names.split(","):
            cfields.append((n.strip(), typ))
    t = open(os.path.join(CRATE, "src", "transformer.rs")).read()
    rbody = re.search(r"pub struct TransformerArgs\s*\{(.*?)\n\}", t, flags=re.S).group(1)
    rfields = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"(?:pub(?:\(crate\))?\s+)?(\w+)\s*:\s*([\w:]+)\s*,", rbody)]
    assert [n for n, _ in cfields] == [n for n, _ in rfields], (cfields, rfields)
    width = {"uint32_t": 4, "float": 4, "uint8_t": 1, "u32": 4, "f32": 4, "u8": 1, "QuantType": 1, "ModelType": 1, "bool": 1}
    for (n, ct), (_, rt) in zip(cfields, rfields):
        assert width[ct] == width[rt], f"{n}: {ct} vs {rt}"
