//! Dumps what the reference itself computes for one LMRS file and one prompt, so that oracle/lmrs_oracle.c (and through it the
//! HIP path) can be compared with it bit for bit.  Uses only the reference's public API, exactly as its own binaries do
//! (src/bin/chat.rs:62-65, 188-222):
//!
//!   lmrs-ref-dump <model.lmrs> <prompt.u32> <n_new> <out_prefix>
//!
//!   <out_prefix>.tokens.u32   the n_new greedy tokens after the prompt (Sampler at temperature 0 = sample_argmax, sampler.rs:29-41)
//!   <out_prefix>.logits.f32   [n_prompt + n_new - 1][vocab_size] logits of EVERY forward call (transformer.rs:316-384)
//!   <out_prefix>.fill.f32     the prompt's embeddings after fill_kv_cache(embeddings, 0) (transformer.rs:672-684), i.e. the
//!                             residual stream forward_layer(sl = n_prompt) leaves behind, on a second, fresh model
//!   <out_prefix>.fill_logits.f32   logits of forward(first generated token, n_prompt) after that batched fill
//!
//! Build as the reference's README asks: RUSTFLAGS="-C target-cpu=native" cargo build --release (README.md:77) - the `wide`
//! crate's f32x8::reduce_add order (rmsnorm, functional.rs:58) depends on the AVX path being compiled in.
use lmrs::sampler::Sampler;
use lmrs::transformer::Transformer;
use memmap2::Mmap;
use std::fs::File;
use std::io::Write;

fn read_u32s(path: &str) -> Vec<u32> {
    let bytes = std::fs::read(path).expect("cannot read the prompt file");
    assert!(bytes.len() % 4 == 0, "prompt file: not a whole number of u32");
    bytes.chunks_exact(4).map(|c| u32::from_le_bytes([c[0], c[1], c[2], c[3]])).collect()
}

fn write_f32s(path: &str, v: &[f32]) {
    let mut f = File::create(path).expect("cannot create output file");
    for x in v {
        f.write_all(&x.to_le_bytes()).unwrap();
    }
}

fn write_u32s(path: &str, v: &[u32]) {
    let mut f = File::create(path).expect("cannot create output file");
    for x in v {
        f.write_all(&x.to_le_bytes()).unwrap();
    }
}

fn main() {
    let a: Vec<String> = std::env::args().collect();
    assert!(a.len() == 5, "usage: lmrs-ref-dump <model.lmrs> <prompt.u32> <n_new> <out_prefix>");
    let prompt = read_u32s(&a[2]);
    let n_new: usize = a[3].parse().expect("n_new");
    let out = &a[4];
    assert!(!prompt.is_empty() && n_new >= 1);

    let file = File::open(&a[1]).expect("Error opening model file");
    let data = unsafe { Mmap::map(&file).expect("MMap failed") };

    // ---- token by token, as chat.rs feeds the prompt and then its own samples (chat.rs:188-222)
    let (mut model, _consumed) = Transformer::new(&data);
    let vocab = { model.args.vocab_size } as usize;
    let mut sampler = Sampler::new(vocab as u32, 0.0, 0.9, 0);
    let steps = prompt.len() + n_new - 1;
    let mut all_logits: Vec<f32> = Vec::with_capacity(steps * vocab);
    let mut generated: Vec<u32> = Vec::with_capacity(n_new);
    let mut token = prompt[0];
    for pos in 0..steps {
        let logits = model.forward(token, pos as u32);
        all_logits.extend_from_slice(logits);
        let next = sampler.sample(logits);
        if pos + 1 < prompt.len() {
            token = prompt[pos + 1];
        } else {
            generated.push(next);
            token = next;
        }
    }
    write_u32s(&format!("{}.tokens.u32", out), &generated);
    write_f32s(&format!("{}.logits.f32", out), &all_logits);

    // ---- the batched form: forward_layer over the whole prompt at once (fill_kv_cache), then one decode step
    let (mut model2, _c2) = Transformer::new(&data);
    let mut emb = model2.get_embeddings(&prompt);
    let new_pos = model2.fill_kv_cache(&mut emb, 0);
    assert!(new_pos as usize == prompt.len());
    write_f32s(&format!("{}.fill.f32", out), &emb);
    let lg = model2.forward(generated[0], new_pos);
    write_f32s(&format!("{}.fill_logits.f32", out), lg);
    println!("{}: {} forward calls, vocab {}, first tokens {:?}", a[1], steps, vocab, &generated[..generated.len().min(8)]);
}
