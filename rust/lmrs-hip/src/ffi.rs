//! The C ABI of liblmrs_hip.so, one declaration per entry point of include/lmrs_hip.h that the Rust side uses.
//! tests/test_rust_crate.py parses this block and compares it with the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

use crate::transformer::TransformerArgs;

#[repr(C)]
pub struct LmrsCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct LmrsVision {
    _private: [u8; 0],
}
#[repr(C)]
pub struct LmrsProcessor {
    _private: [u8; 0],
}
#[repr(C)]
pub struct LmrsSampler {
    _private: [u8; 0],
}

extern "C" {
    pub fn lmrs_create(file: *const u8, len: usize, device: c_int, out: *mut *mut LmrsCtx, bytes_consumed: *mut usize) -> c_int;
    pub fn lmrs_create_sharded(file: *const u8, len: usize, device: c_int, rank: c_int, world: c_int, nccl_unique_id: *const c_void,
                               out: *mut *mut LmrsCtx, bytes_consumed: *mut usize) -> c_int;
    pub fn lmrs_comm_unique_id(out128: *mut c_void) -> c_int;
    pub fn lmrs_destroy(ctx: *mut LmrsCtx);
    pub fn lmrs_get_args(ctx: *const LmrsCtx) -> *const TransformerArgs;
    pub fn lmrs_forward(ctx: *mut LmrsCtx, token: u32, pos: u32, logits: *mut *mut f32) -> c_int;
    pub fn lmrs_forward_argmax(ctx: *mut LmrsCtx, token: u32, pos: u32, next: *mut u32) -> c_int;
    pub fn lmrs_get_embeddings(ctx: *const LmrsCtx, tokens: *const u32, n: usize, out: *mut f32) -> c_int;
    pub fn lmrs_fill_kv_cache(ctx: *mut LmrsCtx, embeddings: *mut f32, n: u32, curr_pos: u32, new_pos: *mut u32) -> c_int;
    pub fn lmrs_generate_greedy(ctx: *mut LmrsCtx, prompt: *const u32, n_prompt: usize, n_new: u32, start_pos: u32,
                                out_tokens: *mut u32, seconds: *mut f64) -> c_int;
    pub fn lmrs_last_error() -> *const c_char;

    pub fn lmrs_vision_create(section: *const u8, len: usize, device: c_int, out: *mut *mut LmrsVision, bytes_consumed: *mut usize) -> c_int;
    pub fn lmrs_vision_destroy(v: *mut LmrsVision);
    pub fn lmrs_vision_forward(v: *mut LmrsVision, pixel_values: *const f32, num_crops: u32, out: *mut f32, new_shape: *mut u32) -> c_int;

    pub fn lmrs_processor_create(section: *const u8, len: usize, device: c_int, out: *mut *mut LmrsProcessor, bytes_consumed: *mut usize) -> c_int;
    pub fn lmrs_processor_destroy(p: *mut LmrsProcessor);
    pub fn lmrs_processor_forward(p: *mut LmrsProcessor, out_patches: *const f32, total_floats: u32, new_shape: u32, patch_side: u32,
                                  w_crop: u32, h_crop: u32, out: *mut f32, n_embeds: *mut u32) -> c_int;

    pub fn lmrs_sampler_create(vocab_size: u32, temperature: f32, top_p: f32, seed: u64, out: *mut *mut LmrsSampler) -> c_int;
    pub fn lmrs_sampler_destroy(s: *mut LmrsSampler);
    pub fn lmrs_sampler_sample(s: *mut LmrsSampler, logits: *mut f32, next: *mut u32) -> c_int;
    pub fn lmrs_forward_sample(ctx: *mut LmrsCtx, token: u32, pos: u32, sampler: *mut LmrsSampler, next: *mut u32) -> c_int;
}

/// The reference panics (`assert!` / `expect`); the C ABI returns a status and a message.  Same behaviour for the caller.
pub(crate) fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(lmrs_last_error()) }.to_string_lossy().into_owned();
        panic!("lmrs-hip: {}", msg);
    }
}

/// Device index for every `new`: LMRS_HIP_DEVICE (default 0).  One process per GPU: a multi-GPU launcher sets it per rank.
pub(crate) fn device() -> c_int {
    std::env::var("LMRS_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0)
}
