//! `Transformer` (reference src/transformer.rs:127-131): `new` :134-314, `forward` :316-384, `get_embeddings` :659-669,
//! `fill_kv_cache` :672-684, `Drop` :688-712 - each a call into liblmrs_hip.so.  Weights, KV cache and activations live in HBM;
//! the mmap is only read while `new` uploads it.
use std::marker::PhantomData;
use std::ptr;

use memmap2::Mmap;

use crate::ffi::{self, check, LmrsCtx};

#[derive(Debug, Copy, Clone, PartialEq)]
#[repr(u8)]
pub enum ModelType {
    GEMMA = 0,
    LLAMA = 1,
    PHI = 2,
}

#[derive(Debug, Copy, Clone, PartialEq)]
#[repr(u8)]
pub enum QuantType {
    None = 0,
    Q8_0 = 1,
    Q4_0 = 2,
}

/// == `lmrs_args` of include/lmrs_hip.h (field for field; the file's packed header is decoded on the C side).
/// Public fields: the ones the reference exposes (`vocab_size`, `model_type`, `multimodal`, transformer.rs:66-73).
#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct TransformerArgs {
    pub(crate) dim: u32,
    pub(crate) hidden_dim: u32,
    pub(crate) n_layers: u32,
    pub(crate) n_heads: u32,
    pub(crate) head_size: u32,
    pub(crate) n_kv_heads: u32,
    pub vocab_size: u32,
    pub(crate) seq_len: u32,
    pub(crate) rms_norm_eps: f32,
    pub(crate) rope_theta: f32,
    pub(crate) q_type: QuantType,
    pub model_type: ModelType,
    pub multimodal: bool,
    _pad: u8,
    pub(crate) group_size: u32,
}

pub struct Transformer<'a> {
    pub args: TransformerArgs,
    ctx: *mut LmrsCtx,
    _data: PhantomData<&'a Mmap>,
}

impl<'a> Transformer<'a> {
    /// transformer.rs:134 - returns the model and the number of bytes of `data` it covers (the offset of the vision
    /// section in a multimodal file).
    pub fn new(data: &'a Mmap) -> (Transformer<'a>, usize) {
        let mut ctx: *mut LmrsCtx = ptr::null_mut();
        let mut used: usize = 0;
        check(unsafe { ffi::lmrs_create(data.as_ptr(), data.len(), ffi::device(), &mut ctx, &mut used) });
        let args = unsafe { *ffi::lmrs_get_args(ctx) };
        (Transformer { args, ctx, _data: PhantomData }, used)
    }

    /// One process per GPU, rows of every weight matrix split over `world` GPUs (RCCL all-gathers over xGMI inside the
    /// step).  `unique_id`: the 128 bytes `comm_unique_id()` returned on rank 0, distributed by the launcher.
    pub fn new_sharded(data: &'a Mmap, rank: i32, world: i32, unique_id: &[u8; 128]) -> (Transformer<'a>, usize) {
        let mut ctx: *mut LmrsCtx = ptr::null_mut();
        let mut used: usize = 0;
        check(unsafe {
            ffi::lmrs_create_sharded(data.as_ptr(), data.len(), ffi::device(), rank, world, unique_id.as_ptr() as *const _, &mut ctx, &mut used)
        });
        let args = unsafe { *ffi::lmrs_get_args(ctx) };
        (Transformer { args, ctx, _data: PhantomData }, used)
    }

    /// transformer.rs:316 - the logits live in pinned host memory owned by the context and stay valid (and mutable: the
    /// reference's sampler scales them in place, sampler.rs:115-117) until the next call on `self`.
    pub fn forward(&mut self, token: u32, pos: u32) -> &mut [f32] {
        let mut p: *mut f32 = ptr::null_mut();
        check(unsafe { ffi::lmrs_forward(self.ctx, token, pos, &mut p) });
        unsafe { std::slice::from_raw_parts_mut(p, self.args.vocab_size as usize) }
    }

    /// `forward` followed by `Sampler::sample_argmax` (sampler.rs:29-41) on the device: no logits leave HBM.
    pub fn forward_argmax(&mut self, token: u32, pos: u32) -> u32 {
        let mut next: u32 = 0;
        check(unsafe { ffi::lmrs_forward_argmax(self.ctx, token, pos, &mut next) });
        next
    }

    /// transformer.rs:659
    pub fn get_embeddings(&self, tokens: &[u32]) -> Vec<f32> {
        let mut out = vec![0.0f32; tokens.len() * self.args.dim as usize];
        check(unsafe { ffi::lmrs_get_embeddings(self.ctx, tokens.as_ptr(), tokens.len(), out.as_mut_ptr()) });
        out
    }

    /// transformer.rs:672 - `embeddings` is updated in place exactly as the reference mutates its argument.
    pub fn fill_kv_cache(&mut self, embeddings: &mut [f32], curr_pos: u32) -> u32 {
        let n = embeddings.len() as u32 / self.args.dim;
        let mut new_pos: u32 = 0;
        check(unsafe { ffi::lmrs_fill_kv_cache(self.ctx, embeddings.as_mut_ptr(), n, curr_pos, &mut new_pos) });
        new_pos
    }

    pub(crate) fn ctx(&mut self) -> *mut LmrsCtx {
        self.ctx
    }

    /// The token loop of src/bin/chat.rs:188-222 at temperature 0, device-resident (one host sync per call): feeds `prompt`
    /// from position `start_pos`, then `n_new - 1` further steps feeding back the argmax; returns the `n_new` generated ids.
    pub fn generate_greedy(&mut self, prompt: &[u32], n_new: u32, start_pos: u32) -> Vec<u32> {
        let mut out = vec![0u32; n_new as usize];
        check(unsafe {
            ffi::lmrs_generate_greedy(self.ctx, prompt.as_ptr(), prompt.len(), n_new, start_pos, out.as_mut_ptr(), ptr::null_mut())
        });
        out
    }
}

/// The communicator id rank 0 makes for `new_sharded`.
pub fn comm_unique_id() -> [u8; 128] {
    let mut id = [0u8; 128];
    check(unsafe { ffi::lmrs_comm_unique_id(id.as_mut_ptr() as *mut _) });
    id
}

impl<'a> Drop for Transformer<'a> {
    fn drop(&mut self) {
        unsafe { ffi::lmrs_destroy(self.ctx) }
    }
}

// one in-flight call per context (`&mut self`), contexts are independent
unsafe impl<'a> Send for Transformer<'a> {}
