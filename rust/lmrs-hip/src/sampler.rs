//! `Sampler` (reference src/sampler.rs:10-17): `new` :19-27, `sample` :109-129 - the same results (the reference never advances its
//! seed, :119, and sorts stale candidates, :81: both kept), behind the library's handle so that `Transformer::forward_sample` can
//! draw ON THE DEVICE: temperature scaling, the in-place softmax with its sequential sum and `sample_mult` run in HBM and only the
//! token id comes back.  `sample` on host logits is the library's host restatement of the same code.
use std::ptr;

use crate::ffi::{self, check, LmrsSampler};
use crate::transformer::Transformer;

pub struct Sampler {
    pub(crate) handle: *mut LmrsSampler,
    vocab_size: u32,
}

impl Sampler {
    /// sampler.rs:19
    pub fn new(vocab_size: u32, temperature: f32, top_p: f32, seed: u64) -> Sampler {
        let mut handle: *mut LmrsSampler = ptr::null_mut();
        check(unsafe { ffi::lmrs_sampler_create(vocab_size, temperature, top_p, seed, &mut handle) });
        Sampler { handle, vocab_size }
    }

    /// sampler.rs:109 - `logits` is scaled and softmax-ed in place exactly as the reference mutates its argument.
    pub fn sample(&mut self, logits: &mut [f32]) -> u32 {
        // the C side reads and writes vocab_size floats: a shorter slice must fail here, as the reference's indexing would (sampler.rs:114)
        assert!(logits.len() >= self.vocab_size as usize, "logits shorter than the sampler's vocabulary");
        let mut next: u32 = 0;
        check(unsafe { ffi::lmrs_sampler_sample(self.handle, logits.as_mut_ptr(), &mut next) });
        next
    }
}

impl Drop for Sampler {
    fn drop(&mut self) {
        unsafe { ffi::lmrs_sampler_destroy(self.handle) };
    }
}

impl<'a> Transformer<'a> {
    /// `forward` (transformer.rs:316) followed by `sampler.sample(logits)` (sampler.rs:109) with the logits staying in HBM: the
    /// token loop of src/bin/chat.rs:188-222 becomes `token = model.forward_sample(token, pos, &mut sampler)`.  Same token as the two
    /// calls in every case (top_p inside (0, 1) falls back to them inside the library).
    pub fn forward_sample(&mut self, token: u32, pos: u32, sampler: &mut Sampler) -> u32 {
        let mut next: u32 = 0;
        check(unsafe { ffi::lmrs_forward_sample(self.ctx(), token, pos, sampler.handle, &mut next) });
        next
    }
}
