//! The device half of `PHI3VProcessor` (reference src/processor.rs:163-166): `new` :169-232 and `forward` :234-342 - the HD
//! transform, the separators and the two-layer projector MLP.  `process` (:344-375: resize, pad, normalise, patchify) is host
//! image code built on the `image` crate and stays lm.rs's own; INTEGRATION.md shows the two-line delegation.
use std::marker::PhantomData;
use std::ptr;

use crate::ffi::{self, check, LmrsProcessor};
use crate::transformer::QuantType;

/// The 13 header bytes of the processor section (processor.rs:147-153).
#[derive(Debug, Copy, Clone)]
pub struct ProcessorArgs {
    pub(crate) hidden_dim: u32,
    pub(crate) text_dim: u32,
    pub(crate) q_type: QuantType,
    pub(crate) group_size: u32,
}

fn rd_u32(d: &[u8], off: usize) -> u32 {
    u32::from_le_bytes([d[off], d[off + 1], d[off + 2], d[off + 3]])
}

pub struct PHI3VProcessor<'a> {
    args: ProcessorArgs,
    h: *mut LmrsProcessor,
    _data: PhantomData<&'a [u8]>,
}

impl<'a> PHI3VProcessor<'a> {
    /// processor.rs:169 - `data` starts right after the vision section.
    pub fn new(data: &'a [u8]) -> PHI3VProcessor<'a> {
        assert!(data.len() >= 128, "processor section shorter than its 128-byte header");
        let args = ProcessorArgs {
            hidden_dim: rd_u32(data, 0),
            text_dim: rd_u32(data, 4),
            q_type: match data[8] {
                1 => QuantType::Q8_0,
                2 => QuantType::Q4_0,
                _ => QuantType::None,
            },
            group_size: rd_u32(data, 9),
        };
        let mut h: *mut LmrsProcessor = ptr::null_mut();
        let mut used: usize = 0;
        check(unsafe { ffi::lmrs_processor_create(data.as_ptr(), data.len(), ffi::device(), &mut h, &mut used) });
        PHI3VProcessor { args, h, _data: PhantomData }
    }

    /// processor.rs:234 - `out_patches`: the tower's output (global crop first); returns
    /// `(h_crop * patch_side) * (w_crop * patch_side + 1) + patch_side * (patch_side + 1) + 1` embeddings of `text_dim` floats.
    pub fn forward(&self, out_patches: &[f32], new_shape: u32, patch_side: u32, w_crop: u32, h_crop: u32) -> Vec<f32> {
        let n = (h_crop * patch_side) * (w_crop * patch_side + 1) + patch_side * (patch_side + 1) + 1;
        let mut out = vec![0.0f32; n as usize * self.args.text_dim as usize];
        let mut n_embeds: u32 = 0;
        check(unsafe {
            ffi::lmrs_processor_forward(self.h, out_patches.as_ptr(), out_patches.len() as u32, new_shape, patch_side, w_crop, h_crop,
                                        out.as_mut_ptr(), &mut n_embeds)
        });
        assert_eq!(n_embeds, n);
        out
    }
}

impl<'a> Drop for PHI3VProcessor<'a> {
    fn drop(&mut self) {
        unsafe { ffi::lmrs_processor_destroy(self.h) }
    }
}
