//! `VisionTransformer` (reference src/vision.rs:68-71): `new` :99-243, `forward` :244-577, `Drop` :581.
use std::marker::PhantomData;
use std::ptr;

use crate::ffi::{self, check, LmrsVision};
use crate::transformer::QuantType;

/// The 37 header bytes of the vision section (vision.rs:11-24); `patch_size` / `image_size` are what chat.rs reads (:102, :108).
#[derive(Debug, Copy, Clone)]
pub struct VisionTransformerArgs {
    pub(crate) dim: u32,
    pub(crate) hidden_dim: u32,
    pub(crate) n_layers: u32,
    pub(crate) n_heads: u32,
    pub(crate) head_size: u32,
    pub(crate) layernorm_eps: f32,
    pub patch_size: u32,
    pub image_size: u32,
    pub(crate) q_type: QuantType,
    pub(crate) group_size: u32,
}

fn rd_u32(d: &[u8], off: usize) -> u32 {
    u32::from_le_bytes([d[off], d[off + 1], d[off + 2], d[off + 3]])
}

impl VisionTransformerArgs {
    fn parse(data: &[u8]) -> VisionTransformerArgs {
        assert!(data.len() >= 128, "vision section shorter than its 128-byte header");
        VisionTransformerArgs {
            dim: rd_u32(data, 0),
            hidden_dim: rd_u32(data, 4),
            n_layers: rd_u32(data, 8),
            n_heads: rd_u32(data, 12),
            head_size: rd_u32(data, 16),
            layernorm_eps: f32::from_bits(rd_u32(data, 20)),
            patch_size: rd_u32(data, 24),
            image_size: rd_u32(data, 28),
            q_type: match data[32] {
                1 => QuantType::Q8_0,
                2 => QuantType::Q4_0,
                _ => QuantType::None,
            },
            group_size: rd_u32(data, 33),
        }
    }
}

pub struct VisionTransformer<'a> {
    pub args: VisionTransformerArgs,
    h: *mut LmrsVision,
    _data: PhantomData<&'a [u8]>,
}

impl<'a> VisionTransformer<'a> {
    /// vision.rs:99 - `data` starts at the offset `Transformer::new` returned; the second element is the size of the vision
    /// section (the processor section follows it).
    pub fn new(data: &'a [u8]) -> (VisionTransformer<'a>, usize) {
        let args = VisionTransformerArgs::parse(data);
        let mut h: *mut LmrsVision = ptr::null_mut();
        let mut used: usize = 0;
        check(unsafe { ffi::lmrs_vision_create(data.as_ptr(), data.len(), ffi::device(), &mut h, &mut used) });
        (VisionTransformer { args, h, _data: PhantomData }, used)
    }

    /// vision.rs:244 - `pixel_values`: `num_crops * 3 * image_size^2` floats as `PHI3VProcessor::process` produces them;
    /// returns the patch embeddings of all crops (class token dropped) and the floats per crop.
    pub fn forward(&mut self, pixel_values: &[f32], num_crops: u32) -> (Vec<f32>, u32) {
        let side = (self.args.image_size / self.args.patch_size) as usize;
        let per_crop = 3 * (self.args.image_size as usize) * (self.args.image_size as usize);
        assert!(pixel_values.len() >= num_crops as usize * per_crop, "pixel_values shorter than num_crops crops");
        let mut out = vec![0.0f32; num_crops as usize * side * side * self.args.dim as usize];
        let mut new_shape: u32 = 0;
        check(unsafe { ffi::lmrs_vision_forward(self.h, pixel_values.as_ptr(), num_crops, out.as_mut_ptr(), &mut new_shape) });
        (out, new_shape)
    }
}

impl<'a> Drop for VisionTransformer<'a> {
    fn drop(&mut self) {
        unsafe { ffi::lmrs_vision_destroy(self.h) }
    }
}
