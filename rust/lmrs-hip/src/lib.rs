//! lmrs-hip: the MI355X (gfx950) back end of lm.rs's transformer forward path.
//!
//! Module names follow the reference crate (`lmrs::transformer`, `lmrs::vision`, `lmrs::processor`), so that
//! `use lmrs::transformer::Transformer` becomes `use lmrs_hip::transformer::Transformer` - or lm.rs re-exports these
//! modules in place of its own.  `sampler::Sampler` has the reference's signatures and adds
//! `Transformer::forward_sample` (the draw on the device); lm.rs's own sampler keeps working on `forward`'s logits.  Tokenizer,
//! the chat / web / desktop binaries and the image pre-processing (`PHI3VProcessor::process`) stay lm.rs's own code.
pub mod ffi;
pub mod sampler;
pub mod transformer;
#[cfg(feature = "multimodal")]
pub mod processor;
#[cfg(feature = "multimodal")]
pub mod vision;
