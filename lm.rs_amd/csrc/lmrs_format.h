// lmrs_format.h — LMRS v4 weight-file layout (host side, no GPU dependency).
//
// Mirrors Transformer::new's walk over the mmap (reference src/transformer.rs:134-160 header,
// :16-48 init_param / init_param_quant, :241-270 tensor order) and the writer it must agree with
// (reference export.py:51-126, utils/io.py:21-56).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/lmrs_hip.h"

namespace lmrs {

struct TensorView {      // one tensor of one layer inside the file image
    size_t q_off = 0;    // byte offset of the int8 / packed-nibble / f32 payload
    size_t q_bytes = 0;
    size_t s_off = 0;    // byte offset of the f32 group scales (quantised tensors only)
    size_t s_bytes = 0;
    size_t rows = 0, cols = 0;
};

struct Layout {
    lmrs_args args{};
    size_t end = 0;      // first byte after the text model == Transformer::new's second return value
    TensorView emb, lm_head, rms_final;
    std::vector<TensorView> rms_att, rms_post_att, rms_pre_ffn, rms_post_ffn;   // per layer, f32
    std::vector<TensorView> wq, wk, wv, wo, w1, w2, w3;                           // per layer
};

// Parses the header and computes every tensor's offsets.  Returns false and sets err on a malformed
// or truncated image (the reference panics: transformer.rs:135 assert_eq!, slice index panics).
bool parse_layout(const uint8_t* file, size_t len, Layout* out, std::string* err);

}  // namespace lmrs
