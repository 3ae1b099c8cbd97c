// lmrs_format.cpp — see lmrs_format.h.
#include "lmrs_format.h"

#include <string.h>

namespace lmrs {

namespace {
struct Walker {
    size_t off;
    const lmrs_args& a;
    TensorView f32(size_t rows, size_t cols) {
        TensorView t; t.rows = rows; t.cols = cols; t.q_off = off; t.q_bytes = rows * cols * 4; off += t.q_bytes; return t;
    }
    // init_param_quant (transformer.rs:24-48): payload immediately followed by its f32 scales
    TensorView quant(size_t rows, size_t cols) {
        if (a.q_type == LMRS_Q_NONE) return f32(rows, cols);
        TensorView t; t.rows = rows; t.cols = cols;
        const size_t cnt = rows * cols;
        t.q_off = off; t.q_bytes = a.q_type == LMRS_Q4_0 ? cnt / 2 : cnt; off += t.q_bytes;
        t.s_off = off; t.s_bytes = cnt / a.group_size * 4; off += t.s_bytes;
        return t;
    }
};
}  // namespace

bool parse_layout(const uint8_t* d, size_t len, Layout* L, std::string* err) {
    auto bad = [&](const char* m) { if (err) *err = m; return false; };
    if (!d || len < 256) return bad("file shorter than the 256-byte LMRS header");
    if (!(d[0] == 0x6c && d[1] == 0x6d && d[2] == 0x72 && d[3] == 0x73)) return bad("Model not in lm.rs format.");
    lmrs_args& a = L->args;
    uint32_t u[8]; memcpy(u, d + 8, 32);                       // packed TransformerArgs at data[8..55]
    a.dim = u[0]; a.hidden_dim = u[1]; a.n_layers = u[2]; a.n_heads = u[3]; a.head_size = u[4];
    a.n_kv_heads = u[5]; a.vocab_size = u[6]; a.seq_len = u[7];
    memcpy(&a.rms_norm_eps, d + 40, 4); memcpy(&a.rope_theta, d + 44, 4);
    a.q_type = d[48]; a.model_type = d[49]; memcpy(&a.group_size, d + 50, 4); a.multimodal = d[54]; a._pad = 0;
    if (a.q_type > 2) return bad("unknown quantization type in header");
    if (a.model_type > 2) return bad("unknown model type in header");
    if (!a.dim || !a.n_layers || !a.n_heads || !a.n_kv_heads || !a.head_size || !a.vocab_size || !a.hidden_dim)
        return bad("zero-sized dimension in header");
    if (a.n_heads % a.n_kv_heads) return bad("n_heads not a multiple of n_kv_heads");
    if (a.q_type != LMRS_Q_NONE && (a.group_size == 0 || a.dim % a.group_size)) return bad("dim not a multiple of group_size");
    if (a.seq_len > 8192) a.seq_len = 8192;                    // transformer.rs:158-160
    const size_t dim = a.dim, nl = a.n_layers, att = (size_t)a.n_heads * a.head_size, kv = (size_t)a.n_kv_heads * a.head_size;
    const size_t hid = a.hidden_dim, V = a.vocab_size;
    const bool gemma = a.model_type == LMRS_GEMMA;
    Walker w{256, a};
    auto per_layer_f32 = [&](std::vector<TensorView>& v, size_t cols) { v.clear(); for (size_t l = 0; l < nl; ++l) v.push_back(w.f32(1, cols)); };
    auto per_layer_q = [&](std::vector<TensorView>& v, size_t rows, size_t cols) { v.clear(); for (size_t l = 0; l < nl; ++l) v.push_back(w.quant(rows, cols)); };
    L->emb = w.quant(V, dim);
    per_layer_f32(L->rms_att, dim);
    per_layer_q(L->wq, att, dim); per_layer_q(L->wk, kv, dim); per_layer_q(L->wv, kv, dim); per_layer_q(L->wo, dim, att);
    per_layer_f32(L->rms_post_att, dim);
    if (gemma) per_layer_f32(L->rms_pre_ffn, dim);
    per_layer_q(L->w1, hid, dim); per_layer_q(L->w2, dim, hid); per_layer_q(L->w3, hid, dim);
    if (gemma) per_layer_f32(L->rms_post_ffn, dim);
    L->rms_final = w.f32(1, dim);
    if (a.model_type == LMRS_PHI) L->lm_head = w.quant(V, dim);
    L->end = w.off;
    if (L->end > len) return bad("LMRS image truncated");
    return true;
}

}  // namespace lmrs
