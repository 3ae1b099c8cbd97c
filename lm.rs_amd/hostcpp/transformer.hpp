// transformer.hpp — C++ host-side mirror of lmrs::transformer::Transformer over the C ABI.
//
// The reference is compiled code (Rust) and its toolchain is absent from the build image, so the host
// layer above the C ABI is provided in C++ with the reference's names, argument meaning and error
// behaviour (reference src/transformer.rs:127-131 struct, :134 new, :316 forward, :659 get_embeddings,
// :672 fill_kv_cache; errors there are panics -> here exceptions).  Header-only; link liblmrs_hip.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../../include/lmrs_hip.h"

namespace lmrs_host {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc) { if (rc != 0) throw Panic(lmrs_last_error()); }

using TransformerArgs = lmrs_args;

class Transformer {
public:
    TransformerArgs args{};

    // Transformer::new(&mmap) -> (Transformer, usize): `data` is the mapped LMRS file.
    static std::pair<Transformer, std::size_t> create(const std::uint8_t* data, std::size_t len, int device = 0) {
        Transformer t; std::size_t used = 0;
        check(lmrs_create(data, len, device, &t.ctx_, &used));
        t.args = *lmrs_get_args(t.ctx_);
        return {std::move(t), used};
    }
    Transformer(Transformer&& o) noexcept : args(o.args), ctx_(o.ctx_) { o.ctx_ = nullptr; }
    Transformer& operator=(Transformer&& o) noexcept { if (this != &o) { reset(); args = o.args; ctx_ = o.ctx_; o.ctx_ = nullptr; } return *this; }
    Transformer(const Transformer&) = delete;
    Transformer& operator=(const Transformer&) = delete;
    ~Transformer() { reset(); }

    // forward(&mut self, token, pos) -> &mut [f32]: vocab_size logits owned by the model, valid until the next call.
    float* forward(std::uint32_t token, std::uint32_t pos) { float* p = nullptr; check(lmrs_forward(ctx_, token, pos, &p)); return p; }
    // forward + Sampler::sample_argmax without moving the logits off the device.
    std::uint32_t forward_argmax(std::uint32_t token, std::uint32_t pos) { std::uint32_t n = 0; check(lmrs_forward_argmax(ctx_, token, pos, &n)); return n; }
    std::vector<float> get_embeddings(const std::vector<std::uint32_t>& tokens) const {
        std::vector<float> out(tokens.size() * args.dim);
        check(lmrs_get_embeddings(ctx_, tokens.data(), tokens.size(), out.data()));
        return out;
    }
    std::uint32_t fill_kv_cache(std::vector<float>& embeddings, std::uint32_t curr_pos) {
        std::uint32_t np = 0;
        check(lmrs_fill_kv_cache(ctx_, embeddings.data(), static_cast<std::uint32_t>(embeddings.size() / args.dim), curr_pos, &np));
        return np;
    }
    // forward + Sampler::sample with the logits staying in HBM (text.hpp: Sampler::forward_sample); `sampler`: a handle of lmrs_sampler_create.
    std::uint32_t forward_sample(std::uint32_t token, std::uint32_t pos, lmrs_sampler* sampler) { std::uint32_t n = 0; check(lmrs_forward_sample(ctx_, token, pos, sampler, &n)); return n; }
    // chat.rs:188-222 at temperature 0.
    std::vector<std::uint32_t> generate_greedy(const std::vector<std::uint32_t>& prompt, std::uint32_t n_new, std::uint32_t start_pos = 0, double* seconds = nullptr) {
        std::vector<std::uint32_t> out(n_new);
        check(lmrs_generate_greedy(ctx_, prompt.data(), prompt.size(), n_new, start_pos, out.data(), seconds));
        return out;
    }

private:
    Transformer() = default;
    void reset() { if (ctx_) { lmrs_destroy(ctx_); ctx_ = nullptr; } }
    lmrs_ctx* ctx_ = nullptr;
};

}  // namespace lmrs_host
