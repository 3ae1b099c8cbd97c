// text.hpp — C++ host-side mirrors of lmrs::tokenizer::Tokenizer (reference src/tokenizer.rs:12-163) and
// lmrs::sampler::Sampler (src/sampler.rs:10-129) over the C ABI: same names, argument meaning, panics -> exceptions.
#pragma once
#include <fstream>
#include <iterator>
#include <string>

#include "transformer.hpp"

namespace lmrs_host {

enum class ModelType : int { GEMMA = 0, LLAMA = 1, PHI = 2 };

class Tokenizer {
public:
    std::uint32_t bos = 0, eos = 0;                       // pub bos / eos (tokenizer.rs:15-16)
    // Tokenizer::new(path)
    explicit Tokenizer(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw Panic("Error reading tokenizer file.");
        const std::vector<char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        check(lmrs_tokenizer_create(reinterpret_cast<const std::uint8_t*>(data.data()), data.size(), &h_));
        check(lmrs_tokenizer_info(h_, nullptr, &bos, &eos));
    }
    Tokenizer(const Tokenizer&) = delete;
    Tokenizer& operator=(const Tokenizer&) = delete;
    ~Tokenizer() { if (h_) lmrs_tokenizer_destroy(h_); }
    // encode(&mut self, text, bos, eos, chat_format, model_type) -> Vec<u32>
    std::vector<std::uint32_t> encode(const std::string& text, bool add_bos, bool add_eos, bool chat_format, ModelType model_type) {
        std::vector<std::uint32_t> out(text.size() + 32);
        std::size_t n = 0;
        check(lmrs_tokenizer_encode(h_, text.data(), text.size(), add_bos, add_eos, chat_format, static_cast<int>(model_type), out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    // decode(&self, token) -> String
    std::string decode(std::uint32_t token) const {
        char buf[512]; std::size_t n = 0;
        check(lmrs_tokenizer_decode(h_, token, buf, sizeof buf, &n));
        return std::string(buf, n);
    }

private:
    lmrs_tokenizer* h_ = nullptr;
};

class Sampler {
public:
    // Sampler::new(vocab_size, temperature, top_p, seed)
    Sampler(std::uint32_t vocab_size, float temperature, float top_p, std::uint64_t seed) { check(lmrs_sampler_create(vocab_size, temperature, top_p, seed, &h_)); }
    Sampler(const Sampler&) = delete;
    Sampler& operator=(const Sampler&) = delete;
    ~Sampler() { if (h_) lmrs_sampler_destroy(h_); }
    // sample(&mut self, logits: &mut [f32]) -> u32   (the logits are scaled / softmax-ed in place when temperature != 0)
    std::uint32_t sample(float* logits) { std::uint32_t next = 0; check(lmrs_sampler_sample(h_, logits, &next)); return next; }
    // model.forward(token, pos) followed by sample(logits), the draw made on the device: only the token id comes back (lmrs_forward_sample)
    std::uint32_t forward_sample(Transformer& model, std::uint32_t token, std::uint32_t pos) { return model.forward_sample(token, pos, h_); }

private:
    lmrs_sampler* h_ = nullptr;
};

}  // namespace lmrs_host
