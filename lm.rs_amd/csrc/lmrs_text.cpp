// lmrs_text.cpp — the two host-side callers either end of the device path, behind the same C ABI:
//   Tokenizer  reference src/tokenizer.rs: new :24-64 (tokenizer.bin), encode :66-151, decode :153-163
//   Sampler    reference src/sampler.rs: new :19-27, sample :109-129 (argmax :29-41, sample_mult :43-55, sample_topp :67-106),
//              with random_f32 / random_u32 of src/functional.rs:34-44
// Both are HOST code on purpose, bit for bit what the reference computes:
//   * the sampler's softmax (functional.rs:122-140) and sample_mult's running cdf (sampler.rs:43-55) are two sequential chains of
//     vocab_size additions; a GPU lane runs such a chain at ~4 cycles per add (2 x 0.21 ms for 128 256 logits, against a 0.44 ms
//     decode step), the host in ~0.1 ms from lmrs_forward's pinned logits (DESIGN.md section 7 has both numbers measured).  Greedy
//     decoding (temperature 0) never comes here: the argmax is fused into the classifier launch on the device.
//   * the tokenizer is string work.
// No GPU is needed for anything in this file (the -m "not gpu" tests exercise it).  Compiled with -ffp-contract=off like
// the rest of the library; expf is the host libm's, which is what Rust's f32::exp calls.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/lmrs_hip.h"

namespace lmrs { int text_fail(const char* msg); }
using lmrs::text_fail;

// ================================================================================================ Tokenizer
struct lmrs_tokenizer {
    uint32_t vocab_size = 0, bos = 0, eos = 0;
    std::vector<std::string> vocab;
    std::vector<float> scores;
    std::vector<uint32_t> sorted;          // ids ordered by token text (byte-wise, ties in id order: a stable sort as in :79)
    int bsearch_flavour = 0;
};

namespace {

uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
float rd_f32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

// Minimal UTF-8 validation: String::from_utf8 (tokenizer.rs:46) panics on an invalid token string.
bool valid_utf8(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        size_t k; uint32_t cp;
        if (c < 0x80) { ++i; continue; }
        else if ((c & 0xE0) == 0xC0) { k = 1; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { k = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { k = 3; cp = c & 0x07; }
        else return false;
        if (i + k >= n) return false;                       // the continuation bytes i+1 .. i+k must exist
        for (size_t j = 1; j <= k; ++j) {
            if ((s[i + j] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + j] & 0x3F);
        }
        if ((k == 1 && cp < 0x80) || (k == 2 && cp < 0x800) || (k == 3 && cp < 0x10000) || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += k + 1;
    }
    return true;
}

// slice::binary_search_by over the sorted vocabulary.  With duplicate token strings "any match" is what the Rust docs promise
// and WHICH match is an implementation detail of the std version the reference is built with (its toolchain is not pinned:
// no rust-toolchain file, no Cargo.lock).  Flavour 0 is the loop of Rust 1.52 .. 1.81 (returns the first probed match);
// flavour 1 the branch-free loop of later versions.  Different flavours can only disagree on vocabularies with duplicate strings.
int find_token(const lmrs_tokenizer& t, const std::string& key) {
    const size_t n = t.sorted.size();
    auto cmp = [&](size_t i) { return t.vocab[t.sorted[i]].compare(key); };   // byte-wise, as String::cmp
    if (t.bsearch_flavour == 0) {
        size_t left = 0, right = n, size = n;
        while (left < right) {
            const size_t mid = left + size / 2;
            const int c = cmp(mid);
            if (c < 0) left = mid + 1;
            else if (c > 0) right = mid;
            else return (int)mid;
            size = right - left;
        }
        return -1;
    }
    if (n == 0) return -1;
    size_t base = 0, size = n;
    while (size > 1) {
        const size_t half = size / 2, mid = base + half;
        if (cmp(mid) <= 0) base = mid;
        size -= half;
    }
    return cmp(base) == 0 ? (int)base : -1;
}

// the chars() of a UTF-8 string: byte length of the character starting at s[i]
size_t char_len(uint8_t c) { return c < 0x80 ? 1 : ((c & 0xE0) == 0xC0 ? 2 : ((c & 0xF0) == 0xE0 ? 3 : 4)); }

}  // namespace

extern "C" int lmrs_tokenizer_create(const uint8_t* data, size_t len, lmrs_tokenizer** out) {
    if (!data || !out) return text_fail("NULL argument");
    *out = nullptr;
    if (len < 16) return text_fail("tokenizer file shorter than its 16-byte header");
    if (rd_u32(data) > (len - 16) / 8) return text_fail("tokenizer file truncated");           // every entry is at least a score and a length: no allocation from an unchecked header
    lmrs_tokenizer* t = nullptr;
    try {
    t = new lmrs_tokenizer();
    t->vocab_size = rd_u32(data); t->bos = rd_u32(data + 8); t->eos = rd_u32(data + 12);      // [4..8) = max_token_len, unused (:28)
    size_t off = 16;
    t->vocab.reserve(t->vocab_size); t->scores.reserve(t->vocab_size);
    for (uint32_t i = 0; i < t->vocab_size; ++i) {
        if (off + 8 > len) { delete t; return text_fail("tokenizer file truncated"); }
        t->scores.push_back(rd_f32(data + off)); off += 4;
        const uint32_t sl = rd_u32(data + off); off += 4;
        if (off + sl > len) { delete t; return text_fail("tokenizer file truncated"); }
        if (!valid_utf8(data + off, sl)) { delete t; return text_fail("Error reading token string"); }
        t->vocab.emplace_back(reinterpret_cast<const char*>(data + off), sl); off += sl;
    }
    const char* fl = getenv("LMRS_BSEARCH_FLAVOUR");
    t->bsearch_flavour = fl ? atoi(fl) : 0;
    *out = t;
    return 0;
    } catch (...) { delete t; return text_fail("out of memory while reading the tokenizer file"); }   // nothing may unwind through the C ABI
}

extern "C" void lmrs_tokenizer_destroy(lmrs_tokenizer* t) { delete t; }

extern "C" int lmrs_tokenizer_info(const lmrs_tokenizer* t, uint32_t* vocab_size, uint32_t* bos, uint32_t* eos) {
    if (!t) return text_fail("NULL argument");
    if (vocab_size) *vocab_size = t->vocab_size;
    if (bos) *bos = t->bos;
    if (eos) *eos = t->eos;
    return 0;
}

// Tokenizer::encode (tokenizer.rs:66-151).  model_type: 0 GEMMA, 1 LLAMA, 2 PHI.  *n = number of ids; if it exceeds `cap`
// nothing is written beyond cap and the call fails (call again with a larger buffer: n <= bytes of text + 16).
static int tokenizer_encode_impl(lmrs_tokenizer* t, const char* text, size_t text_len, int bos, int eos, int chat_format, int model_type,
                                 uint32_t* out, size_t cap, size_t* n);
extern "C" int lmrs_tokenizer_encode(lmrs_tokenizer* t, const char* text, size_t text_len, int bos, int eos, int chat_format, int model_type,
                                     uint32_t* out, size_t cap, size_t* n) {
    try { return tokenizer_encode_impl(t, text, text_len, bos, eos, chat_format, model_type, out, cap, n); }
    catch (...) { return text_fail("out of memory while encoding"); }                                // nothing may unwind through the C ABI
}
static int tokenizer_encode_impl(lmrs_tokenizer* t, const char* text, size_t text_len, int bos, int eos, int chat_format, int model_type,
                                 uint32_t* out, size_t cap, size_t* n) {
    if (!t || !text || !n) return text_fail("NULL argument");
    if (text_len == 0) return text_fail("Text to encode should not be empty");
    if (!valid_utf8(reinterpret_cast<const uint8_t*>(text), text_len)) return text_fail("text is not valid UTF-8 (the reference takes a &str)");
    if (t->sorted.empty()) {                                                        // :69-80, built on first use
        t->sorted.resize(t->vocab_size);
        for (uint32_t i = 0; i < t->vocab_size; ++i) t->sorted[i] = i;
        std::stable_sort(t->sorted.begin(), t->sorted.end(), [&](uint32_t a, uint32_t b) { return t->vocab[a].compare(t->vocab[b]) < 0; });
    }
    std::vector<uint32_t> tok;
    if (bos) tok.push_back(t->bos);
    if (chat_format) {                                                              // :88-96
        if (model_type == 0) { const uint32_t p[] = {t->bos, 106, 1645, 108}; tok.insert(tok.end(), p, p + 4); }
        else if (model_type == 1) { const uint32_t p[] = {128006, 882, 128007, 271}; tok.insert(tok.end(), p, p + 4); }
        else if (model_type == 2) { const uint32_t p[] = {t->bos, 32010, 29871, 13}; tok.insert(tok.end(), p, p + 4); }
    }
    for (size_t i = 0; i < text_len;) {                                             // :98-108: one id per character, else its bytes + 3
        const size_t cl = char_len((uint8_t)text[i]);
        const std::string c(text + i, cl);
        const int idx = find_token(*t, c);
        if (idx >= 0) tok.push_back(t->sorted[idx]);
        else for (size_t b = 0; b < cl; ++b) tok.push_back((uint32_t)(uint8_t)text[i + b] + 3);
        i += cl;
    }
    for (uint32_t id : tok) if (id >= t->vocab_size) return text_fail("token id out of the vocabulary (the reference indexes vocab[id] and panics)");
    for (;;) {                                                                      // :110-135: merge the best-scoring adjacent pair
        float best_score = -1e10f; uint32_t best_id = 0; long best_idx = -1;
        for (size_t i = 0; i + 1 < tok.size(); ++i) {
            const std::string merged = t->vocab[tok[i]] + t->vocab[tok[i + 1]];
            const int idx = find_token(*t, merged);
            if (idx >= 0) {
                const uint32_t id = t->sorted[idx];
                if (t->scores[id] > best_score) { best_score = t->scores[id]; best_id = id; best_idx = (long)i; }
            }
        }
        if (best_idx == -1) break;
        tok[best_idx] = best_id;
        tok.erase(tok.begin() + best_idx + 1);
    }
    if (chat_format) {                                                              // :137-145
        if (model_type == 0) { const uint32_t p[] = {107, 108, 106, 2516, 108}; tok.insert(tok.end(), p, p + 5); }
        else if (model_type == 1) { const uint32_t p[] = {128009, 128006, 78191, 128007, 271}; tok.insert(tok.end(), p, p + 5); }
        else if (model_type == 2) { const uint32_t p[] = {32007, 29871, 13, 32001, 29871, 13}; tok.insert(tok.end(), p, p + 6); }
    }
    if (eos) tok.push_back(t->eos);
    *n = tok.size();
    if (tok.size() > cap || (!out && !tok.empty())) return text_fail("output buffer too small for the encoded ids");
    memcpy(out, tok.data(), tok.size() * 4);
    return 0;
}

// Tokenizer::decode (tokenizer.rs:153-163): the piece, or for "<0xHH>" the character U+00HH (char::from(u8), i.e. two UTF-8
// bytes from 0x80 up).  Not NUL-terminated; *n = bytes written.
extern "C" int lmrs_tokenizer_decode(const lmrs_tokenizer* t, uint32_t token, char* out, size_t cap, size_t* n) {
    if (!t || !out || !n) return text_fail("NULL argument");
    if (token >= t->vocab_size) return text_fail("token out of range");
    const std::string& p = t->vocab[token];
    std::string r = p;
    if (p.size() == 6 && p.compare(0, 3, "<0x") == 0 && p[5] == '>') {
        auto hex = [](char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1)); };
        const int hi = hex(p[3]), lo = hex(p[4]);
        if (hi >= 0 && lo >= 0) {                                                   // u8::from_str_radix(.., 16)
            const unsigned v = (unsigned)(hi * 16 + lo);
            r.clear();
            if (v < 0x80) r.push_back((char)v);
            else { r.push_back((char)(0xC0 | (v >> 6))); r.push_back((char)(0x80 | (v & 0x3F))); }
        }
    }
    *n = r.size();
    if (r.size() > cap) return text_fail("output buffer too small for the piece");
    memcpy(out, r.data(), r.size());
    return 0;
}

// ================================================================================================ Sampler
struct lmrs_sampler {
    uint32_t vocab_size = 0; float temperature = 0, top_p = 0; uint64_t seed = 0;
    struct ProbIndex { float prob; uint32_t index; };
    std::vector<ProbIndex> probindex;      // persists across calls, as in the reference (entries beyond n0 keep older values)
    bool rest_ordered = true;              // the vector is in descending order of prob (true from creation: all zeros; see topp_tail)
    size_t pending_n0 = (size_t)-1;        // between lmrs_sampler_exps_prepare and _finish: this call's candidates sit in probindex[0 .. pending_n0)
};

namespace {
uint32_t random_u32(uint64_t state) {      // functional.rs:34-40 (xorshift*; the multiplication wraps in a release build)
    state ^= state >> 12; state ^= state << 25; state ^= state >> 27;
    return (uint32_t)((state * 0x2545F4914F6CDD1Dull) >> 32);
}
float random_f32(uint64_t state) { return (float)(random_u32(state) >> 8) / 16777216.0f; }   // :42-44
}  // namespace

extern "C" int lmrs_sampler_create(uint32_t vocab_size, float temperature, float top_p, uint64_t seed, lmrs_sampler** out) {
    if (!out) return text_fail("NULL argument");
    *out = nullptr;
    if (vocab_size == 0) return text_fail("vocab_size must be positive");
    lmrs_sampler* s = nullptr;
    try {
        s = new lmrs_sampler();
        s->vocab_size = vocab_size; s->temperature = temperature; s->top_p = top_p; s->seed = seed;
        s->probindex.assign(vocab_size, {0.0f, 0u});
    } catch (...) { delete s; return text_fail("out of memory (sampler candidates)"); }                // nothing may unwind through the C ABI
    *out = s;
    return 0;
}
extern "C" void lmrs_sampler_destroy(lmrs_sampler* s) { delete s; }

// the sampler's parameters and the random number every call of it draws (random_f32(seed): the seed never advances, sampler.rs:119) -
// what lmrs_forward_sample needs to run Sampler::sample on the device
extern "C" int lmrs_sampler_info(const lmrs_sampler* s, uint32_t* vocab_size, float* temperature, float* top_p, float* rnd) {
    if (!s) return text_fail("NULL argument");
    if (vocab_size) *vocab_size = s->vocab_size;
    if (temperature) *temperature = s->temperature;
    if (top_p) *top_p = s->top_p;
    if (rnd) *rnd = random_f32(s->seed);
    return 0;
}

// sample_topp from its sort on (sampler.rs:81-105): probindex[0 .. n0) holds this call's candidates in index order
static int topp_tail(lmrs_sampler* s, size_t n0, float rnd, uint32_t* next, bool presorted = false) {
    // :81 sorts the WHOLE vector (stable, descending by prob), stale entries of earlier calls included.  The vector left that sort fully
    // ordered last time (and starts as all zeros), and this call rewrote only its first n0 entries: the rest is still ordered.  The
    // stable sort of the whole is therefore the stable sort of those n0 entries merged stably with the ordered rest (equal elements:
    // first range first, i.e. lower original position first - exactly what a stable sort of the whole yields) - O(n0 log n0 + n)
    // instead of O(n log n) over 128 256 entries per token (0.65 ms of a 1.4 ms token).  A NaN among the probabilities breaks the
    // ordering argument: from then on the whole vector is sorted every time, as written.
    const auto desc = [](const lmrs_sampler::ProbIndex& a, const lmrs_sampler::ProbIndex& b) { return a.prob > b.prob; };
    for (size_t i = 0; i < n0 && s->rest_ordered; ++i) if (!(s->probindex[i].prob == s->probindex[i].prob)) s->rest_ordered = false;
    if (s->rest_ordered) {
        // (presorted: the caller sorted this call's n0 candidates by (prob descending, index ascending) - what the stable sort of the index-ordered
        // candidates gives - on the device: lmrs_forward_sample, launch_sample_topp_sort)
        if (!presorted) std::stable_sort(s->probindex.begin(), s->probindex.begin() + (ptrdiff_t)n0, desc);
        std::inplace_merge(s->probindex.begin(), s->probindex.begin() + (ptrdiff_t)n0, s->probindex.end(), desc);
    } else std::stable_sort(s->probindex.begin(), s->probindex.end(), desc);
    if (n0 == 0) return text_fail("sample_topp: no candidate above the cutoff (the reference underflows n0 - 1 and panics)");
    float cumulative = 0.0f; size_t last_idx = n0 - 1;
    for (size_t i = 0; i < n0; ++i) { cumulative = cumulative + s->probindex[i].prob; if (cumulative > s->top_p) { last_idx = i; break; } }
    const float r = rnd * cumulative;
    float cdf = 0.0f;
    for (size_t i = 0; i <= last_idx; ++i) { cdf = cdf + s->probindex[i].prob; if (r < cdf) { *next = s->probindex[i].index; return 0; } }
    *next = s->probindex[last_idx].index;
    return 0;
}

// sample_topp for a caller that has run the temperature scaling, the softmax and the cutoff filter elsewhere (lmrs_forward_sample: on
// the device): `pairs` = the n0 candidates {f32 prob, u32 index} with prob >= (1 - top_p) / (vocab_size - 1), in index order - what
// sampler.rs:74-80 writes into probindex[0 .. n0).  The sort over the persistent vector, the cumulative cut and the draw run here.
extern "C" int lmrs_sampler_topp_pairs(lmrs_sampler* s, const void* pairs, size_t n0, uint32_t* next) {
    if (!s || (!pairs && n0) || !next) return text_fail("NULL argument");
    if (!(s->top_p > 0.0f && s->top_p < 1.0f) || s->temperature == 0.0f) return text_fail("not a top-p sampler");
    if (n0 > s->probindex.size()) return text_fail("more candidates than the vocabulary has entries");
    static_assert(sizeof(lmrs_sampler::ProbIndex) == 8, "pair layout");
    if (n0) memcpy(s->probindex.data(), pairs, n0 * sizeof(lmrs_sampler::ProbIndex));
    return topp_tail(s, n0, random_f32(s->seed), next);
}

// Sampler::sample from the softmax's exponentials on (functional.rs:134-139, sampler.rs:119-128), for a caller that formed
// exps[i] = exp(logits[i] / temperature - max) elsewhere (lmrs_forward_sample: on the device, the only part of the sampler that is parallel work).
// What is left is the reference's two sequential chains - the softmax sum and the running cdf (or the candidates' sort) - and those run
// here: a host core adds a dependent f32 in ~1 ns, one GPU lane in ~2.5 ns.  exps become the probabilities, as `logits` does in the reference.
extern "C" int lmrs_sampler_exps_prepare(lmrs_sampler* s, float* exps, float* sum_out, float* cutoff_out, size_t* n0_out) {
    if (!s || !exps) return text_fail("NULL argument");
    if (s->temperature == 0.0f) return text_fail("temperature 0 is sample_argmax: no softmax to finish");
    const size_t n = s->vocab_size;
    float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) sum = sum + exps[i];                              // functional.rs:134 (the adds of the exp loop, in its order)
    for (size_t i = 0; i < n; ++i) exps[i] = exps[i] / sum;                           // :137-139
    size_t n0 = 0; float cutoff = 0.0f;
    if (s->top_p > 0.0f && s->top_p < 1.0f) {                                         // sample_topp :67-80: the candidates, in index order
        cutoff = (1.0f - s->top_p) / (float)(n - 1);
        for (size_t i = 0; i < n; ++i)
            if (exps[i] >= cutoff) { s->probindex[n0].index = (uint32_t)i; s->probindex[n0].prob = exps[i]; ++n0; }
    }
    s->pending_n0 = n0;
    if (sum_out) *sum_out = sum;
    if (cutoff_out) *cutoff_out = cutoff;
    if (n0_out) *n0_out = n0;
    return 0;
}
extern "C" int lmrs_sampler_exps_finish(lmrs_sampler* s, const float* probs, const void* sorted_pairs, uint32_t* next) {
    if (!s || !probs || !next) return text_fail("NULL argument");
    if (s->pending_n0 == (size_t)-1) return text_fail("lmrs_sampler_exps_finish without lmrs_sampler_exps_prepare");
    const size_t n = s->vocab_size, n0 = s->pending_n0;
    s->pending_n0 = (size_t)-1;
    const float rnd = random_f32(s->seed);
    if (s->top_p <= 0.0f || s->top_p >= 1.0f) {                                       // sample_mult :43-55
        float cdf = 0.0f;
        for (size_t i = 0; i < n; ++i) { cdf = cdf + probs[i]; if (rnd < cdf) { *next = (uint32_t)i; return 0; } }
        *next = (uint32_t)(n - 1);
        return 0;
    }
    static_assert(sizeof(lmrs_sampler::ProbIndex) == 8, "pair layout");
    if (sorted_pairs && n0) memcpy(s->probindex.data(), sorted_pairs, n0 * sizeof(lmrs_sampler::ProbIndex));
    return topp_tail(s, n0, rnd, next, sorted_pairs != nullptr);
}
extern "C" int lmrs_sampler_sample_exps(lmrs_sampler* s, float* exps, uint32_t* next) {
    if (!s || !exps || !next) return text_fail("NULL argument");
    if (lmrs_sampler_exps_prepare(s, exps, nullptr, nullptr, nullptr)) return -1;
    return lmrs_sampler_exps_finish(s, exps, nullptr, next);
}

// Sampler::sample (sampler.rs:109-129).  logits (vocab_size floats) are scaled and softmax-ed IN PLACE when temperature != 0, as
// the reference does to the slice `forward` returned.  The random number is random_f32(self.seed) on every call: the seed is
// never advanced (:119), so one Sampler draws the same number each time - reproduced.
extern "C" int lmrs_sampler_sample(lmrs_sampler* s, float* logits, uint32_t* next) {
    if (!s || !logits || !next) return text_fail("NULL argument");
    const size_t n = s->vocab_size;
    if (s->temperature == 0.0f) {                                                   // sample_argmax :29-41
        uint32_t max_i = 0; float max_p = logits[0];
        for (size_t i = 1; i < n; ++i) if (logits[i] > max_p) { max_i = (uint32_t)i; max_p = logits[i]; }
        *next = max_i;
        return 0;
    }
    for (size_t q = 0; q < n; ++q) logits[q] = logits[q] / s->temperature;         // :115
    {                                                                               // softmax, functional.rs:122-140
        float sum = 0.0f, max_val = logits[0];
        for (size_t i = 0; i < n; ++i) if (logits[i] > max_val) max_val = logits[i];
        for (size_t i = 0; i < n; ++i) { logits[i] = expf(logits[i] - max_val); sum = sum + logits[i]; }
        for (size_t i = 0; i < n; ++i) logits[i] = logits[i] / sum;
    }
    const float rnd = random_f32(s->seed);
    if (s->top_p <= 0.0f || s->top_p >= 1.0f) {                                     // sample_mult :43-55
        float cdf = 0.0f;
        for (size_t i = 0; i < n; ++i) { cdf = cdf + logits[i]; if (rnd < cdf) { *next = (uint32_t)i; return 0; } }
        *next = (uint32_t)(n - 1);
        return 0;
    }
    // sample_topp :67-106
    size_t n0 = 0;
    const float cutoff = (1.0f - s->top_p) / (float)(n - 1);
    for (size_t i = 0; i < n; ++i)
        if (logits[i] >= cutoff) { s->probindex[n0].index = (uint32_t)i; s->probindex[n0].prob = logits[i]; ++n0; }
    return topp_tail(s, n0, rnd, next);
}
