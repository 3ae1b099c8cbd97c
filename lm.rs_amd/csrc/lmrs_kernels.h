// lmrs_kernels.h — launch interface of the HIP kernels of the lm.rs decode hot path (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lmrs {

// Device-resident step state: lets one captured hipGraph be replayed for every token (no
// per-step parameter updates, no host round trip between tokens).
struct DevState {
    int pos;          // position of the token being processed
    int prompt_end;   // absolute position from which the argmax is written back as the next input
    int step_count;   // statistics
    int win_base;     // >= 0: first position of a batched forward_layer call (Gemma window quirk, see attention_body); < 0: decode
};

enum Prologue { PRO_PREQ = 0, PRO_QUANT = 1, PRO_RMS_QUANT = 2, PRO_ADD_RMS_QUANT = 3 };   // 3: x + rmsnorm(delta), then rmsnorm, quantise (static kernels only)
enum Epilogue { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_SWIGLU = 3, EPI_CLS = 4, EPI_GELU = 5,
                // CLIP tower (batched GEMM only): + bias with q / sqrt(head) | + bias + residual | + bias, QuickGELU
                EPI_VQKV = 6, EPI_BIAS_RESID = 7, EPI_BIAS_QGELU = 8,
                // image projector (processor.rs:234-342): + bias, tanh-GELU | + bias
                EPI_BIAS_GELU = 9, EPI_BIAS = 10,
                // merged qkv + attention launch: q / raw k / v leave as 8-byte {value, tag} granules (write-through), v also to its cache row
                EPI_QKV_TAG = 11,
                // batched w1/w3 whose epilogue also quantises h for w2 (256-row tiles: one group of h per token and tile): GemmArgs::hq / hs
                EPI_SWIGLU_Q = 12, EPI_GELU_Q = 13 };

struct EmbedArgs {
    const void* emb_q; const float* emb_s; int q4;
    const uint32_t* tokens;  // tokens[pos] is the input token
    float* x; int dim; float scale; int do_scale;   // Gemma: x *= sqrt(dim)
    const DevState* st;
};

struct ArgmaxArgs {
    const float* part_val; const int* part_idx; int n_part;
    int n_groups, group_stride;   // row-sharded classifier: n_groups shards of partials, group_stride entries apart (1 shard: 1, 0)
    // peer-to-peer row-sharded step: the gathered partials are double-buffered by the parity of the exchange that delivered them
    // (*part_par = exchanges finished on their slot): a shard that runs ahead pushes step s + 1 into the OTHER half while a slower peer
    // still reads step s (with one exchange per step - plan "cls" - nothing else orders the two).  null: single buffer.
    const unsigned* part_par; int part_par_floats;
    const float* logits;
    int tail_row;            // > 0: logits[tail_row ..] are never written by the classifier (vocab % 4 rows, functional.rs:183) and hold 0.0
    uint32_t* tokens; DevState* st;
    unsigned* seq;           // optional: the context's count of finished steps (the tags of the merged qkv + attention launch), bumped here
    EmbedArgs emb;           // the winner's (or the next prompt token's) embedding row is written to emb.x
    unsigned long long* dbg;
};

// EPI_CLS with the final argmax folded in (one GPU): workgroup 0 of the launch consumes the partials the GEMV workgroups publish as
// single 8-byte tagged words and does what the stand-alone argmax_final_kernel does (token feedback, position advance, next
// embedding row) - one launch and its boundary less per step.  See cls_consumer (lmrs_kernels.hip).
struct ClsTail {
    unsigned long long* part_pk;   // [grid - 1] packed partials {value | nan flag, tag, index}
    int* err;                      // set if the consumer's bounded sweep gives up
    unsigned* cls_seq;             // count of finished classifier-tail launches: the tags are made of THIS counter, which nothing else bumps
                                   // (the step counter m.seq is also bumped by layers-only steps: 2047 of them in a row would alias an 11-bit tag)
    ArgmaxArgs m;                  // part_val / part_idx / n_part unused here
};

struct GemvArgs {
    // weights: o rows of n int8 (Q8_0) / n/2 bytes (Q4_0), row-major; scales o * (n/128) f32
    const void* wq; const float* ws;
    int n, o;
    int q4;                  // 0: Q8_0, 1: Q4_0
    // activation input
    const float* xin;        // PRO_QUANT / PRO_RMS_QUANT: n f32
    const void* xq_in; const float* xs_in;   // PRO_PREQ: already quantised activation
    int preq_slice, preq_block;              // PRO_PREQ, sliced form (0: contiguous): blocks of [preq_slice int8 | preq_slice/128 scales], preq_block bytes apart
    const float* rms_w; float eps; int add_unit;   // PRO_RMS_QUANT
    const float* delta; const float* add_w; float* xout;   // PRO_ADD_RMS_QUANT: x' = xin + rmsnorm(delta, add_w) -> xout (a buffer other than xin)
    // outputs
    float* out;              // STORE: out[o]; RESID: out[i] += ; SWIGLU/GELU: out[o/2]; CLS: logits
    // EPI_QKV
    float* k_raw; float* v_cache; int att_dim, kv_dim, seq_len, layer;
    const DevState* st;
    // EPI_QKV_TAG: gran[row] = {value, tag = *seq + 1}; seq = steps finished so far on this context (never reset)
    unsigned long long* gran; const unsigned* seq;
    // EPI_CLS
    float* part_val; int* part_idx; int softcap_rows;   // Gemma: tanh soft-cap on (global) rows < softcap_rows
    const unsigned* part_par; int part_par_floats;      // the partials go to the half the NEXT exchange of their slot will push: ((*part_par + 1) & 1) * part_par_floats (see ArgmaxArgs)
    int row_offset;          // global index of this launch's row 0 (row-sharded classifier)
    int has_tail; ClsTail tail;   // EPI_CLS: fold the final argmax into this launch (see ClsTail)
    unsigned long long* dbg; // optional: 8 wall-clock stamps (debug timeline)
    int order_barrier;       // set by launch_gemv: workgroup barrier between the activation loads and the weight tile
    int chain_spread;        // set by launch_gemv: the RMSNorm chain of a CU's second workgroup runs on another wave (SIMD)
};

struct AttnArgs {
    const float* q;          // att_dim raw (un-rotated) query
    const float* k_raw;      // kv_dim raw key of this position
    float* k_cache;          // [layer][kv head][head_size / 4][seq_len][4]: blocked for the score lanes (see attention_body)
    const float* v_cache;    // [layer][seq_len][kv_dim]
    int chunk;               // V rows staged through LDS at a time (set by launch_attention)
    const float* rope;       // [seq_len][head_size/2][2] = (fcr, fci)
    float* out;              // att_dim
    int n_heads, n_kv_heads, head_size, seq_len, layer, gemma;
    unsigned long long* dbg; // optional: 8 wall-clock stamps (debug timeline)
    const DevState* st;
};

// launches (all asynchronous on `s`)
hipError_t launch_gemv(const GemvArgs& a, int pro, int epi, hipStream_t s, int grid_hint = 0);
void set_gemv_launch_events(hipEvent_t start, hipEvent_t stop);   // measurement: attach events to the next GEMV dispatches (null: off)
void set_launch_event_pool(hipEvent_t* pairs, int n_pairs, int* tags = nullptr);   // measurement: (start, stop) pairs for every following launch, in launch order (null: off)
void set_launch_tag(int tag);                                     // ... each labelled with the tag current at its launch
int launch_event_pool_used();                                     // launches seen since the pool was set (may exceed n_pairs)
bool next_launch_events(hipEvent_t* a, hipEvent_t* b);            // for launches made outside lmrs_kernels.hip
int gemv_grid(const GemvArgs& a, int pro, int epi);      // number of workgroups launch_gemv uses
bool gemv_is_static(const GemvArgs& a, int pro, int epi); // a compile-time-shape kernel exists for this launch
// unquantised models (q_type None): a.wq = o rows of n f32, a.xin = n f32; pro PRO_QUANT (as is) / PRO_RMS_QUANT (rmsnorm first); lmrs_f32.inc
hipError_t launch_gemv_f32(const GemvArgs& a, int pro, int epi, hipStream_t s);
int gemv_f32_grid(const GemvArgs& a, int epi);
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
// qkv GEMV + attention of one layer as ONE launch (short contexts): n_heads attention workgroups, dispatched first, prefetch the K / V
// rows of the earlier positions and poll the {value, tag} granules the GEMV workgroups of the same launch write (g.gran, g.seq).
// max_T: longest context (pos + 1) the launch will ever see (sizes the LDS score vector).  hipErrorNotSupported: no merged class
// for this shape - the caller launches the two kernels separately.
struct QkvAttnArgs { GemvArgs g; AttnArgs t; int* err; };
bool qkv_attn_supported(const GemvArgs& g, int pro, const AttnArgs& t);
int qkv_attn_wave_T(int head_size);       // longest context (pos + 1) of the one-wave-per-head form; 0: none for this head size
hipError_t launch_qkv_attn(const GemvArgs& g, int pro, const AttnArgs& t, int* err, int max_T, bool wave, hipStream_t s);
// long contexts: scores by (head, 256-key chunk), softmax + V by (head, quarter of the dims); S: attention_split_scratch_floats(..)
size_t attention_split_scratch_floats(int n_heads, int seq_len);
hipError_t launch_attention_split(const AttnArgs& a, float* S, int n_key_chunks, hipStream_t s);
hipError_t launch_embed(const EmbedArgs& a, hipStream_t s);
hipError_t launch_addvec(float* x, const float* d, int n, hipStream_t st);
hipError_t launch_addnorm(float* x, const float* delta, const float* w, int n, float eps, hipStream_t st);   // Gemma: x += rmsnorm(delta, 1 + w)
hipError_t launch_argmax_final(const ArgmaxArgs& a, hipStream_t s);
hipError_t launch_dequant_rows(const void* q, const float* s, int q4, const uint32_t* tokens, int n_tok, int dim, float* out, hipStream_t st);

// ---- the parallel part of Sampler::sample (sampler.rs:109-129, temperature != 0): logits[i] /= temperature (:115), the maximum, and
// logits[i] = exp(logits[i] - max) (functional.rs:126-133) in place.  The sequential chains behind it run on the host (lmrs_sampler_sample_exps).
// part: kSampleGrid + 1 floats of scratch (per-workgroup maxima, x[0]).
struct SampleArgs { float* logits; int n; float temperature; float* part; };
constexpr int kSampleGrid = 256;
hipError_t launch_sample_exps(const SampleArgs& a, hipStream_t s);
// sample_topp's candidate sort on the device (sampler.rs:67-81): see lmrs_kernels.hip.  N: power of two >= the candidates, >= sample_sort_min_n()
hipError_t launch_sample_topp_sort(const float* exps, int n, float sum, float cutoff, int N, unsigned long long* keys, unsigned* count, float* pairs_out, hipStream_t s);
int sample_sort_min_n();

// thin kernels over the same device functions, for the lmrs_op_* unit-parity entry points
hipError_t launch_quantize(const float* x, void* q, float* s, int n, int q4, hipStream_t st);
// batched prefill on row shards (Q8_0): this shard's slices of n_tok tokens quantised into its exchange block; the gathered blocks as a GEMM operand
hipError_t launch_quantize_rows(const float* x, int n, int n_tok, int q4, int8_t* q, float* s, hipStream_t st);
hipError_t launch_gather_rows(const char* blocks, size_t blk_stride, size_t s_off, int world, int n_l, int n_tok, int8_t* xq, float* xs, hipStream_t st, int xs_ld = 0);
hipError_t launch_scatter_rows(const char* blocks, size_t blk_stride, int world, int n_l, int n_tok, float* dst, int add, hipStream_t st);
hipError_t launch_rmsnorm(const float* x, const float* w, float* o, int n, float eps, int add_unit, hipStream_t st);
hipError_t launch_softmax(float* x, int n, hipStream_t st);
hipError_t launch_expf(const float* x, float* y, size_t n, hipStream_t st);
hipError_t launch_tanh_cast(const float* x, float* y, size_t n, double c, hipStream_t st);   // y = (float)tanh(c * (double)x)

constexpr int kMaxArgmaxParts = 4096;

// ---- peer-to-peer exchange of the row-sharded step (exchange_push_kernel)
constexpr int kMaxWorld = 8;              // shards of a peer-to-peer group: the GPUs of one node (sizes every per-peer array and the flag rows)
struct ExchangeArgs {
    const char* local;            // my block in my arena
    char* peer_dst[kMaxWorld];            // the same place in every peer's arena
    unsigned* peer_flag[kMaxWorld];       // my flag word in every peer's flag row of this slot
    unsigned* my_flags;           // my flag row of this slot (one word per source shard)
    unsigned* my_seq; int* err;
    int bytes, rank, world, slot; long long timeout_ticks;
    int par_bytes;                // > 0: double-buffered block - this exchange moves the half (sequence number & 1) * par_bytes (see ArgmaxArgs)
    const float* qsrc; int qn;    // optional: f32 slice to quantise (Q8_0) into `local` first: [qn int8 | qn / 128 f32 scales]
};
hipError_t launch_exchange_push(const ExchangeArgs& a, hipStream_t s);
hipError_t launch_exchange_copy_push(const ExchangeArgs& a, hipStream_t s);   // token batches: wide copy, then the flags

// ---- batched forward_layer (lmrs_prefill.inc): matmul_q8 over n_tok tokens on the int8 matrix cores
struct GemmArgs {
    const void* wq; const float* ws;     // weights [o][n] int8 + scales [o][n/128]
    const int8_t* xq; const float* xs;   // activations [n_tok][n] int8 + scales [n_tok][n/128]
    const float* xf;                     // launch_matmul_f32_rows only: activations [n_tok][n] f32 (wq then holds f32 weights [o][n])
    int n, o, n_tok; int q4;             // q4: weights [o][n/2] packed nibbles, activations int8 (q - 8), de-interleaved per 8
    float* out;                          // STORE / RESID: [n_tok][o]; SWIGLU: [n_tok][o/2]; QKV: q [n_tok][att_dim]
    float* k_raw; float* v_cache; int att_dim, kv_dim, seq_len, layer, pos0;    // EPI_QKV
    const float* bias; const float* resid; float qscale;                         // CLIP epilogues: bias [o], residual [n_tok][o], sqrt(head_size)
    // Scale layouts of the ring kernels (round 6): 0 = [row][n/128] as everywhere else; ld > 0 = TRANSPOSED, the scale of (row, group g) at base[g * ld + row] -
    // a group's scales of 64 rows are then 256 consecutive bytes, two cache lines, instead of 64 lines (a row's line holds 32 groups' scales of ONE row).
    // ws_ld: weights (the library keeps a transposed copy of the layers' scales for the batched path); xs_ld / hs_ld: activations in / quantised h out.
    int ws_ld, xs_ld, hs_ld;
    int8_t* hq; float* hs;                                                       // EPI_SWIGLU_Q / EPI_GELU_Q: quantised h [n_tok][o/2] int8 + scales [n_tok][o/256]
};
bool gemm_q8_hq_fused(int n, int o, int n_tok, bool q4);                         // host only: this w1/w3 launch can take the quantising epilogue
hipError_t launch_gemm_q8(const GemmArgs& a, int epi, hipStream_t s);
struct GemmTile { int tm, tn, waves; };                                        // weight rows x tokens of a workgroup's output tile, waves per workgroup
GemmTile gemm_q8_ring_tile(int n, int o, int n_tok, bool q4);                   // host only: the LDS-DMA ring kernel's tile for a launch of >= 48 tokens
hipError_t launch_matmul_f32_rows(const GemmArgs& a, int epi, hipStream_t s);   // q_type None sections of the image path (lmrs_f32.inc)
bool rows_prologue_supported(int n);
hipError_t launch_rows_prologue(float* x, const float* rms_w, const float* delta, const float* add_w, float eps, int add_unit, int mode, int q4,
                                int n, int n_tok, int8_t* xq, float* xs, hipStream_t s, int xs_ld = 0);      // xs_ld > 0: scales transposed, (token, group) at xs[g * xs_ld + token]
hipError_t launch_transpose_scales(const float* ws, int rows, int groups, float* wsT, hipStream_t s);          // wsT[g * rows + r] = ws[r * groups + g]
hipError_t launch_rows_addnorm(float* x, const float* delta, const float* w, float eps, int n, int n_tok, hipStream_t s);
hipError_t launch_rope_rows(float* q, const float* k_raw, float* k_cache, const float* rope, int n_heads, int n_kv_heads, int hs, int seq_len,
                            int layer, int pos0, int n_tok, hipStream_t s);
// block attention for batched prefill (Llama / Phi): 64 queries of one head per workgroup; scratch: attention_block_scratch_floats(..)
bool attention_block_supported(const AttnArgs& a, int n_tok);
size_t attention_block_scratch_floats(int n_heads, int n_tok, int T);
hipError_t launch_attention_block(const AttnArgs& a, int pos0, int n_tok, float* scratch, hipStream_t s);
hipError_t launch_attention_rows(const AttnArgs& a, int pos0, int n_tok, hipStream_t s);

// ---- CLIP vision tower (lmrs_vision.inc; reference src/vision.rs:244-577), dim 1024 / 16 heads x 64 / 577 tokens per crop
struct VisPatchArgs { const float* pixels; const float* kernel; const float* class_emb; const float* pos_emb; float* out; int dim, n_patches, kdim; };
hipError_t launch_vis_patch_embed(const VisPatchArgs& a, int num_crops, hipStream_t s);
// layernorm rows (functional.rs:80-114): out_f32 != null: f32 result; xq/xs != null: quantised result (Q8_0)
hipError_t launch_vis_layernorm(const float* x, const float* w, const float* b, float eps, int dim, int n_tok, float* out_f32, int8_t* xq, float* xs, hipStream_t s, int xs_ld = 0);
hipError_t launch_vis_attention(const float* qkv, float* out, float* scratch, int num_crops, int n_heads, int T, int dim, bool stray_workgroups, hipStream_t s);
size_t vis_attention_scratch_floats(int num_crops, int n_heads, int T);
hipError_t launch_vis_att_scores(const float* qkv, float* scratch, int num_crops, int n_heads, int nqb, int T, int dim, hipStream_t s);   // (a phase of launch_vis_attention)

}  // namespace lmrs
