/*
 * lmrs_hip.h — C ABI of the MI355X-native decode hot path for lm.rs models.
 *
 * This is the drop-in boundary: every entry point replaces one
 * member of the public surface of `lmrs::transformer::Transformer`
 * (reference src/transformer.rs) or one of the L2 free functions in
 * src/functional.rs / src/quantization.rs.  The reference has no FFI of its own,
 * so the signatures below are what a Rust `extern "C"` block would bind (the
 * binding itself is shown in INTEGRATION.md).  Plain pointers and sizes only.
 *
 * The same prototypes, with the prefix `lmrs_ref_` instead of `lmrs_`, are
 * implemented by the CPU oracle (oracle/lmrs_oracle.c).  The oracle is test
 * infrastructure; nothing in this library calls it.
 *
 * Conventions
 *   - every function returning int: 0 = ok, <0 = error (text via lmrs_last_error()).
 *     The reference panics on error (assert!/expect); a Rust shim turns !=0 into panic!.
 *   - a context is NOT thread-safe (reference: `&mut self`); several contexts may coexist.
 *   - all host pointers are caller-owned unless stated.
 */
#ifndef LMRS_HIP_H
#define LMRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmrs_ctx lmrs_ctx;

/* Mirror of TransformerArgs (reference src/transformer.rs:57-74), natural C layout
 * (the on-disk struct is packed; see lmrs_create).  seq_len is already clamped to
 * 8192 as the reference does (src/transformer.rs:158-160). */
typedef struct lmrs_args {
    uint32_t dim, hidden_dim, n_layers, n_heads, head_size, n_kv_heads, vocab_size, seq_len;
    float    rms_norm_eps, rope_theta;
    uint8_t  q_type;      /* 0 None, 1 Q8_0, 2 Q4_0      (src/quantization.rs:1-6)  */
    uint8_t  model_type;  /* 0 GEMMA, 1 LLAMA, 2 PHI      (src/transformer.rs:50-55) */
    uint8_t  multimodal;
    uint8_t  _pad;
    uint32_t group_size;
} lmrs_args;

enum { LMRS_Q_NONE = 0, LMRS_Q8_0 = 1, LMRS_Q4_0 = 2 };
enum { LMRS_GEMMA = 0, LMRS_LLAMA = 1, LMRS_PHI = 2 };

/* ---- Transformer::new  (src/transformer.rs:134-314) -------------------------------
 * Parses an LMRS v4 image (`file`,`len` = the mmap the reference hands to
 * Transformer::new), uploads the weights to HBM on HIP device `device`, allocates the
 * KV cache and activation buffers there.  *bytes_consumed = offset of the first byte
 * after the text model (what the reference returns as the second tuple element).
 * The host image is not referenced after return.  Q8_0, Q4_0 (group size 128) and unquantised (q_type None, f32: the reference's
 * `matmul`, functional.rs:142-171) files of the Llama / Gemma-2 / Phi families; f32 files on one GPU only.  Geometry limits (all files the
 * reference's exporter writes for its model families meet them): dim, n_heads * head_size and hidden_dim multiples of 128 - also for f32
 * files, where the reference itself only needs multiples of 8 -, head_size 64 / 96 / 128 / 256, group_size 128. */
int lmrs_create(const uint8_t* file, size_t len, int device, lmrs_ctx** out, size_t* bytes_consumed);

/* Row-sharded variant (no reference counterpart - the reference is one process on one host; replaces the same
 * Transformer::new, transformer.rs:134): this process is shard `rank` of `world`
 * (one process per GPU).  `nccl_unique_id` points at the 128-byte ncclUniqueId created
 * by rank 0 and distributed by the host (e.g. torch.distributed broadcast); it may be
 * NULL when world == 1.  NOTE: with world > 1 a NULL id does NOT fail - it selects the peer-to-peer transport: the context is created
 * unconnected and every step fails with "peers not connected" until lmrs_p2p_connect has run (lmrs_p2p_handle / lmrs_p2p_connect below;
 * at most 8 shards: the GPUs of one node).  Results are bit-identical to world == 1. */
int lmrs_create_sharded(const uint8_t* file, size_t len, int device, int rank, int world,
                        const void* nccl_unique_id, lmrs_ctx** out, size_t* bytes_consumed);

/* Writes the 128-byte ncclUniqueId for lmrs_create_sharded (call on rank 0).  No reference counterpart. */
int lmrs_comm_unique_id(void* out128);

/* Peer-to-peer transport (no reference counterpart): lmrs_create_sharded with world > 1 and nccl_unique_id == NULL makes a shard whose
 * exchanges are direct pushes into its peers' exchange arenas over xGMI (a store + flag kernel per exchange) instead of RCCL
 * all-gathers.  Every rank exports the 64-byte IPC handle of its arena, the launcher distributes all `world` handles in rank order
 * (any host transport), every rank connects; then the context is used like any other. */
int lmrs_p2p_handle(lmrs_ctx* ctx, void* out64);
int lmrs_p2p_connect(lmrs_ctx* ctx, const void* handles);

/* Host-only: the row ranges shard `rank` of `world` owns.  plan[0..9] = q-head first,count; kv-head first,count;
 * wo/w2 row first,count; gate/up pair first,count; classifier row first,count.  wo and w2 are replicated by default
 * (every shard: first 0, count dim - no gather after them); LMRS_SHARD_SPLIT_OUT=1 row-splits them as well.
 * Which matrices are split at all depends on the model and the world size: when the gate / up / down bytes a shard would stop reading
 * (3 * dim * hidden_dim * (1 - 1/world), quantised) are under ~57 MB per layer - two latency-bound exchanges per layer cost more than
 * that streams - the plan is "cls": heads, rows and pairs are all of them on every shard (the layers run whole, no exchange inside
 * them) and only the classifier rows are split: one exchange of argmax partials per token.  LMRS_SHARD_PLAN=tp|cls overrides.  (The
 * 1B / 2B models: cls at every world size; 3B / 3.8B: cls up to 4 GPUs; 8B and larger: tp.)  lmrs_create_sharded follows this plan. */
int lmrs_shard_plan(const lmrs_args* args, int rank, int world, int* plan10);
/* 1: the sharded step is one captured hipGraph (RCCL inside); 0: enqueued call by call; -1: not an RCCL shard.  No reference counterpart. */
int lmrs_shard_uses_graph(const lmrs_ctx* ctx);
/* Ranks of the context's RCCL communicator as RCCL itself counts them (ncclCommCount); 0: the context has no communicator (one GPU, or
 * the peer-to-peer transport); -1: error.  Measurement aid: lets a benchmark record that the all-gathers really spanned N ranks. */
int lmrs_comm_ranks(const lmrs_ctx* ctx);

/* Verification aid (no reference counterpart): `world` row shards of one model as `world` contexts on ONE device,
 * exchanged by device-to-device copies instead of RCCL, so the sharding can be checked bit for bit on a 1-GPU box. */
int lmrs_group_create(const uint8_t* file, size_t len, int device, int world, lmrs_ctx** shards, size_t* bytes_consumed);
/* One decode step over such a group (Transformer::forward, transformer.rs:316-384, on the sharded layout).  *logits (optional) = the assembled logits (pinned, owned by shards[0]);
 * *next (optional) = the greedy token. */
int lmrs_group_forward(lmrs_ctx** shards, int world, uint32_t token, uint32_t pos, float** logits, uint32_t* next);

/* Drop for Transformer (src/transformer.rs:688-712). */
void lmrs_destroy(lmrs_ctx* ctx);

/* `pub args` (src/transformer.rs:128). Pointer is valid for the life of ctx. */
const lmrs_args* lmrs_get_args(const lmrs_ctx* ctx);

/* ---- Transformer::forward  (src/transformer.rs:316-384) ---------------------------
 * One decode step.  *logits points at ctx-owned pinned host memory holding
 * vocab_size floats, valid (and mutable: the reference sampler scales them in place,
 * src/sampler.rs:115-117) until the next call on this ctx. */
int lmrs_forward(lmrs_ctx* ctx, uint32_t token, uint32_t pos, float** logits);

/* Same step followed by Sampler::sample_argmax (src/sampler.rs:29-41: first index of the
 * maximum) on the device; no logits leave HBM. */
int lmrs_forward_argmax(lmrs_ctx* ctx, uint32_t token, uint32_t pos, uint32_t* next);

/* ---- Transformer::get_embeddings  (src/transformer.rs:659-669) -------------------- */
int lmrs_get_embeddings(const lmrs_ctx* ctx, const uint32_t* tokens, size_t n, float* out /* n*dim */);

/* ---- Transformer::fill_kv_cache  (src/transformer.rs:672-684) ---------------------
 * Runs all layers over `n` embeddings (n*dim floats, updated in place exactly as the
 * reference mutates its argument) at positions curr_pos..curr_pos+n-1; no logits.
 * *new_pos = curr_pos + n.  The supported model shapes (Q8_0 / Q4_0; Llama / Phi heads, Gemma-2) on
 * one GPU and on row-split shards run forward_layer over the whole batch (int8 matrix-core GEMMs,
 * transformer.rs:388-657 with sl = n; row shards: two all-gathers of quantised token-batch blocks
 * per layer, four when wo / w2 are split too); other shapes (f32 files, other geometries) go token
 * by token through the decode kernels.  Same values either way.
 * One documented deviation from the reference: n > 1 on a Q4_0 file (INTEGRATION.md
 * "Deviations": the reference multiplies token j with token 2j's nibbles; this computes the
 * token-by-token result). */
int lmrs_fill_kv_cache(lmrs_ctx* ctx, float* embeddings, uint32_t n, uint32_t curr_pos, uint32_t* new_pos);

/* ---- the generation loop of src/bin/chat.rs:188-222 on token IDs, greedy ------------
 * Feeds prompt[0..n_prompt) token by token starting at position start_pos (sampler
 * output ignored while the prompt lasts, as chat.rs does), then n_new-1 further steps
 * feeding back the argmax.  out_tokens[i] (i < n_new) = i-th generated token, i.e. the
 * argmax after step n_prompt-1+i.  Runs device-resident: one host sync at the end.
 * *seconds (optional) = wall time of the whole call's device work (HIP events). */
int lmrs_generate_greedy(lmrs_ctx* ctx, const uint32_t* prompt, size_t n_prompt, uint32_t n_new,
                         uint32_t start_pos, uint32_t* out_tokens, double* seconds);

const char* lmrs_last_error(void);

/* ---- L2 free functions, for unit parity (host pointers in and out) ------------------
 * Each runs the SAME device kernel the decode path uses, on device `device`.          */
/* functional.rs:173-214.  x: sl rows of n int8 + sl*n/gs scales; w: o*n int8 + o*n/gs scales. */
int lmrs_op_matmul_q8(int device, float* xout, const int8_t* xq, const float* xs,
                      const int8_t* wq, const float* ws, size_t n, size_t o, size_t gs, size_t sl);
/* functional.rs:216-250 (decode form, sl = 1).  xq: n/2 bytes, wq: o*n/2 bytes. */
int lmrs_op_matmul_q4(int device, float* xout, const uint8_t* xq, const float* xs,
                      const uint8_t* wq, const float* ws, size_t n, size_t o, size_t gs);
/* quantization.rs:44-67 */
int lmrs_op_quantize(int device, int8_t* q, float* s, const float* x, size_t n, size_t gs);
/* quantization.rs:69-95 */
int lmrs_op_quantize_q4(int device, uint8_t* q, float* s, const float* x, size_t n, size_t gs);
/* functional.rs:48-78 */
int lmrs_op_rmsnorm(int device, float* o, const float* x, const float* weight, size_t size, float eps, int add_unit_offset);
/* functional.rs:122-140 (in place) */
int lmrs_op_softmax(int device, float* x, size_t n);
/* The decode step's classifier launch + final argmax on caller-supplied rows: final rmsnorm + quantize + matmul_q8 (transformer.rs:341-381)
 * and Sampler::sample_argmax (sampler.rs:29-41: starts at index 0, moves on a strict `>`: first index of the maximum; a NaN at index 0
 * is never displaced, NaNs elsewhere never win).  logits (optional): the o logits. */
int lmrs_op_classifier_argmax(int device, const float* x, const float* rms_w, const int8_t* wq, const float* ws, size_t n, size_t o,
                              float eps, uint32_t* token, float* logits);
/* f32::exp as used by softmax (functional.rs:133) and SiLU (transformer.rs:617): the device's bit-exact restatement of glibc expf. */
int lmrs_op_expf(int device, float* y, const float* x, size_t n);
/* y = (float)tanh(c * (double)x): f64::tanh as the reference calls it for Gemma's soft-caps (transformer.rs:520-522, 377-379; c = 1) and
 * the tanh-GELU (transformer.rs:614; c = 0.7978845608028654) - the device's f64 tanh, for comparison with the host libm (oracle/tanh_check.c). */
int lmrs_op_tanh_cast(int device, float* y, const float* x, size_t n, double c);
/* Sampler::sample (sampler.rs:109-129) for temperature != 0 and sample_mult on caller-supplied logits through lmrs_forward_sample's route (scaling,
 * maximum and exponentials on the device, the two sequential chains on the host): they are scaled and softmax-ed IN PLACE (as the reference does
 * to the slice) and *token = the draw for the random number rnd.  Unit parity for lmrs_forward_sample. */
int lmrs_op_sample_mult(int device, float* logits, size_t n, float temperature, float rnd, uint32_t* token);

/* ---- measurement hooks (bench.py) ---------------------------------------------------
 * Runs, `iters` times, the dequant-GEMV launches of ONE decode step in step order (per layer: qkv, wo,
 * w1w3, w2; then the classifier) so the weight stream is the real one, each launch bracketed by HIP events
 * on the context's stream.  Index k of the outputs: 0 qkv, 1 wo, 2 w1w3, 3 w2, 4 classifier;
 * us5[k] = summed duration (microseconds), bytes5[k] = summed algorithmic bytes (int8 + f32 scales),
 * count5[k] = launches.  Activations hold garbage afterwards; weights and older KV rows are untouched. */
int lmrs_bench_gemv(lmrs_ctx* ctx, int iters, double* us5, double* bytes5, int* count5);
/* The real decode step at position `pos` (then pos+1, ...) replayed eagerly `iters` times from the context's live state with HIP events on
 * every dispatch: per-kernel durations as they occur INSIDE the step.  kind k: 0 qkv, 1 attention, 2 wo, 3 w1w3, 4 w2, 5 classifier,
 * 6 argmax + next embedding row, 7 glue launches of the sharded / unfused forms, 8 peer-to-peer exchanges (time waiting for the peers
 * included); us9[k] = summed duration, bytes9[k] = summed algorithmic bytes, count9[k] = launches.  On a row-sharded context every
 * rank calls it together.  Measurement aid, no reference counterpart; the decode state advances by `iters` + 1 valid greedy steps. */
int lmrs_bench_step(lmrs_ctx* ctx, uint32_t pos, int iters, double* us9, double* bytes9, int* count9);
/* Debug timeline (LMRS_DEBUG_TIMELINE=1 in the environment at lmrs_create): 8 wall-clock stamps (100 MHz) per
 * kernel of the last decode step, in launch order: [0..3] first workgroup, [4..7] last workgroup:
 * start, prologue done, first rows done, end. */
int lmrs_debug_timeline(lmrs_ctx* ctx, unsigned long long* out, int max_nodes, int* n_nodes);
/* Verification aid (no reference counterpart; the reference's key_cache / value_cache are private, transformer.rs:302-303): one row of
 * the KV cache as the reference lays it out (which: 0 key, 1 value; kv_dim floats of `layer` at `pos`). */
int lmrs_debug_kv(lmrs_ctx* ctx, int which, uint32_t layer, uint32_t pos, float* out);
/* Device time (ms, HIP events) the last BATCHED lmrs_fill_kv_cache of this context spent between the upload of its embeddings and the
 * download of the residual stream: forward_layer(sl = n) itself.  Measurement aid (bench.py's `prefill` object), no reference counterpart. */
int lmrs_last_fill_ms(const lmrs_ctx* ctx, double* ms);
/* Fault injection for the tests of the multi-GPU paths (no reference counterpart; explicit calls, nothing is read from the environment):
 *   what = 0: the next lmrs_p2p_connect of this context fails ("injected failure"), so that a launcher's "every rank falls back
 *             together" logic can be exercised;
 *   what = 1: a row-sharded context enqueues its steps eagerly from now on and spins `b` microseconds on the device right after the
 *             exchange that follows segment `a` of every step (4 * n_layers = the argmax partials) - a peer that runs ahead then pushes
 *             its next block while this shard has not consumed the current one. */
int lmrs_debug_inject(lmrs_ctx* ctx, int what, int a, int b);
/* Number of kernel launches per decode step and the sum of algorithmic bytes per step at `pos` (the byte model of SURVEY.md §8d;
 * measurement aid, no reference counterpart). */
int lmrs_step_info(const lmrs_ctx* ctx, uint32_t pos, int* n_launches, double* algo_bytes);
/* The output tile (weight rows x tokens per workgroup, waves per workgroup) the batched matmul_q8 / matmul_q4 (functional.rs:173-250 with
 * sl = n_tok) runs a launch of `o` rows over K = `n` with: the cost model of DESIGN.md section 4.1, host arithmetic only (no device is
 * touched).  0 x 0: fewer than 48 tokens - the direct kernels.  Inspection aid, no reference counterpart. */
int lmrs_debug_gemm_tile(uint32_t n, uint32_t o, uint32_t n_tok, int q4, int* tile_rows, int* tile_tokens, int* waves);
/* The batched w1 / w3 projection with the activation and the NEXT matmul's quantiser in its epilogue, as fill_kv_cache runs it from a few
 * hundred tokens on (transformer.rs:588-630 with sl = n_tok: matmul_q8, SiLU(gate) * up or GELU(gate) * up, quantize): `wq` holds o rows of n
 * int8 with gate / up rows interleaved (row 2i = w1's row i, row 2i + 1 = w3's), hq receives n_tok x o/2 int8 and hs n_tok x o/256 scales.
 * Returns -1 with a message when the shape does not take the fused epilogue (lmrs_debug_gemm_tile: fewer than 128 rows x 128 tokens per
 * tile).  Unit-parity aid, no reference counterpart. */
int lmrs_debug_w13_quant(int device, int8_t* hq, float* hs, const int8_t* xq, const float* xs, const int8_t* wq, const float* ws,
                         size_t n, size_t o, size_t n_tok, int gemma);

/* ---- CLIP image tower of the multimodal models  (src/vision.rs) ---------------------------
 * lmrs_vision_create   <- VisionTransformer::new(data) -> (VisionTransformer, usize)   vision.rs:99-243
 *   `section` points at the vision section of an LMRS multimodal file (offset = *bytes_consumed of lmrs_create);
 *   *bytes_consumed = size of the section (the processor section follows).  CLIP ViT-L/14-336 geometry; Q8_0 (the tuned path),
 *   Q4_0 (quantize_q4 rows, matmul_q4 from packed nibbles on the matrix cores) and q_type None (the f32 `matmul`, correct, not tuned).
 * lmrs_vision_forward  <- VisionTransformer::forward(pixel_values, num_crops) -> (Vec<f32>, u32)   vision.rs:244-577
 *   pixel_values: num_crops * 3 * image_size^2 floats, normalised and cut into patches as PHI3VProcessor::process
 *   produces them; out: num_crops * 576 * dim floats (class token dropped); *new_shape = 576 * dim.
 *   Projections run as int8 matrix-core GEMMs over all tokens of all crops; the f32x8 lane structure of the
 *   reference's layernorm / matmul_rest sums (and the patch-embedding tail quirk) is reproduced: same values. */
typedef struct lmrs_vision lmrs_vision;
int lmrs_vision_create(const uint8_t* section, size_t len, int device, lmrs_vision** out, size_t* bytes_consumed);
void lmrs_vision_destroy(lmrs_vision* v);
int lmrs_vision_forward(lmrs_vision* v, const float* pixel_values, uint32_t num_crops, float* out, uint32_t* new_shape);

/* --------------------------------------------------------------------------------------------------------------------------
 * Image projector of the multimodal models (reference src/processor.rs, struct PHI3VProcessor).  Q8_0, Q4_0 and unquantised sections.
 *
 * lmrs_processor_create   <- PHI3VProcessor::new(data) -> PHI3VProcessor              processor.rs:168-232
 *     section = the bytes that follow the vision tower's section in the model file (128-byte header: hidden_dim, text_dim,
 *     q_type, group_size; glb_GN, sub_GN, the two projections, their biases).
 * lmrs_processor_forward  <- PHI3VProcessor::forward(out_patches, new_shape, patch_side, w_crop, h_crop) -> (Vec<f32>, u32)
 *                                                                                       processor.rs:234-342
 *     out_patches = lmrs_vision_forward's output (global crop first), total_floats floats; the HD transform
 *     (reshape_hd_patches_2x2merge :377-418, add_image_newline :480-484), the separators and the two-layer tanh-GELU MLP;
 *     out receives n_embeds * text_dim floats, n_embeds = (h_crop*12)*(w_crop*12+1) + 12*13 + 1.
 * lmrs_processor_destroy  <- Drop
 */
typedef struct lmrs_processor lmrs_processor;
int lmrs_processor_create(const uint8_t* section, size_t len, int device, lmrs_processor** out, size_t* bytes_consumed);
void lmrs_processor_destroy(lmrs_processor* p);
int lmrs_processor_forward(lmrs_processor* p, const float* out_patches, uint32_t total_floats, uint32_t new_shape, uint32_t patch_side,
                           uint32_t w_crop, uint32_t h_crop, float* out, uint32_t* n_embeds);
/* Host-only verification aid (works without a GPU): the rows lmrs_processor_forward feeds to the projector - the HD transform
 * reshape_hd_patches_2x2merge + add_image_newline (processor.rs:377-418, 480-484) of the sub-images, glb_GN, the same of the
 * global crop (:240-254) - for given separators.  out: n_embeds * 4096 floats. */
int lmrs_processor_hd_transform(const float* out_patches, uint32_t total_floats, uint32_t new_shape, uint32_t w_crop, uint32_t h_crop,
                                const float* glb_gn, const float* sub_gn, float* out, uint32_t* n_embeds);

/* Host-only verification aid (works without a GPU): the (cos, sin) pair lmrs_create tabulates for position `pos` and pair `j` of a
 * head (j < head_size / 2) - the RoPE frequency arithmetic of transformer.rs:446-477 (Llama-3 wavelength scaling, Phi LongRoPE short
 * factors and magnitude) for the model family / rope_theta / head_size in `args`.  Lets the tests check this host function against a
 * transcription that shares no code with it. */
int lmrs_rope_terms(const lmrs_args* args, uint32_t pos, uint32_t j, float* fcr, float* fci);

/* --------------------------------------------------------------------------------------------------------------------------
 * The callers either side of the device path, HOST code with the reference's exact results (no GPU needed):
 *
 * lmrs_tokenizer_create   <- Tokenizer::new(path)   src/tokenizer.rs:24-64   (data = the bytes of tokenizer.bin)
 * lmrs_tokenizer_encode   <- Tokenizer::encode(text, bos, eos, chat_format, model_type) -> Vec<u32>   :66-151
 *     one id per character (or its UTF-8 bytes + 3), then greedy merging of the best-scoring adjacent pair; chat_format wraps the
 *     ids in the model family's hard-coded template ids.  model_type: 0 GEMMA, 1 LLAMA, 2 PHI.  *n = ids produced (<= cap).
 * lmrs_tokenizer_decode   <- Tokenizer::decode(token) -> String   :153-163   (bytes, not NUL-terminated; "<0xHH>" -> U+00HH)
 * lmrs_tokenizer_info     <- the pub fields bos / eos (:15-16) and vocab_size
 *
 * lmrs_sampler_create     <- Sampler::new(vocab_size, temperature, top_p, seed)   src/sampler.rs:19-27
 * lmrs_sampler_sample     <- Sampler::sample(&mut logits) -> u32   :109-129: temperature 0 -> sample_argmax (:29-41); otherwise the
 *     logits are divided by the temperature and softmax-ed IN PLACE (functional.rs:122-140, one sequential sum - which is why this
 *     stays on the host: see lmrs_text.cpp) and sample_mult (:43-55) or sample_topp (:67-106) draws with random_f32(seed)
 *     (functional.rs:34-44).  The reference never advances the seed (:119) - every call of one Sampler uses the same random
 *     number - and sorts its whole candidate vector, stale entries included (:81): both reproduced. */
typedef struct lmrs_tokenizer lmrs_tokenizer;
int lmrs_tokenizer_create(const uint8_t* data, size_t len, lmrs_tokenizer** out);
void lmrs_tokenizer_destroy(lmrs_tokenizer* t);
int lmrs_tokenizer_info(const lmrs_tokenizer* t, uint32_t* vocab_size, uint32_t* bos, uint32_t* eos);
int lmrs_tokenizer_encode(lmrs_tokenizer* t, const char* text, size_t text_len, int bos, int eos, int chat_format, int model_type,
                          uint32_t* out, size_t cap, size_t* n);
int lmrs_tokenizer_decode(const lmrs_tokenizer* t, uint32_t token, char* out, size_t cap, size_t* n);
typedef struct lmrs_sampler lmrs_sampler;
int lmrs_sampler_create(uint32_t vocab_size, float temperature, float top_p, uint64_t seed, lmrs_sampler** out);
void lmrs_sampler_destroy(lmrs_sampler* s);
int lmrs_sampler_sample(lmrs_sampler* s, float* logits, uint32_t* next);
/* sample_topp (sampler.rs:67-106) from its sort on, for a caller that ran the temperature scaling, the softmax and the cutoff filter
 * elsewhere: pairs = n0 candidates {f32 prob, u32 index} with prob >= (1 - top_p) / (vocab_size - 1), in index order - what :74-80 leaves
 * in probindex[0 .. n0).  A stand-alone host entry point: the library itself goes through lmrs_sampler_exps_prepare / _finish. */
int lmrs_sampler_topp_pairs(lmrs_sampler* s, const void* pairs, size_t n0, uint32_t* next);
/* Sampler::sample from the softmax's exponentials on (functional.rs:134-139, then sampler.rs:119-128): exps[i] = exp(logits[i] / temperature - max)
 * were formed elsewhere (lmrs_forward_sample forms them on the device); the sequential sum, the division, and sample_mult / sample_topp run here.
 * exps become the probabilities in place.  Same token and probabilities as lmrs_sampler_sample on the logits. */
int lmrs_sampler_sample_exps(lmrs_sampler* s, float* exps, uint32_t* next);
/* The same in two halves, for a caller that sorts sample_topp's candidates itself (lmrs_forward_sample: on the device, when most of the vocabulary
 * passes the cutoff).  prepare: the sequential sum (functional.rs:134), the division (:137-139; exps become the probabilities in place) and, for a
 * top-p sampler, the cutoff filter of sampler.rs:71-80 - *n0 candidates are left in the sampler's vector in index order (0 for sample_mult).
 * finish: sorted_pairs = NULL: the reference's sort (:81) runs here; else the n0 candidates {f32 prob, u32 index} sorted by descending prob,
 * ties in index order - exactly what :81 makes of them; then the merge with the stale rest of the vector, the cumulative cut and the draw. */
int lmrs_sampler_exps_prepare(lmrs_sampler* s, float* exps, float* sum, float* cutoff, size_t* n0);
int lmrs_sampler_exps_finish(lmrs_sampler* s, const float* probs, const void* sorted_pairs, uint32_t* next);
int lmrs_sampler_info(const lmrs_sampler* s, uint32_t* vocab_size, float* temperature, float* top_p, float* rnd);   /* rnd = random_f32(seed), the same on every call (:119) */
/* Transformer::forward (src/transformer.rs:316) followed by Sampler::sample (src/sampler.rs:109-129) without the host ever touching the
 * logits: temperature 0 -> the argmax fused into the decode step; temperature != 0 -> the parallel part of the sampler on the device -
 * logits / temperature (:115), the maximum and exp(x - max) (functional.rs:126-133) - then the vocab_size exponentials cross to the host,
 * where lmrs_sampler_exps_prepare / _finish run the reference's sequential chains (the softmax sum, the running cdf) and sample_mult (:43-55) or
 * sample_topp (:67-106, the reference's default, chat.rs:28-31) over the sampler's persistent candidate vector; when more than 4096 candidates pass
 * the cutoff (a flat distribution) their sort (:81) runs on the device - a bitonic network over the unique keys (prob, index): the same permutation
 * as the reference's stable sort (7.05 -> 0.9 ms per token with 100 k candidates, profiles/r6_sampler_rate.txt).  Round 4 ran the chains
 * on the device as well (one wave, lane by lane): 1106 us per token against 1020 for the host sampler on copied logits - a dependent f32
 * add is ~1 ns on a host core and ~2.5 ns on one GPU lane; this split was measured at profiles/r5_sampler_rate.txt.  Same token as
 * lmrs_forward + lmrs_sampler_sample in every case.  One-GPU contexts (sharded ones copy the gathered logits). */
int lmrs_forward_sample(lmrs_ctx* ctx, uint32_t token, uint32_t pos, lmrs_sampler* sampler, uint32_t* next);

#ifdef __cplusplus
}
#endif
#endif /* LMRS_HIP_H */
