// lmrs_kernels.hip — hand-written HIP kernels (CDNA4 / gfx950, wave64) for the lm.rs decode hot path.
//
// What each kernel replaces in the reference (samuel-vitorino/lm.rs):
//   gemv_kernel      src/functional.rs:173-214 matmul_q8 / :216-250 matmul_q4, with the producers and
//                    consumers that surround every call fused in:
//                      prologue  rmsnorm (functional.rs:48-78) + quantize (quantization.rs:44-95)
//                      epilogue  residual add (transformer.rs:574-576, 652-654), K/V-cache store
//                                (:413-416), SiLU*up / GELU*up (:607-624), logits + argmax partials
//                                (:355-381, sampler.rs:29-41)
//   attention_kernel src/transformer.rs:443-544 (RoPE, scores, softmax, weighted V sum)
//   embed_kernel     src/transformer.rs:324-332 (+ quantization.rs:25-42 dequantize of one row)
//   argmax_final     src/sampler.rs:29-41
//
// Arithmetic contract: every float result is bit-identical to the CPU path:
//   integer group sums are exact in any order; float accumulation follows the reference's order
//   (groups ascending per row; 8 strided partials + wide's reduce tree for RMSNorm; sequential over
//   t for softmax sum and the V accumulation); no FMA contraction (-ffp-contract=off); IEEE div/sqrt.
//
// Mapping for MI355X: decode GEMV is HBM-bound (1 int8 MAC per weight byte), so the design goal is
// "every weight byte crosses HBM once, 16 B per lane, as many bytes in flight as the chip accepts":
//   - a row is read by L lanes (L = 8..64) x 16 B per step; an aligned cluster of 8 lanes (Q8_0) or
//     4 lanes (Q4_0) covers exactly one 128-element quantisation group, so the group's int32 sum is
//     a 3-step (2-step) DPP butterfly and the per-group float combine needs no cross-lane traffic
//     when L == cluster size; U steps (<= 16 KiB per wave) are issued back to back before first use;
//   - weights are loaded non-temporally (read once); the quantised activation vector lives in LDS
//     (n bytes + n/128 scales), read with conflict-free ds_read_b128 broadcasts;
//   - the activation prologue (RMSNorm / quantise) is recomputed by every workgroup from the
//     L2-resident f32 vector: it runs while the workgroup's first weight loads are in flight and
//     saves a dependent kernel boundary (~1.2-1.9 us on this chip) per use.
#include <hip/hip_ext.h>
#include <mutex>
#include <map>
#include <set>
#include <utility>

#include "lmrs_device_math.h"
#include "lmrs_kernels.h"
#include "lmrs_stage.h"

namespace lmrs {

constexpr int kBlock = 256;          // 4 waves
constexpr int kGS = 128;             // quantisation group size (the reference exporter always uses 128)

// ------------------------------------------------------------------------------------------------
// Activation prologue device functions (shared by the fused GEMV and the lmrs_op_* kernels)
// ------------------------------------------------------------------------------------------------

// Thread t owns elements e = i*1024 + 4t .. +3 for i = 0..P-1  (P = ceil(n / 1024)), so that 32
// consecutive lanes own one 128-element quantisation group.
constexpr int kMaxP = 16;            // n <= 16384 (NP, the per-lane float4 count, is a template parameter <= kMaxP)

// RMSNorm (functional.rs:48-78) of x[n] (held in v[]), result written back into v[].
// wv[] = the norm weights of the same elements (loaded by the caller, early, so that no late global
// load sits behind the weight stream in the in-order vmcnt queue).
// scratch: (8 * (n/8 + 4) + 1) floats of LDS.
template <int NP>
__device__ __forceinline__ void rmsnorm_inplace(float4 (&v)[NP], const float4 (&wv)[NP], int n, float eps,
                                                int add_unit, float* scratch) {
    const int t = threadIdx.x;
    const int P = (n + 1023) >> 10;
    const int JP = (n >> 3) + 4;                 // padded row length of the transposed square table
    // squares, transposed: T[k][j] = x[8j+k]^2 so that lane k walks j contiguously
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < P) {
            const int e = i * 1024 + t * 4;
            if (e < n) {
                const int j = e >> 3, k0 = e & 7;          // k0 in {0, 4}
                scratch[(k0 + 0) * JP + j] = v[i].x * v[i].x;
                scratch[(k0 + 1) * JP + j] = v[i].y * v[i].y;
                scratch[(k0 + 2) * JP + j] = v[i].z * v[i].z;
                scratch[(k0 + 3) * JP + j] = v[i].w * v[i].w;
            }
        }
    }
    lds_barrier();
    if (t < 64) {
        // lanes 0..7: the 8 strided partial sums, each a sequential chain over j (ss_sim += x*x).
        // The adds are inherently serial (float addition is not associative); the LDS reads are not, so
        // they are issued 4 x 16 B ahead of the chain (two register batches, ping-pong).
        float p = 0.0f;
        if (t < 8) {
            const float4* row = reinterpret_cast<const float4*>(scratch + t * JP);
            const int nj4 = n >> 5;                          // (n/8)/4 float4 per row  (n % 32 == 0)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);   // p >= +0, so p + 0 == p exactly: padding is free
            float4 A[4], B[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) A[u] = u < nj4 ? row[u] : z4;
            for (int j0 = 0; j0 < nj4; j0 += 8) {
#pragma unroll
                for (int u = 0; u < 4; ++u) B[u] = (j0 + 4 + u) < nj4 ? row[j0 + 4 + u] : z4;
#pragma unroll
                for (int u = 0; u < 4; ++u) { p = p + A[u].x; p = p + A[u].y; p = p + A[u].z; p = p + A[u].w; }
#pragma unroll
                for (int u = 0; u < 4; ++u) A[u] = (j0 + 8 + u) < nj4 ? row[j0 + 8 + u] : z4;
#pragma unroll
                for (int u = 0; u < 4; ++u) { p = p + B[u].x; p = p + B[u].y; p = p + B[u].z; p = p + B[u].w; }
            }
        }
        const float p0 = __shfl(p, 0), p1 = __shfl(p, 1), p2 = __shfl(p, 2), p3 = __shfl(p, 3);
        const float p4 = __shfl(p, 4), p5 = __shfl(p, 5), p6 = __shfl(p, 6), p7 = __shfl(p, 7);
        if (t == 0) {
            float ss = reduce_add8(p0, p1, p2, p3, p4, p5, p6, p7);
            ss = ss / (float)n;
            ss = ss + eps;
            ss = 1.0f / sqrtf(ss);
            scratch[8 * JP] = ss;
        }
    }
    lds_barrier();
    const float ss = scratch[8 * JP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < P) {
            const int e = i * 1024 + t * 4;
            if (e < n) {
                float4 o;
                if (add_unit) {
                    o.x = (1.0f + wv[i].x) * (ss * v[i].x); o.y = (1.0f + wv[i].y) * (ss * v[i].y);
                    o.z = (1.0f + wv[i].z) * (ss * v[i].z); o.w = (1.0f + wv[i].w) * (ss * v[i].w);
                } else {
                    o.x = wv[i].x * (ss * v[i].x); o.y = wv[i].y * (ss * v[i].y);
                    o.z = wv[i].z * (ss * v[i].z); o.w = wv[i].w * (ss * v[i].w);
                }
                v[i] = o;
            }
        }
    }
}

// quantize (quantization.rs:44-67) / quantize_q4 (:69-95) of the vector held in v[] into LDS.
//   Q8_0: xq[e] = int8                                     (n bytes)
//   Q4_0: xq holds the UNPACKED signed nibble values (q-8), de-interleaved per 8 elements as
//         [e0 e2 e4 e6 | e1 e3 e5 e7] so that they line up with (w & 0x0F0F0F0F) / (w >> 4 & ...).
// If gq/gs_out are non-null (lmrs_op_quantize), block 0 also writes the reference's packed form.
template <bool Q4, int NP>
__device__ __forceinline__ void quantize_to_lds(const float4 (&v)[NP], int n, int8_t* xq, float* xs, void* gq, float* gs_out) {
    const int t = threadIdx.x;
    const int P = (n + 1023) >> 10;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < P) {
            const int e = i * 1024 + t * 4;
            const bool live = e < n;
            float m = 0.0f;
            if (live) m = fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
            m = group32_max(m);                       // wmax of the 128-group (max is order-free)
            if (live) {
                if constexpr (!Q4) {
                    const float scale = m / 127.0f, inv = __builtin_amdgcn_rcpf(scale);
                    float dev = 0.0f;
                    int q0 = quant_q8_try(v[i].x, inv, dev), q1 = quant_q8_try(v[i].y, inv, dev);
                    int q2 = quant_q8_try(v[i].z, inv, dev), q3 = quant_q8_try(v[i].w, inv, dev);
                    if (quant_slow(m, dev)) { q0 = quant_q8(v[i].x, scale); q1 = quant_q8(v[i].y, scale); q2 = quant_q8(v[i].z, scale); q3 = quant_q8(v[i].w, scale); }
                    const unsigned packed = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
                    *reinterpret_cast<unsigned*>(xq + e) = packed;
                    if ((t & 31) == 0) xs[e >> 7] = scale;
                    if (gq) {
                        *reinterpret_cast<unsigned*>(reinterpret_cast<int8_t*>(gq) + e) = packed;
                        if ((t & 31) == 0) gs_out[e >> 7] = scale;
                    }
                } else {
                    const float scale = m / -8.0f;
                    const unsigned a = quant_q4(v[i].x, scale), b = quant_q4(v[i].y, scale);
                    const unsigned c = quant_q4(v[i].z, scale), d = quant_q4(v[i].w, scale);
                    // elements e..e+3 sit at in-octet positions k0..k0+3 (k0 = e & 7 in {0,4})
                    const int base = e & ~7, k0 = e & 7;
                    xq[base + (k0 >> 1) + 0] = (int8_t)((int)a - 8);        // even element -> low-nibble lane
                    xq[base + 4 + (k0 >> 1) + 0] = (int8_t)((int)b - 8);    // odd element  -> high-nibble lane
                    xq[base + (k0 >> 1) + 1] = (int8_t)((int)c - 8);
                    xq[base + 4 + (k0 >> 1) + 1] = (int8_t)((int)d - 8);
                    if ((t & 31) == 0) xs[e >> 7] = scale;
                    if (gq) {
                        uint8_t* g8 = reinterpret_cast<uint8_t*>(gq);
                        g8[(e >> 1) + 0] = (uint8_t)(a | (b << 4));
                        g8[(e >> 1) + 1] = (uint8_t)(c | (d << 4));
                        if ((t & 31) == 0) gs_out[e >> 7] = scale;
                    }
                }
            }
        }
    }
}

template <int NP>
__device__ __forceinline__ void load_vec(float4 (&v)[NP], const float* __restrict__ x, int n) {
    const int t = threadIdx.x;
    const int P = (n + 1023) >> 10;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < P) {
            const int e = i * 1024 + t * 4;
            if (e < n) v[i] = *reinterpret_cast<const float4*>(x + e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Embedding row element (transformer.rs:324-332; quantization.rs:25-42) - dequantised on the fly, which is bit-identical to reading
// the reference's load-time f32 copy of the table.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dequant_elem(const void* q, const float* s, int q4, size_t idx) {     // q4: 0 Q8_0, 1 Q4_0, 2 f32 table (q_type None)
    if (q4 == 2) return reinterpret_cast<const float*>(q)[idx];
    if (!q4) return (float)reinterpret_cast<const int8_t*>(q)[idx] * s[idx / kGS];
    const int8_t v = reinterpret_cast<const int8_t*>(q)[idx >> 1];
    const int nib = (idx & 1) ? ((v >> 4) & 0x0F) - 8 : (v & 0x0F) - 8;
    return (float)nib * s[idx / kGS];
}

// sample_argmax (sampler.rs:29-41) over per-thread candidates (larger value wins, ties -> lower index = first index of the maximum),
// then token feedback, position advance and the embedding row of the next input token.  Whole workgroup (kBlock threads); shared by
// the stand-alone argmax_final_kernel and the classifier's last-arriving workgroup (ClsTail).
// pre: the step state the tail needs (position, end of the prompt, the prompt token that may follow), loaded by the caller BEFORE its
// long-running work when it can (the classifier: at kernel start), so that the tail is not a chain of dependent memory round trips.
struct TailPre { int pos, prompt_end; uint32_t tok_next; };
__device__ __forceinline__ TailPre tail_preload(const ArgmaxArgs& a) {
    TailPre p; p.pos = a.st->pos; p.prompt_end = a.st->prompt_end; p.tok_next = a.tokens[p.pos + 1]; return p;
}
__device__ __forceinline__ void argmax_tail(const ArgmaxArgs& a, float best, int best_i, int nan0, const TailPre& pre) {
    __shared__ float sv[kBlock / 64];
    __shared__ int si[kBlock / 64];
    __shared__ uint32_t s_next;
    if (threadIdx.x == 0 && a.tail_row > 0 && (0.0f > best || (0.0f == best && a.tail_row < best_i))) { best = 0.0f; best_i = a.tail_row; }   // the unwritten last rows: 0.0, the first of them wins their tie
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {                   // wave: larger value, ties -> lower index
        const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(best_i, off);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = best_i; }
    nan0 = __syncthreads_or(nan0);
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < best_i)) { best = sv[w]; best_i = si[w]; }
        int win = best_i;
        if (nan0 || win == 0x7fffffff) win = 0;             // NaN at index 0 is never displaced (strict >); nothing above -inf: index 0
        const int pos = pre.pos;
        uint32_t next = (uint32_t)win;
        if (pos + 1 >= pre.prompt_end) a.tokens[pos + 1] = next;     // chat.rs:188-193: sampler output ignored while the prompt lasts
        else next = pre.tok_next;
        a.st->pos = pos + 1;
        a.st->step_count += 1;
        if (a.seq) *a.seq += 1u;
        s_next = next;
    }
    __syncthreads();
    // embedding row of the next input token (transformer.rs:324-332) for the next replay of the step graph
    const uint32_t token = s_next;
    if (a.emb.q4 == 0 && a.emb.dim > 0 && (a.emb.dim & 7) == 0 && a.emb.dim <= 8 * 4 * kBlock) {
        // Q8_0 rows: 8 elements per thread and pass, ALL loads of the row issued before the first use (the generic loop below is one
        // dependent memory round trip per element: 8 of them for a 2048-wide row, 4 us on the critical path of every step)
        const int8_t* qrow = reinterpret_cast<const int8_t*>(a.emb.emb_q) + (size_t)token * a.emb.dim;
        const float* srow = a.emb.emb_s + ((size_t)token * a.emb.dim) / kGS;       // dim % 128 == 0: the row's scales are contiguous
        uint2 q8[4]; float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * kBlock + (int)threadIdx.x) * 8;
            const bool live = e < a.emb.dim;
            q8[k] = *reinterpret_cast<const uint2*>(qrow + (live ? e : 0));
            sc[k] = srow[(live ? e : 0) / kGS];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * kBlock + (int)threadIdx.x) * 8;
            if (e < a.emb.dim) {
                float o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = (int)(int8_t)(((u < 4 ? q8[k].x : q8[k].y) >> (8 * (u & 3))) & 0xff);
                    float v = (float)b * sc[k];
                    if (a.emb.do_scale) v = v * a.emb.scale;
                    o[u] = v;
                }
                *reinterpret_cast<float4*>(a.emb.x + e) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(a.emb.x + e + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    } else if (a.emb.q4 == 1 && a.emb.dim > 0 && (a.emb.dim & 7) == 0 && a.emb.dim <= 8 * 4 * kBlock) {
        // Q4_0 rows the same way: 8 elements = 4 packed bytes per thread and pass, every load of the row in flight before the first use
        // (even element = low nibble; value = (nibble - 8) * scale, quantization.rs:25-42)
        const uint8_t* qrow = reinterpret_cast<const uint8_t*>(a.emb.emb_q) + ((size_t)token * a.emb.dim) / 2;
        const float* srow = a.emb.emb_s + ((size_t)token * a.emb.dim) / kGS;
        unsigned q4w[4]; float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * kBlock + (int)threadIdx.x) * 8;
            const bool live = e < a.emb.dim;
            q4w[k] = *reinterpret_cast<const unsigned*>(qrow + (live ? e : 0) / 2);
            sc[k] = srow[(live ? e : 0) / kGS];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (k * kBlock + (int)threadIdx.x) * 8;
            if (e < a.emb.dim) {
                float o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int nib = (int)((q4w[k] >> (4 * u)) & 0xfu) - 8;
                    float v = (float)nib * sc[k];
                    if (a.emb.do_scale) v = v * a.emb.scale;
                    o[u] = v;
                }
                *reinterpret_cast<float4*>(a.emb.x + e) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(a.emb.x + e + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    } else {
        for (int i = threadIdx.x; i < a.emb.dim; i += kBlock) {
            float v = dequant_elem(a.emb.emb_q, a.emb.emb_s, a.emb.q4, (size_t)token * a.emb.dim + i);
            if (a.emb.do_scale) v = v * a.emb.scale;
            a.emb.x[i] = v;
        }
    }
}

// The classifier launch with the argmax folded in (ClsTail).  Every workgroup publishes its partial as ONE 8-byte write-through word
// {value | nan-at-0 flag, 11-bit tag, 20-bit index} - the data is the flag, as in the merged qkv + attention launch - and workgroup
// 0, once its own rows are done, sweeps the words until every tag is this step's, reduces them and finishes the step (token
// feedback, position advance, next embedding row).  Measured alternatives: an arrival ticket (512 arrivals on one counter serialise
// at ~12 ns each exactly when the bandwidth-bound launch ends and everybody arrives together: as slow as the separate argmax launch
// it replaced); a dedicated consumer workgroup polling from the start (its sweeps return behind the weight tiles of the GEMV workgroup
// it shares a CU with - a CU answers its loads in request order: the last partial was seen 4 us late).  Nobody waits for workgroup 0;
// its sweeps are bounded (err).
__device__ __forceinline__ unsigned cls_tag(const GemvArgs& a) { return (*a.tail.cls_seq + 1u) & 0x7ffu; }
__device__ __forceinline__ void cls_publish(const GemvArgs& a, int bid, unsigned tag, float best, int best_i) {
    // index field: 0xfffff = "nothing above -inf in my rows" (best_i = 0x7fffffff); the host admits vocabularies below 2^20 - 1 only
    const unsigned hi = best_i < 0 ? (0x80000000u | (tag << 20)) : ((tag << 20) | (best_i == 0x7fffffff ? 0xfffffu : (unsigned)best_i));
    __hip_atomic_store(a.tail.part_pk + bid, ((unsigned long long)hi << 32) | __float_as_uint(best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cls_consumer(const GemvArgs& a, int nprod, const TailPre& pre, unsigned tag) {
    if (a.dbg && threadIdx.x == 0) a.dbg[4] = wall_clock64();
    float bv; int bi, nan0;
    for (unsigned spins = 0;; ++spins) {
        bv = __uint_as_float(0xff800000u); bi = 0x7fffffff; nan0 = 0;
        int ok = 1;
        for (int i = threadIdx.x; i < nprod; i += kBlock) {
            const unsigned long long pk = __hip_atomic_load(a.tail.part_pk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned hi = (unsigned)(pk >> 32);
            if (((hi >> 20) & 0x7ffu) != tag) { ok = 0; continue; }
            if (hi & 0x80000000u) { nan0 = 1; continue; }
            const float v = __uint_as_float((unsigned)pk); const int idx = (hi & 0xfffffu) == 0xfffffu ? 0x7fffffff : (int)(hi & 0xfffffu);
            if (v > bv || (v == bv && idx < bi)) { bv = v; bi = idx; }
        }
        if (__syncthreads_and(ok)) break;
        if (spins > (1u << 18)) {                                // bounded: report and finish with garbage instead of hanging
            if (threadIdx.x == 0) __hip_atomic_store(a.tail.err, 1000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[5] = wall_clock64();
    argmax_tail(a.tail.m, bv, bi, nan0, pre);
    if (threadIdx.x == 0) *a.tail.cls_seq += 1u;                 // every workgroup of this launch read its tag at kernel start
    if (a.dbg && threadIdx.x == 0) a.dbg[7] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// Fused dequant-GEMV
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ i32x4 ld_nt(const i32x4* p) { return __builtin_nontemporal_load(p); }

// signed-nibble unpack of 4 packed bytes: per byte (v & 0xF) - 8 without cross-byte borrows
__device__ __forceinline__ int nib_signed(unsigned v) { return (int)((((v & 0x0F0F0F0Fu) | 0x80808080u) - 0x08080808u) ^ 0x80808080u); }

template <bool Q4>
__device__ __forceinline__ int group_partial_dot(const i32x4& w, const int8_t* xq_lds, int chunk) {
    int d = 0;
    if constexpr (!Q4) {
        const int4 x = *reinterpret_cast<const int4*>(xq_lds + chunk * 16);
        d = __builtin_amdgcn_sdot4(w.x, x.x, d, false);
        d = __builtin_amdgcn_sdot4(w.y, x.y, d, false);
        d = __builtin_amdgcn_sdot4(w.z, x.z, d, false);
        d = __builtin_amdgcn_sdot4(w.w, x.w, d, false);
    } else {
        // 16 weight bytes = 32 elements = 32 bytes of unpacked x: per dword [lo lanes | hi lanes]
        const int4 xa = *reinterpret_cast<const int4*>(xq_lds + chunk * 32);
        const int4 xb = *reinterpret_cast<const int4*>(xq_lds + chunk * 32 + 16);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.x), xa.x, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.x >> 4), xa.y, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.y), xa.z, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.y >> 4), xa.w, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.z), xb.x, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.z >> 4), xb.y, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.w), xb.z, d, false);
        d = __builtin_amdgcn_sdot4(nib_signed((unsigned)w.w >> 4), xb.w, d, false);
    }
    return d;
}

// PRO_PREQ input.  Contiguous (preq_slice == 0): xq_in = n int8 (n/2 packed bytes for Q4_0), xs_in = n/128 scales.  Sliced
// (row-sharded step, Q8_0): the activation was quantised by the shards that produced it and gathered as one block per shard,
// [preq_slice int8 | preq_slice / 128 f32 scales] every preq_block bytes; element e lives in block e / preq_slice.
__device__ __forceinline__ const int8_t* preq_bytes(const GemvArgs& a, int e) {
    const int8_t* base = reinterpret_cast<const int8_t*>(a.xq_in);
    if (a.preq_slice == 0) return base + e;
    const int r = e / a.preq_slice;
    return base + (size_t)r * a.preq_block + (e - r * a.preq_slice);
}
__device__ __forceinline__ float preq_scale(const GemvArgs& a, int g) {
    if (a.preq_slice == 0) return a.xs_in[g];
    const int e = g * 128, r = e / a.preq_slice;
    return *reinterpret_cast<const float*>(reinterpret_cast<const int8_t*>(a.xq_in) + (size_t)r * a.preq_block + a.preq_slice + ((e - r * a.preq_slice) >> 7) * 4);
}

// sample_argmax (sampler.rs:29-41) starts from probabilities[0] and only moves on a strict `>`: a NaN logit at index 0 is
// never displaced.  The workgroup that owns global row 0 (block 0 of the launch whose row_offset is 0) reports that case by
// storing index -1 in its partial; argmax_final_kernel - on every shard, the partials are gathered - then answers 0.
// (Called by thread 0 after the workgroup's __syncthreads(): the logit was stored by this workgroup and has reached L2.)
__device__ __forceinline__ int cls_flag_nan_at_zero(const GemvArgs& a, int best_i, int bid) {
    if (bid == 0 && a.row_offset == 0) {
        const float l0 = __hip_atomic_load(a.out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(l0 == l0)) return -1;
    }
    return best_i;
}

// Debug timeline: when a.dbg is set, lane 0 of the first and of the last workgroup record the 100 MHz
// wall clock at four points (start, prologue done, first pass consumed, end).
#define LMRS_STAMP(k)                                                                                         \
    do {                                                                                                      \
        if (a.dbg && threadIdx.x == 0) {                                                                      \
            if (blockIdx.x == 0) a.dbg[k] = wall_clock64();                                                   \
            if (blockIdx.x == gridDim.x - 1) a.dbg[4 + (k)] = wall_clock64();                                 \
        }                                                                                                     \
    } while (0)

// L lanes per row, U steps in flight, PRO/EPI fused stages, Q4 = packed-nibble weights.
// CL = lanes covering one quantisation group: 8 (Q8_0: 8 x 16 B = 128 B) or 4 (Q4_0: 4 x 16 B = 64 B).
// A row is split into NC = L / CL contiguous segments, one per cluster of CL lanes; a cluster walks the
// groups of its segment in ascending order, 16 B per lane per step (a full 128-B / 64-B line per cluster).
//   NC == 1: the row's float accumulation `acc += (isum * ws) * xs` runs lane-locally, groups ascending.
//   NC  > 1: every cluster first turns its (<= U) groups into products p, then the accumulation chain is
//            run segment by segment — cluster j starts from cluster j-1's running sum (one cross-lane move
//            per segment) — which is the reference's left-to-right order with NC-1 hops instead of G.
template <int L, int U, int NP, int PRO, int EPI, bool Q4>
__global__ __launch_bounds__(kBlock) void gemv_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CL = Q4 ? 4 : 8;
    constexpr int NC = L / CL;
    static_assert(L >= CL && (L & (L - 1)) == 0 && L <= 64, "bad L");
    LMRS_STAMP(0);
    const int bid = blockIdx.x, nblk = gridDim.x;
    unsigned ctag = 0; TailPre tpre{}; int pofs = 0;
    if constexpr (EPI == EPI_CLS) {
        if (a.has_tail) { ctag = cls_tag(a); if (bid == 0) tpre = tail_preload(a.tail.m); }     // (workgroup 0 will finish the step: ClsTail)
        else if (a.part_par) pofs = (int)((*a.part_par + 1u) & 1u) * a.part_par_floats;          // double-buffered partials (ArgmaxArgs)
    }
    const int n = a.n, G = n / kGS;
    int8_t* xq = reinterpret_cast<int8_t*>(smem);                       // n bytes
    float* xs = reinterpret_cast<float*>(smem + ((n + 15) & ~15));      // G floats
    float* scratch = xs + ((G + 3) & ~3);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane % L;                       // lane within the row
    const int cl = r / CL, rc = r % CL;           // cluster within the row, lane within the cluster
    constexpr int RW = 64 / L;                    // rows per wave
    constexpr int RB = RW * (kBlock / 64);        // rows per workgroup pass
    const int row_bytes = Q4 ? n / 2 : n;
    const int Gc = G / NC;                        // groups (= steps) per cluster; host guarantees G % NC == 0 (and Gc <= U if NC > 1)
    const int g0 = cl * Gc;                       // first group of my segment
    const int o = a.o;
    const int n_pass = (o + RB - 1) / RB;

    // ---------------- issue order matters: vmcnt retires in order, so the (small, latency-critical)
    // activation loads go first, then the first U weight steps of this workgroup's first rows; the
    // prologue's arithmetic then runs underneath the weight stream's HBM latency.
    float4 v[NP], nw[NP];
    if constexpr (PRO != PRO_PREQ) {
        load_vec(v, a.xin, n);
        if constexpr (PRO == PRO_RMS_QUANT) load_vec(nw, a.rms_w, n);
    }
    i32x4 w[U]; float sc[U];
    auto issue = [&](int pass, int s0) __attribute__((always_inline)) {
        int row = pass * RB + wave * RW + lane / L;
        row = row < o ? row : o - 1;
        const i32x4* wrow = reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(a.wq) + (size_t)row * row_bytes) + g0 * CL + rc;
        const float* srow = a.ws + (size_t)row * G + g0;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s0 + u < Gc) {
                w[u] = ld_nt(wrow + (s0 + u) * CL);
                sc[u] = srow[s0 + u];
            }
    };
    if (bid < n_pass) issue(bid, 0);

    // ---------------- prologue: build the quantised activation vector in LDS
    if constexpr (PRO == PRO_PREQ) {
        if constexpr (!Q4) {
            for (int e = threadIdx.x * 16; e < n; e += kBlock * 16)
                *reinterpret_cast<int4*>(xq + e) = *reinterpret_cast<const int4*>(preq_bytes(a, e));
        } else {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(a.xq_in);
            for (int b = threadIdx.x * 4; b < n / 2; b += kBlock * 4) {   // 4 packed bytes = one octet of elements
                const unsigned pk = *reinterpret_cast<const unsigned*>(src + b);
                *reinterpret_cast<int*>(xq + 2 * b) = nib_signed(pk);
                *reinterpret_cast<int*>(xq + 2 * b + 4) = nib_signed(pk >> 4);
            }
        }
        for (int g = threadIdx.x; g < G; g += kBlock) xs[g] = preq_scale(a, g);
    } else {
        if constexpr (PRO == PRO_RMS_QUANT) rmsnorm_inplace(v, nw, n, a.eps, a.add_unit, scratch);
        quantize_to_lds<Q4, NP>(v, n, xq, xs, nullptr, nullptr);
    }
    lds_barrier();
    LMRS_STAMP(1);

    // ---------------- main loop
    float best = __uint_as_float(0xff800000u); int best_i = 0x7fffffff;   // EPI_CLS
    bool preloaded = true;
    const int wlane = NC == 1 ? 0 : L - CL;       // lane (within the row) that owns the finished sum

    for (int pass = bid; pass < n_pass; pass += nblk) {
        const int row = pass * RB + wave * RW + lane / L;
        const bool valid = row < o;
        float acc = 0.0f;
        if constexpr (NC == 1) {
            for (int s0 = 0; s0 < Gc; s0 += U) {
                if (!preloaded) issue(pass, s0);
                preloaded = false;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (s0 + u < Gc) {
                        const int g = s0 + u;
                        int d = group_partial_dot<Q4>(w[u], xq, g * CL + rc);
                        if constexpr (Q4) { d += dpp_i<0xB1>(d); d += dpp_i<0x4E>(d); }
                        else d = cluster8_sum(d);
                        float p = (float)d * sc[u];                  // (ival as f32) * w.s[..]
                        p = p * xs[g];                               //   * x.s[..]
                        acc = acc + p;                               // xout += ..., groups ascending
                    }
            }
        } else {
            if (!preloaded) issue(pass, 0);
            preloaded = false;
            float pb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pb[u] = 0.0f;
                if (u < Gc) {
                    const int g = g0 + u;
                    int d = group_partial_dot<Q4>(w[u], xq, g * CL + rc);
                    if constexpr (Q4) { d += dpp_i<0xB1>(d); d += dpp_i<0x4E>(d); }
                    else d = cluster8_sum(d);
                    float p = (float)d * sc[u];
                    pb[u] = p * xs[g];
                }
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const float carry = j == 0 ? 0.0f : __shfl(acc, (lane & ~(L - 1)) + (j - 1) * CL);
                if (cl == j) {
                    acc = carry;
#pragma unroll
                    for (int u = 0; u < U; ++u) if (u < Gc) acc = acc + pb[u];
                }
            }
        }
        if (pass == bid) LMRS_STAMP(2);
        // ---------------- epilogue (the finished sum is replicated over the lanes of the row's last cluster)
        if constexpr (EPI == EPI_STORE) {
            if (valid && r == wlane) a.out[row] = acc;
        } else if constexpr (EPI == EPI_RESID) {
            if (valid && r == wlane) a.out[row] = a.out[row] + acc;
        } else if constexpr (EPI == EPI_QKV) {
            if (valid && r == wlane) {
                if (row < a.att_dim) a.out[row] = acc;
                else if (row < a.att_dim + a.kv_dim) a.k_raw[row - a.att_dim] = acc;
                else a.v_cache[((size_t)a.layer * a.seq_len + a.st->pos) * a.kv_dim + (row - a.att_dim - a.kv_dim)] = acc;
            }
        } else if constexpr (EPI == EPI_SWIGLU || EPI == EPI_GELU) {
            // rows are interleaved: 2i = gate (w1) row i, 2i+1 = up (w3) row i
            const float up = __shfl_down(acc, L);
            if (valid && r == wlane && ((lane / L) & 1) == 0) {
                float val = acc;
                if constexpr (EPI == EPI_SWIGLU) {
                    const float e = expf_glibc(-val);
                    const float g = 1.0f / (1.0f + e);
                    val = val * g;
                } else {
                    float cube = 0.044715f * val; cube = cube * val; cube = cube * val;
                    const float inner = val + cube;
                    const double th = tanh(0.7978845608028654 * (double)inner);
                    const float g = 0.5f * (1.0f + (float)th);
                    val = val * g;
                }
                val = val * up;
                a.out[row >> 1] = val;
            }
        } else if constexpr (EPI == EPI_CLS) {
            if (valid && r == wlane) {
                float vv = acc;
                if (row + a.row_offset < a.softcap_rows) {    // transformer.rs:375-381 (first `dim` logits only)
                    vv = vv / 30.0f;
                    vv = (float)tanh((double)vv);
                    vv = vv * 30.0f;
                }
                a.out[row] = vv;
                if (vv > best) { best = vv; best_i = row + a.row_offset; }   // rows ascend per lane: first maximum kept
            }
        }
    }

    LMRS_STAMP(3);
    if constexpr (EPI == EPI_CLS) {
        // workgroup argmax partial: larger value wins, ties -> lower index (== first index of the max)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(best_i, off);
            if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
        }
        __syncthreads();                                         // LDS reuse
        float* rv = reinterpret_cast<float*>(smem); int* ri = reinterpret_cast<int*>(smem + 16);
        if (lane == 0) { rv[wave] = best; ri[wave] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w2 = 1; w2 < kBlock / 64; ++w2)
                if (rv[w2] > best || (rv[w2] == best && ri[w2] < best_i)) { best = rv[w2]; best_i = ri[w2]; }
            best_i = cls_flag_nan_at_zero(a, best_i, bid);
            if (!a.has_tail) { a.part_val[pofs + bid] = best; a.part_idx[pofs + bid] = best_i; }
            else cls_publish(a, bid, ctag, best, best_i);
        }
        if (a.has_tail && bid == 0) { __syncthreads(); cls_consumer(a, nblk, tpre, ctag); }     // workgroup-uniform
    }
}

constexpr unsigned kTagSpinMax = 1u << 20;           // bound of every in-launch poll (err is set, the launch finishes with garbage instead of hanging)

// The value of lane i + D (D = 8 / 16: the up row of a gate row of L = D lanes) without the LDS crossbar round trip of __shfl_down:
// row_shl:8 inside a 16-lane row; across rows gfx950's v_permlane16_swap (with both operands = v its second result is [r1 r1 r3 r3]).
template <int D> __device__ __forceinline__ float lane_plus(float v) {
    if constexpr (D == 8) return dpp_f<0x108>(v);
    else if constexpr (D == 16) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(sw[1]);
    } else return __shfl_down(v, D);
}

// ------------------------------------------------------------------------------------------------
// Static-shape fused dequant-GEMV (Q8_0): N, L compile-time (lmrs_stage.h).  One tile (U = groups per
// cluster) covers a lane's whole share of a row; pass p handles rows p*RB .. p*RB+RB-1.
// Issue order: activation (+ norm weight) loads, then the first tile; the prologue runs under the tile's
// HBM latency with counted vmcnt waits; further passes are double-buffered (tile p+1 in flight while p is
// consumed).
// ------------------------------------------------------------------------------------------------
#define LMRS_STAMP0(k) do { if (a.dbg && threadIdx.x == 0 && bid == 0) a.dbg[k] = wall_clock64(); } while (0)
// Q4: packed-nibble weights and the reference's Q4_0 activation quantiser (lmrs_stage.h).
// PRO_ADD_RMS_QUANT (Gemma-2, transformer.rs:563-572 / 643-650 + the next norm): x' = x + rmsnorm(delta, add_w) is formed by
// every workgroup from the previous GEMV's output vector `delta` - that folds the reference's separate
// "x += rmsnorm(branch output)" step into the prologue of the kernel that consumes x' (one launch less per branch, a
// second serial norm chain more) - and workgroup 0 stores x' to `xout`, a DIFFERENT buffer from `xin` (the other
// workgroups are still reading it); the caller ping-pongs the two residual buffers.
// (device body: `bid` of `nblk` workgroups - the stand-alone launch passes bid / nblk, the merged qkv + attention
// launch the index among its GEMV workgroups)
template <int N, int L, int PRO, int EPI, int NTH, bool Q4 = false>
__device__ __forceinline__ void gemv_static_body(const GemvArgs& a, char* smem, const int bid, const int nblk) {
    using R = RowGeom<N, L, NTH, Q4>;
    using V = VecGeom<N, NTH>;
    LMRS_STAMP0(0);
    int8_t* xq = reinterpret_cast<int8_t*>(smem);
    float* xs = reinterpret_cast<float*>(smem + N);
    float* scratch = xs + ((V::G + 3) & ~3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane % L;
    const int o = a.o, n_pass = (o + R::RB - 1) / R::RB;
    const int8_t* wq = reinterpret_cast<const int8_t*>(a.wq);
    constexpr bool HAS_RMS = PRO == PRO_RMS_QUANT || PRO == PRO_ADD_RMS_QUANT;
    constexpr bool PREQ = PRO == PRO_PREQ;                                             // the activation arrives quantised

    unsigned ctag = 0; TailPre tpre{}; int pofs = 0;
    if constexpr (EPI == EPI_CLS) {
        if (a.has_tail) { ctag = cls_tag(a); if (bid == 0) tpre = tail_preload(a.tail.m); }     // (workgroup 0 will finish the step: ClsTail)
        else if (a.part_par) pofs = (int)((*a.part_par + 1u) & 1u) * a.part_par_floats;          // double-buffered partials (ArgmaxArgs)
    }
    int pos_pre = 0; unsigned tag_pre = 0;
    if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_TAG) pos_pre = a.st->pos;      // the kernel's first load: nothing it has to wait behind
    if constexpr (EPI == EPI_QKV_TAG) tag_pre = *a.seq + 1u;
    float4 v[V::NP], nw[V::NP], dl[V::NP], aw[V::NP];
    if constexpr (!PREQ) {
        if constexpr (PRO == PRO_ADD_RMS_QUANT) { vec_load<N, false, NTH>(dl, a.delta); vec_load<N, false, NTH>(aw, a.add_w); }
        vec_load<N, false, NTH>(v, a.xin);
        if constexpr (HAS_RMS) vec_load<N, false, NTH>(nw, a.rms_w);
    }
    auto row_of = [&](int pass) __attribute__((always_inline)) { const int rw = pass * R::RB + wave * R::RW + lane / L; return rw < o ? rw : o - 1; };
    uint64_t etab = 0;
    if constexpr (EPI == EPI_SWIGLU) etab = exp2f_tab_lane();
    asm volatile("" ::: "memory");               // keep the activation loads ahead of the weight tile in issue order
    // A CU returns vector-memory data in request order across its waves: without this barrier the activation loads of
    // the workgroup's later waves (L2 hits) queue behind the earlier waves' weight tiles (HBM misses).
    if (!PREQ && a.order_barrier) __builtin_amdgcn_s_barrier();
    // (waiting for the activation before issuing the tile, or issuing only part of it first, was measured: no gain)
    WTile<R::U> ta, tb;
    int pass = bid;                      // grid <= n_pass
    tile_issue<N, L, Q4>(ta, wq, a.ws, row_of(pass));
    // (requesting the second pass's tile here as well - it would stream under the prologue - was measured slower on every model, Gemma's
    // 5 us folded prologue included: the more bytes are queued ahead of a workgroup's activation loads, the later they land)
    // Epilogue operands that live in memory are fetched here, under the prologue, instead of as a dependent round trip at
    // the very end of the kernel: the residual value of the first pass's row (each row has exactly one writer, nobody else
    // touches it during the launch) and the position the QKV epilogue stores the V row at.
    // (unconditional, every lane of the row: a load under a lane predicate would cost the kernel its counted vmcnt waits)
    float resid0 = 0.0f;
    if constexpr (EPI == EPI_RESID) resid0 = a.out[row_of(pass)];
    // (two-pass launches: issuing the second tile here as well was measured - slower on every model: the more bytes the
    // chip has in flight, the later every workgroup's activation lands)
    __builtin_amdgcn_sched_barrier(0);           // the prologue's first wait must not be scheduled above the tile's loads

    if constexpr (PRO == PRO_PREQ) {
        constexpr int XB = Q4 ? N / 2 : N;
        for (int e = threadIdx.x * 16; e < XB; e += NTH * 16)
            *reinterpret_cast<int4*>(xq + e) = *reinterpret_cast<const int4*>(preq_bytes(a, e));
        for (int g = threadIdx.x; g < V::G; g += NTH) xs[g] = preq_scale(a, g);
    } else {
        unsigned long long* dbg = bid == 0 ? a.dbg : nullptr;
        if constexpr (PRO == PRO_ADD_RMS_QUANT) {
            // (Round 4, measured and removed: requesting the second pass's tile from inside this ~5 us prologue, once the branch output has
            // landed.  Gemma-2-2B Q4_0: 889 -> 894 us per step with three passes per workgroup, 898 -> 952 with two - the tiles of 288
            // workgroups queued ahead of each other's norm-weight and residual loads.)
            vec_rmsnorm<N, NTH>(dl, aw, a.eps, a.add_unit, scratch);                 // rmsnorm(branch output)
#pragma unroll
            for (int i = 0; i < V::NP; ++i) {
                v[i].x = v[i].x + dl[i].x; v[i].y = v[i].y + dl[i].y; v[i].z = v[i].z + dl[i].z; v[i].w = v[i].w + dl[i].w;   // x[i] += emb[i]
                const int e = V::elem(i, (int)threadIdx.x);
                if (bid == 0 && (V::FULL || i < V::NP - 1 || e < N)) *reinterpret_cast<float4*>(a.xout + e) = v[i];
            }
        }
        if constexpr (HAS_RMS) vec_rmsnorm<N, NTH>(v, nw, a.eps, a.add_unit, scratch, dbg, NoHook(), a.chain_spread ? (((int)blockIdx.x >> 8) * 2) & (NTH / 64 - 1) : 0);
        if constexpr (Q4) vec_quantize_q4<N, NTH>(v, xq, xs, dbg);
        else vec_quantize_q8<N, NTH>(v, xq, xs, dbg);
        if (a.dbg && bid == 0 && threadIdx.x == 0) a.dbg[7] = wall_clock64();
    }
    lds_barrier();
    LMRS_STAMP0(1);

    float best = __uint_as_float(0xff800000u); int best_i = 0x7fffffff;   // EPI_CLS
    const bool writer = r == R::WR;                                        // the lane of the row that holds the finished sum

    auto finish = [&](float acc, int ps) __attribute__((always_inline)) {
        const int row = ps * R::RB + wave * R::RW + lane / L;
        const bool valid = row < o;
        if constexpr (EPI == EPI_STORE) {
            if (valid && writer) a.out[row] = acc;
        } else if constexpr (EPI == EPI_RESID) {
            if (valid && writer) a.out[row] = (ps == bid ? resid0 : a.out[row]) + acc;
        } else if constexpr (EPI == EPI_QKV) {
            if (valid && writer) {
                if (row < a.att_dim) a.out[row] = acc;
                else if (row < a.att_dim + a.kv_dim) a.k_raw[row - a.att_dim] = acc;
                else a.v_cache[((size_t)a.layer * a.seq_len + pos_pre) * a.kv_dim + (row - a.att_dim - a.kv_dim)] = acc;
            }
        } else if constexpr (EPI == EPI_QKV_TAG) {
            // the attention workgroups of the SAME launch are polling for these: one aligned 8-byte write-through store per value,
            // the tag in the upper half (the data is the flag: no fence, no counter)
            if (valid && writer) {
                __hip_atomic_store(a.gran + row, ((unsigned long long)tag_pre << 32) | __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (row >= a.att_dim + a.kv_dim) a.v_cache[((size_t)a.layer * a.seq_len + pos_pre) * a.kv_dim + (row - a.att_dim - a.kv_dim)] = acc;   // for the later steps
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            const float up = __shfl_down(acc, L);                       // rows interleaved: 2i gate, 2i+1 up
            const float hval = swiglu_t(acc, up, etab);                 // all lanes (expf shuffles its table)
            if (valid && writer && ((lane / L) & 1) == 0) a.out[row >> 1] = hval;
        } else if constexpr (EPI == EPI_GELU) {
            const float up = __shfl_down(acc, L);
            if (valid && writer && ((lane / L) & 1) == 0) a.out[row >> 1] = geglu(acc, up);
        } else if constexpr (EPI == EPI_CLS) {
            if (valid && writer) {
                float vv = acc;
                if (row + a.row_offset < a.softcap_rows) {              // Gemma, transformer.rs:375-381 (first `dim` logits only)
                    vv = vv / 30.0f;
                    vv = (float)tanh((double)vv);
                    vv = vv * 30.0f;
                }
                a.out[row] = vv;
                if (vv > best) { best = vv; best_i = row + a.row_offset; }
            }
        }
    };

    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_GELU) {
        // Gate/up launches: the activation (SiLU: glibc expf in double; GELU: an f64 tanh, ~230 instructions) costs the WAVE its
        // whole instruction stream however few lanes need it, and only one lane in 2L holds a (gate, up) pair.  So the passes are
        // taken up to three at a time: the second pass's pair moves one lane up, the third's two (DPP row_shr, same 16-lane row) and
        // ONE evaluation serves all of them (three since round 4: Gemma-2-2B's three passes per workgroup paid two evaluations).
        for (;;) {
            const int p1 = pass + nblk;
            const bool have1 = p1 < n_pass;
            if (have1) tile_issue<N, L, Q4>(tb, wq, a.ws, row_of(p1));
            const float acc_a = tile_consume<N, L, Q4>(ta, xq, xs);
            if (pass == bid) LMRS_STAMP0(2);
            const int p2 = p1 + nblk;
            const bool have2 = have1 && p2 < n_pass;
            float acc_b = 0.0f;
            if (have1) {                                                   // wave-uniform
                if (have2) tile_issue<N, L, Q4>(ta, wq, a.ws, row_of(p2));
                acc_b = tile_consume<N, L, Q4>(tb, xq, xs);
            }
            const int p3 = p2 + nblk;
            const bool have3 = have2 && p3 < n_pass;
            float acc_c = 0.0f;
            if (have2) {                                                   // wave-uniform
                if (have3) tile_issue<N, L, Q4>(tb, wq, a.ws, row_of(p3));
                acc_c = tile_consume<N, L, Q4>(ta, xq, xs);
            }
            const bool third = have2;
            const float up_a = lane_plus<L>(acc_a), up_b = lane_plus<L>(acc_b), up_c = lane_plus<L>(acc_c);         // rows interleaved: 2i gate, 2i+1 up
            const float gate_b1 = dpp_f<0x111>(acc_b), up_b1 = dpp_f<0x111>(up_b);           // pass B's pair, one lane up
            const float gate_c2 = dpp_f<0x112>(acc_c), up_c2 = dpp_f<0x112>(up_c);           // pass C's pair, two lanes up
            const bool lane_a = writer && ((lane / L) & 1) == 0;                            // holds pass A's pair
            const bool lane_b = have1 && (r == R::WR + 1) && ((lane / L) & 1) == 0;         // its neighbour: pass B's pair
            const bool lane_c = third && (r == R::WR + 2) && ((lane / L) & 1) == 0;         // the next one: pass C's pair
            const float gate = lane_c ? gate_c2 : lane_b ? gate_b1 : acc_a, up = lane_c ? up_c2 : lane_b ? up_b1 : up_a;
            float hval;
            if constexpr (EPI == EPI_SWIGLU) hval = swiglu_t(gate, up, etab);               // all lanes (expf shuffles its table)
            else hval = (lane_a || lane_b || lane_c) ? geglu(gate, up) : 0.0f;
            const int row_a = pass * R::RB + wave * R::RW + lane / L, row_b = p1 * R::RB + wave * R::RW + lane / L, row_c = p2 * R::RB + wave * R::RW + lane / L;
            if (lane_a && row_a < o) a.out[row_a >> 1] = hval;
            if (lane_b && row_b < o) a.out[row_b >> 1] = hval;
            if (lane_c && row_c < o) a.out[row_c >> 1] = hval;
            // (a fourth pass would find its tile in `tb` while the next round consumes `ta` first: gemv_grid / launch_gemv never give a gate/up
            // workgroup more than three passes, and the loop ends here)
            break;
        }
    } else
    // double-buffered passes
    for (;;) {
        const int p1 = pass + nblk;
        if (p1 < n_pass) tile_issue<N, L, Q4>(tb, wq, a.ws, row_of(p1));
        finish(tile_consume<N, L, Q4>(ta, xq, xs), pass);
        if (pass == bid) LMRS_STAMP0(2);
        if (p1 >= n_pass) break;
        const int p2 = p1 + nblk;
        if (p2 < n_pass) tile_issue<N, L, Q4>(ta, wq, a.ws, row_of(p2));
        finish(tile_consume<N, L, Q4>(tb, xq, xs), p1);
        if (p2 >= n_pass) break;
        pass = p2;
    }
    LMRS_STAMP0(3);
    if constexpr (EPI == EPI_CLS) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(best_i, off);
            if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
        }
        __syncthreads();
        float* rv = reinterpret_cast<float*>(smem); int* ri = reinterpret_cast<int*>(smem + 64);
        if (lane == 0) { rv[wave] = best; ri[wave] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w2 = 1; w2 < NTH / 64; ++w2)
                if (rv[w2] > best || (rv[w2] == best && ri[w2] < best_i)) { best = rv[w2]; best_i = ri[w2]; }
            best_i = cls_flag_nan_at_zero(a, best_i, bid);
            if (!a.has_tail) { a.part_val[pofs + bid] = best; a.part_idx[pofs + bid] = best_i; }
            else cls_publish(a, bid, ctag, best, best_i);
        }
        if constexpr (NTH == kBlock) {
            if (a.has_tail && bid == 0) { __syncthreads(); cls_consumer(a, nblk, tpre, ctag); }     // workgroup-uniform
        }
    }
}

// Kernel-argument preload: a kernel's first instruction used to be an s_load of the argument block, and the activation loads could
// only be issued when it came back - two dependent memory round trips between the launch and the first useful byte.  gfx950 can
// deliver the first kernel-argument dwords in SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count, see the Makefile), but only
// for SCALAR leading arguments, not for a struct passed by value: the pointers the first loads need are therefore repeated in front
// of the argument struct (HotArgs, at most 7 pointers = 14 user SGPRs) and folded back into the struct's SSA copy here.
struct HotPtrs { const float* xin; const void* wq; const float* ws; const float* rms_w; const DevState* st; const unsigned* seq; float* out; };
__device__ __forceinline__ GemvArgs with_hot(const GemvArgs& a0, const float* xin, const void* wq, const float* ws, const float* rms_w, const DevState* st, const unsigned* seq, float* out) {
    GemvArgs a = a0;
    a.xin = xin; a.wq = wq; a.ws = ws; a.rms_w = rms_w; a.st = st; a.seq = seq; a.out = out;
    return a;
}
#define LMRS_HOT_PARAMS const float* h_xin, const void* h_wq, const float* h_ws, const float* h_rms_w, const DevState* h_st, const unsigned* h_seq, float* h_out
#define LMRS_HOT_OF(g) (g).xin, (g).wq, (g).ws, (g).rms_w, (g).st, (g).seq, (g).out

// The 512-thread classes (w2: a long quantise prologue over 8192 / 9216 values) ask for two workgroups per CU - four waves per SIMD, at most 128
// VGPRs: 148 under the max-ilp scheduling strategy left the 384 workgroups of the 3072-wide models one per CU, in two rounds.  Same-box A/B
// (profiles/r5_ab_w2_launch_bounds.txt): Llama-3.2-3B 975 -> 985 tok/s, Phi-3.5 860 -> 870, Llama-3.2-1B unchanged; a grid capped at 256 / 192
// workgroups with a second pass instead: 981 / 978.
// Not for Gemma-2-2B's 9216-wide class: under the bound it spills (20 / 24 bytes of scratch per lane) and its w2 launch went 5.3 -> 7.5 us
// (profiles/r5_bench_gemma2b_q4.json of the first r5 artefact run against r4b: 1139 -> 1088 tok/s); its 36-72 workgroups never share a CU anyway.
#define LMRS_STATIC_BOUNDS(N_, NTH_) __launch_bounds__(NTH_, ((NTH_) == 512 && (N_) <= 8192) ? 4 : 1)
template <int N, int L, int PRO, int EPI, int NTH, bool Q4 = false>
__global__ LMRS_STATIC_BOUNDS(N, NTH) void gemv_static_kernel(LMRS_HOT_PARAMS, const GemvArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemvArgs a = with_hot(a0, h_xin, h_wq, h_ws, h_rms_w, h_st, h_seq, h_out);
    gemv_static_body<N, L, PRO, EPI, NTH, Q4>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Static classes (N, L, PRO, EPI, threads, Q4): the instantiated shapes of the supported model families.
#define LMRS_STATIC_TABLE(X)                                                                                            \
    /* Q8_0, dim 2048 / hidden 8192: Llama-3.2-1B */                                                                    \
    X(2048, 32, PRO_RMS_QUANT, EPI_QKV, 256, false) X(2048, 32, PRO_QUANT, EPI_RESID, 256, false) X(2048, 16, PRO_RMS_QUANT, EPI_SWIGLU, 256, false) \
    X(2048, 8, PRO_RMS_QUANT, EPI_CLS, 256, false) X(2048, 32, PRO_PREQ, EPI_STORE, 256, false) X(2048, 8, PRO_PREQ, EPI_STORE, 256, false) \
    X(8192, 64, PRO_QUANT, EPI_RESID, 512, false) X(8192, 32, PRO_PREQ, EPI_STORE, 256, false) \
    /* row-sharded step: the activation arrives quantised from the shards that produced it */                         \
    X(2048, 32, PRO_PREQ, EPI_RESID, 256, false) X(8192, 32, PRO_PREQ, EPI_RESID, 256, false) X(3072, 32, PRO_PREQ, EPI_RESID, 256, false) \
    /* Q8_0, dim 3072: Llama-3.2-3B, Phi-3.5 */                                                                         \
    X(3072, 32, PRO_RMS_QUANT, EPI_QKV, 256, false) X(3072, 16, PRO_RMS_QUANT, EPI_QKV, 256, false) X(3072, 32, PRO_QUANT, EPI_RESID, 256, false) \
    X(3072, 16, PRO_RMS_QUANT, EPI_SWIGLU, 256, false) X(3072, 16, PRO_RMS_QUANT, EPI_CLS, 256, false)                  \
    X(3072, 32, PRO_PREQ, EPI_STORE, 256, false) X(3072, 16, PRO_PREQ, EPI_STORE, 256, false)                           \
    /* Q8_0, Gemma-2-2B: dim 2304, att 2048, hidden 9216 */                                                             \
    X(2304, 16, PRO_RMS_QUANT, EPI_QKV, 256, false) X(2304, 16, PRO_ADD_RMS_QUANT, EPI_QKV, 256, false) X(2048, 32, PRO_QUANT, EPI_STORE, 256, false) \
    X(2304, 16, PRO_ADD_RMS_QUANT, EPI_GELU, 256, false) X(9216, 64, PRO_QUANT, EPI_STORE, 512, false) X(2304, 8, PRO_ADD_RMS_QUANT, EPI_CLS, 256, false) \
    /* Q4_0, Gemma-2-2B */                                                                                              \
    X(2304, 8, PRO_RMS_QUANT, EPI_QKV, 256, true) X(2304, 8, PRO_ADD_RMS_QUANT, EPI_QKV, 256, true) X(2048, 32, PRO_QUANT, EPI_STORE, 256, true) \
    X(2304, 8, PRO_ADD_RMS_QUANT, EPI_GELU, 256, true) X(9216, 32, PRO_QUANT, EPI_STORE, 512, true) X(2304, 8, PRO_ADD_RMS_QUANT, EPI_CLS, 256, true) \
    /* Q4_0, Llama-3.2-1B */                                                                                            \
    X(2048, 16, PRO_RMS_QUANT, EPI_QKV, 256, true) X(2048, 32, PRO_QUANT, EPI_RESID, 256, true) X(2048, 8, PRO_RMS_QUANT, EPI_SWIGLU, 256, true) \
    X(8192, 32, PRO_QUANT, EPI_RESID, 512, true) X(2048, 8, PRO_RMS_QUANT, EPI_CLS, 256, true)

static int env_flag(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// The class (L, threads) for a launch, or L = 0 when the shape has no static kernel (the generic kernel takes it).
//   * L: enough workgroups to cover the chip with rows split over as few clusters as possible; w1w3 at 16 lanes per row and
//     two passes per workgroup (measured best of L = 8/16/32 x grid 256/512/1024 at dim 2048).
//   * threads: the w2 projection has few rows and a long quantise prologue (8192 / 9216 elements of dependent VALU chains):
//     512 threads keep the workgroup count and give the prologue twice the lanes (measured 5.5 -> 4.4 us per launch).
struct StaticClass { int L, nt; };
static StaticClass static_class(const GemvArgs& a, int pro, int epi) {
    const bool q4 = a.q4 != 0, glu = epi == EPI_SWIGLU || epi == EPI_GELU;
    int L = 0, nt = 256;
    if (!q4) {
        if (a.n == 2048) L = glu ? 16 : (a.o >= 8192 ? 8 : 32);
#ifndef LMRS_PHI_QKV16
        else if (a.n == 3072) L = (a.o >= 8192 && epi != EPI_QKV) ? 16 : 32;     // (qkv: the merged launch runs in one round of resident workgroups - 24 KB tiles balance its passes better than 48 KB ones)
#else
        else if (a.n == 3072) L = a.o >= 8192 ? 16 : 32;
#endif
        else if (a.n == 2304) L = epi == EPI_CLS ? 8 : 16;
        // (round 4, with the grouped quantiser: 32 lanes per row and 256 threads for w2 - half the waves to dispatch, twice the prologue per
        // lane - measured 423 us per step against 414: removed)
        else if (a.n == 8192) { if (pro == PRO_QUANT) { L = 64; nt = 512; } else L = 32; }
        else if (a.n == 9216) { L = 64; nt = 512; }
    } else {
        if (a.n == 2304) L = 8;
        else if (a.n == 2048) L = epi == EPI_QKV ? 16 : (a.o >= 8192 ? 8 : 32);
        else if (a.n == 9216 || a.n == 8192) { L = 32; nt = 512; }
    }
#define X(n_, l_, p_, e_, nt_, q_) if (a.n == n_ && L == l_ && pro == p_ && epi == e_ && nt == nt_ && q4 == q_) return {L, nt};
    LMRS_STATIC_TABLE(X)
#undef X
    return {0, 256};
}
bool gemv_is_static(const GemvArgs& a, int pro, int epi) { return static_class(a, pro, epi).L != 0; }

// Measurement: when set, the next GEMV launch carries these events on its own dispatch (hipExtLaunchKernelGGL), so
// that hipEventElapsedTime(start, stop) is that kernel's begin->end time, the same interval rocprofv3 reports.
static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;
void set_gemv_launch_events(hipEvent_t start, hipEvent_t stop) { t_ev_start = start; t_ev_stop = stop; }
// ... or a pool of (start, stop) pairs handed out to EVERY launch of the decode step in launch order (GEMVs, attention, argmax):
// an eager replay of the real step then yields the duration of each of its kernels (lmrs_bench_step).
static thread_local hipEvent_t* t_ev_pool = nullptr; static thread_local int t_ev_cap = 0, t_ev_used = 0;
static thread_local int* t_ev_tags = nullptr; static thread_local int t_ev_tag = 0;      // the caller's label for the launches that follow
void set_launch_event_pool(hipEvent_t* pairs, int n_pairs, int* tags) { t_ev_pool = pairs; t_ev_cap = n_pairs; t_ev_used = 0; t_ev_tags = tags; t_ev_tag = 0; }
void set_launch_tag(int tag) { t_ev_tag = tag; }
int launch_event_pool_used() { return t_ev_used; }
bool next_launch_events(hipEvent_t* a, hipEvent_t* b) {
    if (t_ev_pool && t_ev_used < t_ev_cap) {
        *a = t_ev_pool[2 * t_ev_used]; *b = t_ev_pool[2 * t_ev_used + 1];
        if (t_ev_tags) t_ev_tags[t_ev_used] = t_ev_tag;
        ++t_ev_used; return true;
    }
    if (t_ev_pool) { ++t_ev_used; return false; }                                           // pool exhausted: keep counting (the caller sizes it from a dry pass)
    if (t_ev_start) { *a = t_ev_start; *b = t_ev_stop; return true; }
    return false;
}
#define LMRS_LAUNCH_GRID(kern, grid3, nt, smem, s, ...)                                                     \
    do {                                                                                                    \
        hipEvent_t ea_, eb_;                                                                                \
        if (next_launch_events(&ea_, &eb_)) hipExtLaunchKernelGGL(kern, grid3, dim3(nt), smem, s, ea_, eb_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid3, dim3(nt), smem, s, __VA_ARGS__);                               \
    } while (0)
#define LMRS_LAUNCH_NT(kern, grid, nt, smem, s, a) LMRS_LAUNCH_GRID(kern, dim3(grid), nt, smem, s, a)
#define LMRS_LAUNCH(kern, grid, smem, s, a) LMRS_LAUNCH_NT(kern, grid, kBlock, smem, s, a)

// Kernels that ask for more than 64 KB of dynamic LDS need the attribute once per (function, device).
static void allow_big_lds(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.insert({fn, dev}).second) {
        // the limit covers static + dynamic LDS: a kernel with a few static bytes (the __syncthreads_and / _or helpers keep 256) is refused
        // the full 160 KB, and the refusal would surface as `invalid argument` at its first launch above 64 KB
        hipFuncAttributes fa{};
        const size_t fixed = hipFuncGetAttributes(&fa, fn) == hipSuccess ? fa.sharedSizeBytes : 0;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - fixed));
    }
}

static size_t gemv_smem(const GemvArgs& a, int pro) {
    const int n = a.n, G = n / kGS;
    size_t s = ((n + 15) & ~15) + (size_t)((G + 3) & ~3) * 4;
    if (pro == PRO_RMS_QUANT || pro == PRO_ADD_RMS_QUANT) s += (size_t)rms_scratch_floats(n) * 4;
    return s < 64 ? 64 : s;
}

// ---- shape -> (L, U, NP) -----------------------------------------------------------------------
// L: as few lanes per row as still give >= 256 workgroups (one per CU), because a row split over
//    several clusters pays NC-1 cross-lane hops in its accumulation chain; valid only if the row's
//    groups divide evenly over the clusters and (NC > 1) a cluster's groups fit one U-step batch.
// U: 16-byte steps in flight per lane.  NP: float4 per lane of the activation vector (ceil(n/1024)).
struct GemvShape { int L, U, NP; };

static GemvShape pick_shape(const GemvArgs& a, int epi) {
    const int cl = a.q4 ? 4 : 8, G = a.n / kGS;
    const int NP = a.n <= 2048 ? 2 : (a.n <= 4096 ? 4 : (a.n <= 10240 ? 10 : 16));
    const int maxL = (epi == EPI_SWIGLU || epi == EPI_GELU) ? 32 : 64;
    int best = cl;
    for (int L = cl; L <= maxL; L *= 2) {
        const int NC = L / cl;
        if (G % NC) continue;
        const int Gc = G / NC;
        if (NC > 1 && Gc > 16) continue;
        best = L;
        const int RB = (64 / L) * (kBlock / 64);
        if ((a.o + RB - 1) / RB >= 256) break;
    }
    const int NC = best / cl, Gc = G / NC;
    const int U = NC == 1 ? 16 : (Gc <= 4 ? 4 : (Gc <= 8 ? 8 : 16));
    return {best, U, NP};
}

// Instantiated kernels: the classes the supported model shapes map to, plus a generic class per
// (prologue, epilogue) that handles any n (one cluster per row, NP = 10).
#define LMRS_GEMV_TABLE(X)                                                                                 \
    /* generic */                                                                                          \
    X(8, 16, 10, PRO_PREQ, EPI_STORE, false) X(8, 16, 10, PRO_QUANT, EPI_STORE, false) X(8, 16, 10, PRO_QUANT, EPI_RESID, false) X(8, 16, 10, PRO_PREQ, EPI_RESID, false) \
    X(8, 16, 10, PRO_RMS_QUANT, EPI_STORE, false) X(8, 16, 10, PRO_RMS_QUANT, EPI_QKV, false)              \
    X(8, 16, 10, PRO_RMS_QUANT, EPI_SWIGLU, false) X(8, 16, 10, PRO_RMS_QUANT, EPI_GELU, false) X(8, 16, 10, PRO_RMS_QUANT, EPI_CLS, false) \
    /* generic, 10240 < n <= 16384 (hidden_dim 14336 of Gemma-2-9B / Llama-3.1-8B): the inputs that are only quantised */      \
    X(8, 16, 16, PRO_PREQ, EPI_STORE, false) X(8, 16, 16, PRO_QUANT, EPI_STORE, false) X(8, 16, 16, PRO_QUANT, EPI_RESID, false) X(8, 16, 16, PRO_PREQ, EPI_RESID, false) \
    X(4, 16, 16, PRO_PREQ, EPI_STORE, true) X(4, 16, 16, PRO_QUANT, EPI_STORE, true) X(4, 16, 16, PRO_QUANT, EPI_RESID, true) \
    /* n = 2048 (Llama-3.2-1B dim; Gemma-2-2B att_dim) */                                                  \
    X(32, 4, 2, PRO_RMS_QUANT, EPI_QKV, false) X(32, 4, 2, PRO_QUANT, EPI_RESID, false) X(32, 4, 2, PRO_QUANT, EPI_STORE, false) \
    X(8, 16, 2, PRO_RMS_QUANT, EPI_SWIGLU, false) X(8, 16, 2, PRO_RMS_QUANT, EPI_CLS, false) X(8, 16, 2, PRO_RMS_QUANT, EPI_QKV, false) \
    X(32, 4, 2, PRO_PREQ, EPI_STORE, false) X(8, 16, 2, PRO_PREQ, EPI_STORE, false) X(16, 8, 2, PRO_PREQ, EPI_STORE, false) \
    /* n = 2304 / 3072 (Gemma-2-2B, Llama-3.2-3B, Phi-3.5 dim) */                                          \
    X(16, 16, 4, PRO_RMS_QUANT, EPI_QKV, false) X(32, 8, 4, PRO_QUANT, EPI_RESID, false) X(16, 16, 4, PRO_QUANT, EPI_RESID, false) \
    X(8, 16, 4, PRO_RMS_QUANT, EPI_SWIGLU, false) X(8, 16, 4, PRO_RMS_QUANT, EPI_GELU, false) X(8, 16, 4, PRO_RMS_QUANT, EPI_CLS, false) \
    X(16, 16, 4, PRO_PREQ, EPI_STORE, false) X(32, 8, 4, PRO_PREQ, EPI_STORE, false)                       \
    /* n = 8192 / 9216 (hidden_dim) */                                                                     \
    X(32, 16, 10, PRO_QUANT, EPI_RESID, false) X(32, 16, 10, PRO_QUANT, EPI_STORE, false) X(64, 16, 10, PRO_QUANT, EPI_STORE, false) \
    X(64, 16, 10, PRO_QUANT, EPI_RESID, false) X(32, 16, 10, PRO_PREQ, EPI_STORE, false) X(64, 16, 10, PRO_PREQ, EPI_STORE, false) \
    /* Q4_0: generic classes per cluster layout */                                                         \
    X(4, 16, 10, PRO_PREQ, EPI_STORE, true) X(4, 16, 10, PRO_QUANT, EPI_STORE, true) X(4, 16, 10, PRO_QUANT, EPI_RESID, true) \
    X(4, 16, 10, PRO_RMS_QUANT, EPI_STORE, true) X(4, 16, 10, PRO_RMS_QUANT, EPI_QKV, true) X(4, 16, 10, PRO_RMS_QUANT, EPI_SWIGLU, true) \
    X(4, 16, 10, PRO_RMS_QUANT, EPI_GELU, true) X(4, 16, 10, PRO_RMS_QUANT, EPI_CLS, true)                 \
    X(16, 4, 10, PRO_PREQ, EPI_STORE, true) X(16, 4, 10, PRO_QUANT, EPI_RESID, true) X(16, 4, 10, PRO_QUANT, EPI_STORE, true) \
    X(16, 4, 10, PRO_RMS_QUANT, EPI_QKV, true) X(16, 16, 10, PRO_QUANT, EPI_RESID, true) X(16, 16, 10, PRO_QUANT, EPI_STORE, true) \
    X(16, 16, 10, PRO_PREQ, EPI_STORE, true) X(8, 16, 10, PRO_PREQ, EPI_STORE, true) X(8, 16, 10, PRO_RMS_QUANT, EPI_QKV, true)

static bool gemv_instantiated(int L, int U, int NP, int pro, int epi, bool q4) {
#define X(l, u, np, p, e, q) if (L == l && U == u && NP == np && pro == p && epi == e && q4 == q) return true;
    LMRS_GEMV_TABLE(X)
#undef X
    return false;
}

static GemvShape resolve_shape(const GemvArgs& a, int pro, int epi) {
    GemvShape sh = pick_shape(a, epi);
    if (!gemv_instantiated(sh.L, sh.U, sh.NP, pro, epi, a.q4 != 0)) {
        const int big = a.n <= 10240 ? 10 : 16;
        if (gemv_instantiated(sh.L, sh.U, big, pro, epi, a.q4 != 0)) sh.NP = big;
        else sh = {a.q4 ? 4 : 8, 16, big};
    }
    return sh;
}

int gemv_grid(const GemvArgs& a, int pro, int epi) {
    const StaticClass sc = static_class(a, pro, epi);
    const GemvShape sh = sc.L ? GemvShape{sc.L, 0, 0} : resolve_shape(a, pro, epi);
    const int RB = (64 / sh.L) * ((sc.L ? sc.nt : kBlock) / 64);
    const int n_pass = (a.o + RB - 1) / RB;
    int cap = 4096;
    if (epi == EPI_CLS) cap = 512;                         // classifier: persistent-style grid, prologue paid once per workgroup
    else if (sc.L && (epi == EPI_SWIGLU || epi == EPI_GELU)) {                             // w1w3: two passes per workgroup ...
        int k = 2;
        // ... unless that leaves a grid between one and two workgroups per CU (Gemma-2-2B: 288 on 256 CUs - 32 CUs then carry twice the
        // stream and two prologues, and the launch ends with them): three passes per workgroup when that fits one per CU
        if (n_pass / 2 > 256 && n_pass / 2 < 448 && (n_pass + 2) / 3 <= 256) k = 3;
        cap = (n_pass + k - 1) / k;                        // (never more than three: the gate/up loop of gemv_static_body takes up to three passes)
    }
    return n_pass < cap ? n_pass : cap;
}

hipError_t launch_gemv(const GemvArgs& a0, int pro, int epi, hipStream_t s, int grid_hint) {
    static const int order_barrier = env_flag("LMRS_ORDER_BARRIER", 1);
    static const int chain_spread = env_flag("LMRS_CHAIN_SPREAD", 1);
    GemvArgs a = a0;
    a.order_barrier = order_barrier; a.chain_spread = chain_spread;
    if (a.n % kGS != 0 || a.n > kMaxP * 1024 || a.o <= 0) return hipErrorInvalidValue;
    int grid = grid_hint > 0 ? grid_hint : gemv_grid(a, pro, epi);
    if (grid_hint > 0 && (epi == EPI_SWIGLU || epi == EPI_GELU)) {      // a caller's grid must not give a gate/up workgroup more than the three passes its loop takes
        const int least = gemv_grid(a, pro, epi);
        if (grid < least) grid = least;
    }
    if (epi == EPI_CLS && a.has_tail && (!a.tail.cls_seq || !a.tail.err || !a.tail.part_pk || a.o + a.row_offset >= (1 << 20) - 1)) return hipErrorInvalidValue;
    const size_t smem = gemv_smem(a, pro);
    const StaticClass sc = static_class(a, pro, epi);
    if (sc.L) {
#define X(n_, l_, p_, e_, nt_, q_)                                                                         \
        if (a.n == n_ && sc.L == l_ && pro == p_ && epi == e_ && sc.nt == nt_ && (a.q4 != 0) == q_) {      \
            LMRS_LAUNCH_GRID((gemv_static_kernel<n_, l_, p_, e_, nt_, q_>), dim3(grid), nt_, smem, s, LMRS_HOT_OF(a), a);            \
            return hipGetLastError();                                                                      \
        }
        LMRS_STATIC_TABLE(X)
#undef X
    }
    if (pro == PRO_ADD_RMS_QUANT) return hipErrorInvalidValue;       // static kernels only (callers check gemv_is_static)
    const GemvShape sh = resolve_shape(a, pro, epi);
#define X(l, u, np, p, e, q)                                                                               \
    if (sh.L == l && sh.U == u && sh.NP == np && pro == p && epi == e && (a.q4 != 0) == q) {              \
        LMRS_LAUNCH((gemv_kernel<l, u, np, p, e, q>), grid, smem, s, a);                                   \
        return hipGetLastError();                                                                          \
    }
    LMRS_GEMV_TABLE(X)
#undef X
    return hipErrorInvalidValue;
}

constexpr int kAttF4 = 16;           // float4 per lane per chunk in the stand-alone kernel: CH * HS / 4 / 256 <= 16

// Chunk geometry shared by the loads and the LDS tile writes.  Slot i of a lane is float4 number f = tid + i * 256 of
// the chunk (row f / HS4, column f % HS4).  Slots are switched on and off in groups of 4 by a wave-uniform test (scalar
// branch, no exec masking, and the address / predicate arithmetic of the dead slots is skipped with them); rows past
// the chunk's last one are clamped to it (a harmless duplicate load) instead of being masked off lane by lane.
// nrows = rows written to the tile = ct rounded up to 16 (the V chain runs in batches of 16).
template <int HS> struct AttGeom {
    static constexpr int HS4 = HS / 4, RS = HS + 4;
    static constexpr bool ALIGNED = (kBlock % HS4 == 0) && ((16 * HS4) % kBlock == 0 || kBlock % (16 * HS4) == 0);   // a slot never straddles nrows
};
// rows t0 .. t0+ct of one kv head (row stride kv_dim floats) -> registers, 16 B per lane per load, coalesced per row
template <int HS, int NF>
__device__ __forceinline__ void att_gload(float4 (&rg)[NF], const float* __restrict__ base, int t0, int T, int CH, int kv_dim) {
    constexpr int HS4 = HS / 4;
    const int ct = (T - t0) < CH ? (T - t0) : CH, nrows = (ct + 15) & ~15, nf = nrows * HS4;
    const int r0 = (int)threadIdx.x / HS4, c4 = (int)threadIdx.x - r0 * HS4;
    const float* lane_base = base + (size_t)t0 * kv_dim + c4 * 4;
#pragma unroll
    for (int i0 = 0; i0 < NF; i0 += 4) {
        if (i0 * kBlock < nf) {                                              // wave-uniform
#pragma unroll
            for (int i = i0; i < i0 + 4 && i < NF; ++i) {
                int row;
                if constexpr (kBlock % HS4 == 0) row = r0 + i * (kBlock / HS4);
                else row = ((int)threadIdx.x + i * kBlock) / HS4;
                const int col4 = (kBlock % HS4 == 0) ? 0 : (((int)threadIdx.x + i * kBlock) - row * HS4 - c4) * 4;
                row = row < ct - 1 ? row : ct - 1;
                rg[i] = ld_f32x4<false>(lane_base + (size_t)row * kv_dim + col4);
            }
        }
    }
}
// registers -> LDS tile (row stride HS + 4 floats); optionally scaled per row (V phase: products a_t * v_t[d]; rows
// ct .. nrows-1 get the zero-padded weights, i.e. +-0.0 products, so that the serial chain can run in full batches of 16).
// Row `patch_t` (absolute; -1: none) is then overwritten with patch[] (* its weight) BY THE SAME THREAD that wrote it, so
// the two LDS writes are ordered without a barrier.
template <int HS, int NF, bool SCALE>
__device__ __forceinline__ void att_tstore(const float4 (&rg)[NF], float* tile, const float* rowscale, int t0, int T, int CH, int patch_t, const float* patch) {
    constexpr int HS4 = HS / 4, RS = HS + 4;
    const int ct = (T - t0) < CH ? (T - t0) : CH, nrows = (ct + 15) & ~15, nf = nrows * HS4;
    const int r0 = (int)threadIdx.x / HS4, c4 = (int)threadIdx.x - r0 * HS4;
#pragma unroll
    for (int i0 = 0; i0 < NF; i0 += 4) {
        if (i0 * kBlock < nf) {                                              // wave-uniform
            int row[4], col[4]; float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if constexpr (kBlock % HS4 == 0) { row[u] = r0 + i * (kBlock / HS4); col[u] = c4; }
                else { const int f = (int)threadIdx.x + i * kBlock; row[u] = f / HS4; col[u] = f - row[u] * HS4; }
                if constexpr (SCALE) sc[u] = rowscale[row[u] < nrows ? row[u] : nrows - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i < NF && (AttGeom<HS>::ALIGNED || row[u] < nrows)) {
                    float4 v = rg[i];
                    if constexpr (SCALE) { v.x = sc[u] * v.x; v.y = sc[u] * v.y; v.z = sc[u] * v.z; v.w = sc[u] * v.w; }
                    *reinterpret_cast<float4*>(tile + row[u] * RS + col[u] * 4) = v;
                }
            }
        }
    }
    if (patch_t >= t0 && patch_t < t0 + ct) {                                // wave-uniform
        const int fb = (patch_t - t0) * HS4, d = ((int)threadIdx.x - fb) & (kBlock - 1);
        if (d < HS4) {
            float4 v = *reinterpret_cast<const float4*>(patch + d * 4);
            if constexpr (SCALE) { const float ap = rowscale[patch_t - t0]; v.x = ap * v.x; v.y = ap * v.y; v.z = ap * v.z; v.w = ap * v.w; }
            *reinterpret_cast<float4*>(tile + (patch_t - t0) * RS + d * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Attention for one new token: RoPE + scores + softmax + weighted V  (transformer.rs:443-544)
// One workgroup per query head, head size HS a compile-time constant.  The reference's sums are sequential - the dot
// product over the head dims, softmax's sum over t, the V accumulation over t - and float addition is not associative, so
// each stays one serial chain of adds on one lane; everything order-free runs in parallel: the scores across t (one lane
// per timestep), max, exp, divide, the HS output dims (one lane per dim) and the products a_t * v_t[d].
//
// K: the score chain of timestep t wants k_t[0], k_t[1], .. in ONE lane, so the K cache is kept in the order the lanes
// read it: k_cache[layer][kv head][d / 4][t][d % 4].  Lane t loads 16 bytes (4 consecutive dims of its own key) per
// instruction, consecutive lanes consecutive 16-byte words: the keys go straight from memory into the lane that runs the
// chain - no LDS staging, no transposition, no barrier - and the first batch is in flight across the RoPE step.  (The cache
// is internal to the library; the reference's API never exposes it.)  The new key, rotated here, is stored to the cache by
// every head of its kv head (identical values) before the loads that may hit it.
// V keeps the reference's [t][kv_dim] rows.  The chain of output dim d wants v_0[d], v_1[d], ..: rows are loaded by all
// 256 lanes (16 B each, coalesced per row), scaled by a_t while they are written to an LDS tile - so the serial part is
// add-only - and lane d walks down its column.  The whole workgroup's registers are the prefetch buffer: the first chunk
// of CH rows is in flight from the top of the kernel, the next chunk while the current one is consumed.
// COH: q, the raw key and the V row of this position were produced by other workgroups of the SAME launch
// (persistent engine) and are read / the output written with agent-scope accesses; the V row of `pos` is
// then patched in from a coherent read instead of the (possibly stale) prefetched copy.
// ------------------------------------------------------------------------------------------------
#define ATT_STAMP(k) do { if (a.dbg && threadIdx.x == 0 && blockIdx.x == 0) a.dbg[k] = wall_clock64(); } while (0)

// Inputs of the RoPE step that the stand-alone kernel loads itself, ahead of its K/V loads (a CU returns loads in
// request order: issued after them, these few bytes would only arrive behind the whole first K/V chunk).
struct AttPre { float q0, q1, k0, k1; float2 cs; };

typedef float f32x4v __attribute__((ext_vector_type(4)));
// K of one kv head, blocked for the score lanes: [HS / 4][seq_len][4]
__device__ __forceinline__ float* att_k_head(const AttnArgs& a, int kvh, int hs) {
    return a.k_cache + ((size_t)a.layer * a.n_kv_heads + kvh) * hs * (size_t)a.seq_len;
}
// NG 16-byte words of this lane's key (dim groups g0 .. g0+NG-1): SGPR resource + scalar group offset + lane offset,
// one instruction per load and no address arithmetic
template <int NG>
__device__ __forceinline__ void att_kload(f32x4v (&dst)[NG], __amdgpu_buffer_rsrc_t krs, int voff, int g0, int S) {
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(krs, voff, (g0 + u) * S * 16, 0);
        dst[u] = f32x4v{__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
    }
}
template <int NG>
__device__ __forceinline__ float att_kuse(float score, const f32x4v (&src)[NG], const float4* q4, int g0) {
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const float4 qq = q4[g0 + u];
        float pr;
        pr = qq.x * src[u].x; score = score + pr;
        pr = qq.y * src[u].y; score = score + pr;
        pr = qq.z * src[u].z; score = score + pr;
        pr = qq.w * src[u].w; score = score + pr;
    }
    return score;
}
// The score chain of this lane's timestep: kk holds the first batch of its key (already loaded), further batches
// ping-pong between kk and kb, one batch of loads ahead of the adds.  The running sum passes through an opaque asm after
// every batch: that anchors the batch's adds between its neighbours' loads (left alone, instruction selection sinks all the
// arithmetic below all the loads, and a whole key + q in registers does not fit for the big heads).
template <int HS, int NG>
__device__ __forceinline__ float att_score_chain(f32x4v (&kk)[NG], __amdgpu_buffer_rsrc_t krs, int voff, const float* q, int S) {
    constexpr int HS4 = HS / 4;
    f32x4v kb[NG];
    float score = 0.0f;
    const float4* q4 = reinterpret_cast<const float4*>(q);
#pragma unroll
    for (int g0 = 0; g0 < HS4; g0 += 2 * NG) {
        if (g0 + NG < HS4) att_kload<NG>(kb, krs, voff, g0 + NG, S);
        score = att_kuse<NG>(score, kk, q4, g0);
        asm volatile("" : "+v"(score) : : "memory");
        if (g0 + NG < HS4) {
            if (g0 + 2 * NG < HS4) att_kload<NG>(kk, krs, voff, g0 + 2 * NG, S);
            score = att_kuse<NG>(score, kb, q4, g0 + NG);
            asm volatile("" : "+v"(score) : : "memory");
        }
    }
    return score;
}

// GEMMA: score soft-cap + window mask (a compile-time switch: the f64 tanh is ~230 instructions and a dozen registers)
// ROT (batched prefill): q is already rotated and the key of `pos` already in the cache (rope_rows_kernel).
// TAG (merged qkv + attention launch): q, the raw key and the value row of this position are produced by the GEMV workgroups of
// the SAME launch as 8-byte {value, tag} granules (EPI_QKV_TAG); the lanes that need them poll the granules themselves, after
// all K / V loads of the earlier positions have been issued.
struct AttTag { const unsigned long long* gran; unsigned tag; int att_dim, kv_dim; int* err; };
template <int HS, int NF, bool COH, bool PRE = false, bool GEMMA = false, bool ROT = false, bool TAG = false>
__device__ __forceinline__ void attention_body(const AttnArgs& a, int h, int pos, char* smem, uint64_t etab, const AttPre& pre = AttPre(), const AttTag& tg = AttTag()) {
    static_assert(!PRE || HS / 2 <= kBlock, "one RoPE pair per lane");
    static_assert(!TAG || (HS <= kBlock && !COH && !PRE && !ROT), "granule polling: one value of each vector per lane");
    constexpr int half = HS / 2, RS = HS + 4;
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul;
    const int kv_dim = a.n_kv_heads * HS;
    const int T = pos + 1;
    const int tid = threadIdx.x;
    const int CH = a.chunk;
    float* q = reinterpret_cast<float*>(smem);        // HS
    float* kn = q + HS;                               // HS: rotated key of this position
    float* vn = kn + HS;                              // HS: value row of this position (COH only)
    float* red = vn + HS;                             // 16 floats of reduction scratch
    float* tile = red + 16;                           // CH (+16) rows of RS floats (RS = HS + 4: conflict-free float4 row reads)
    float* att = tile + (size_t)(CH + 32) * RS;       // T (+32 floats of zero padding, +32 of read-ahead)
    const size_t loff = (size_t)a.layer * a.seq_len * kv_dim;
    const int nchunks = (T + CH - 1) / CH;
    const float* vbase = a.v_cache + loff + kvh * HS;
    const int S = a.seq_len;
    float* kT = att_k_head(a, kvh, HS);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(kT, 0, HS * S * 4, 0x00020000);
    // ---- this lane's key (timesteps 0..255), first batch of dims, and the first V chunk: in flight across the RoPE step
    constexpr int KG = HS / 4 <= 32 ? HS / 4 : 16;    // 16-byte words per batch
    static_assert((HS / 4) % KG == 0, "head size");
    const int tc0 = tid < T ? tid : T - 1;            // lanes past the sequence re-read its last key (no predication)
    f32x4v kk[KG];
    att_kload<KG>(kk, krs, tc0 * 16, 0, S);
    float4 vreg[NF];
    att_gload<HS, NF>(vreg, vbase, 0, T, CH, kv_dim);  // row `pos` of V was stored by the QKV kernel (COH: patched from vn)

    if constexpr (ROT) { for (int j = tid; j < HS; j += kBlock) q[j] = a.q[h * HS + j]; }
    float2 cs_tag = make_float2(0.f, 0.f);
    if constexpr (TAG) {
        cs_tag = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + (tid < half ? tid : 0)) * 2);   // ahead of the polls
        if (tid < HS) {
            const unsigned long long* gq = tg.gran + h * HS + tid;
            const unsigned long long* gk = tg.gran + tg.att_dim + kvh * HS + tid;
            const unsigned long long* gv = tg.gran + tg.att_dim + tg.kv_dim + kvh * HS + tid;
            unsigned long long x0, x1, x2, y0, y1, y2;
            auto fresh = [&](unsigned long long p0, unsigned long long p1, unsigned long long p2) __attribute__((always_inline)) {
                return __all((unsigned)(p0 >> 32) == tg.tag && (unsigned)(p1 >> 32) == tg.tag && (unsigned)(p2 >> 32) == tg.tag) != 0;
            };
            // two sweeps in flight (a sweep's latency, not up to twice it, after the store)
            x0 = __hip_atomic_load(gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x1 = __hip_atomic_load(gk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x2 = __hip_atomic_load(gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spins = 0;; ++spins) {
                y0 = __hip_atomic_load(gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                y1 = __hip_atomic_load(gk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                y2 = __hip_atomic_load(gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fresh(x0, x1, x2)) break;
                x0 = __hip_atomic_load(gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x1 = __hip_atomic_load(gk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x2 = __hip_atomic_load(gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fresh(y0, y1, y2)) { x0 = y0; x1 = y1; x2 = y2; break; }
                if (spins > kTagSpinMax || (spins & 1023) == 1023) {     // bounded: report and finish with garbage instead of hanging
                    const int e = __hip_atomic_load(tg.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (e != 0 || spins > kTagSpinMax) {
                        if (e == 0) __hip_atomic_store(tg.err, a.layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            q[tid] = __uint_as_float((unsigned)x0); kn[tid] = __uint_as_float((unsigned)x1); vn[tid] = __uint_as_float((unsigned)x2);
        }
        lds_barrier();
        ATT_STAMP(1);
    }
    // RoPE (transformer.rs:480-491) with the host-built (fcr, fci) table
    for (int j = tid; j < (ROT ? 0 : half); j += kBlock) {
        float2 cs;
        if constexpr (PRE) cs = pre.cs; else if constexpr (TAG) cs = cs_tag; else cs = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + j) * 2);
        const float fcr = cs.x, fci = cs.y;
        if constexpr (TAG) {                                   // raw values in place in LDS; this thread owns both halves of pair j
            {
                const float v0 = q[j], v1 = q[j + half];
                const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
                q[j] = a0 - a1; q[j + half] = b0 + b1;
            }
            const float v0 = kn[j], v1 = kn[j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            const float r0 = a0 - a1, r1 = b0 + b1;
            kn[j] = r0; kn[j + half] = r1;
            kT[(((size_t)(j >> 2) * S + pos) << 2) + (j & 3)] = r0;
            kT[(((size_t)((j + half) >> 2) * S + pos) << 2) + ((j + half) & 3)] = r1;
            continue;
        }
        {
            const float v0 = PRE ? pre.q0 : ld_f32<COH>(a.q + h * HS + j), v1 = PRE ? pre.q1 : ld_f32<COH>(a.q + h * HS + j + half);
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            q[j] = a0 - a1; q[j + half] = b0 + b1;
        }
        {
            const float v0 = PRE ? pre.k0 : ld_f32<COH>(a.k_raw + kvh * HS + j), v1 = PRE ? pre.k1 : ld_f32<COH>(a.k_raw + kvh * HS + j + half);
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            const float r0 = a0 - a1, r1 = b0 + b1;
            kn[j] = r0; kn[j + half] = r1;
            // Every head of the kv head stores the (identical) new key: the score loads below then find it in memory like
            // every other key - the waves of a workgroup share the CU's L1, so store, vmcnt(0), barrier, load is coherent -
            // and later launches read it from there anyway.
            kT[(((size_t)(j >> 2) * S + pos) << 2) + (j & 3)] = r0;
            kT[(((size_t)((j + half) >> 2) * S + pos) << 2) + ((j + half) & 3)] = r1;
        }
        if constexpr (COH) {
            vn[j] = ld_f32<true>(vbase + (size_t)pos * kv_dim + j);
            vn[j + half] = ld_f32<true>(vbase + (size_t)pos * kv_dim + j + half);
        }
    }
    // the key stores must have reached the cache hierarchy before a lane loads the key of `pos` from memory: further batches of the key
    // (heads wider than one batch) or further passes (T > 256).  Otherwise that lane's whole key is patched from LDS below and the wait -
    // a memory round trip on the critical path of every layer - is skipped (merged launch, short contexts).
    if (!TAG || HS / 4 > KG || T > kBlock) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    ATT_STAMP(2);
    if (!ROT && tc0 == pos) {                          // batch 0 was loaded before the new key existed: patch it from LDS
#pragma unroll
        for (int u = 0; u < KG; ++u) { const float4 t4 = reinterpret_cast<const float4*>(kn)[u]; kk[u] = f32x4v{t4.x, t4.y, t4.z, t4.w}; }
    }

    // ---- scores (transformer.rs:507-529): one lane per t, sequential dot over the head dims
    // Gemma's window test uses the position of the FIRST token of a forward_layer call (`pos`, not pos + i, :525): for the
    // later tokens of a batched call (fill_kv_cache) pos - t wraps around in u32 and the keys after the first token get the
    // mask value.  win_base >= 0 carries that first position; decode (one token per call) has win_base < 0.
    int wpos = pos;
    if constexpr (GEMMA) { const int wb = a.st->win_base; wpos = wb >= 0 ? wb : pos; }
    const float sqrt_hs = sqrtf((float)HS);
    float lmax = __uint_as_float(0xff800000u);
    auto finish_score = [&](float score, int t) __attribute__((always_inline)) {
        score = score / sqrt_hs;
        if constexpr (GEMMA) {                         // transformer.rs:518-526
            score = score / 50.0f;
            score = (float)tanh((double)score);
            score = score * 50.0f;
            score = score + (((unsigned)(wpos - t) <= 4096u) ? 0.0f : -2.3819763e38f);   // :525, u32 arithmetic as in the reference
        }
        if (t < T) { att[t] = score; lmax = fmaxf(lmax, score); }
    };
    finish_score(att_score_chain<HS, KG>(kk, krs, tc0 * 16, q, S), tid);                 // timesteps 0..255
    for (int t0 = kBlock; t0 < T; t0 += kBlock) {                                        // longer sequences: further passes
        const int t = t0 + tid, tc = t < T ? t : T - 1;
        f32x4v k0[KG];
        att_kload<KG>(k0, krs, tc * 16, 0, S);
        finish_score(att_score_chain<HS, KG>(k0, krs, tc * 16, q, S), t);
    }
    ATT_STAMP(4);
    // softmax (functional.rs:122-140): max (order-free), exp, sequential sum, divide
    lmax = wave64_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    lds_barrier();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int t0 = 0; t0 < T; t0 += kBlock) {                  // whole waves call expf together (it shuffles)
        const int t = t0 + tid;
        const float e = expf_glibc_t(t < T ? att[t] - mx : 0.0f, etab);
        if (t < T) att[t] = e;
    }
    if (tid < 64) att[T + tid] = 0.0f;                       // +0.0 padding for the serial sum (exact: the running sum of exponentials is >= +0)
    lds_barrier();
    if (tid < 64) {
        // serial sum (functional.rs:134) in wave 0: 64 exponentials per register, added lane by lane through the add's own DPP operand
        // (wave_serial_sum: one dependent add per term and nothing else - the LDS-fed chain cost ~7 cycles per term)
        float sum = 0.0f;
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int left = T - t0;
            sum = wave_serial_sum(sum, att[t0 + tid], left >= 64 ? 4 : (left + 15) >> 4);
        }
        if (tid == 0) red[4] = sum;
    }
    lds_barrier();
    const float sum = red[4];
    for (int t = tid; t < T; t += kBlock) att[t] = att[t] / sum;
    lds_barrier();
    ATT_STAMP(5);

    // weighted sum of values (transformer.rs:533-541): products a_t * v_t[d] by all lanes into the tile, then one
    // lane per output dim adds them up sequentially over t
    float o = 0.0f;
    for (int c = 0; c < nchunks; ++c) {
        const int t0 = c * CH, ct = (T - t0) < CH ? (T - t0) : CH;
        att_tstore<HS, NF, true>(vreg, tile, att + t0, t0, T, CH, (COH || TAG) ? pos : -1, vn);
        lds_barrier();
        if (c + 1 < nchunks) att_gload<HS, NF>(vreg, vbase, t0 + CH, T, CH, kv_dim);
        if (c == 0) ATT_STAMP(6);
        if (tid < HS) {
            // serial chain of adds over t; rows ct .. ceil16(ct)-1 of the tile hold +-0.0 products (att_tstore), exact
            // no-ops for an accumulator that starts at +0.0 and therefore can never be -0.0
            o = serial_sum16<RS, false>(o, tile + tid, ct);
        }
        lds_barrier();
    }
    if (tid < HS) st_f32<COH>(a.out + h * HS + tid, o);
    ATT_STAMP(7);
}

template <int HS, bool GEMMA>
__global__ __launch_bounds__(kBlock) void attention_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = blockIdx.x, kvh = h / (a.n_heads / a.n_kv_heads);
    ATT_STAMP(0);
    const uint64_t etab = exp2f_tab_lane();
    constexpr int half = HS / 2;
    const int j = threadIdx.x < half ? (int)threadIdx.x : 0;
    const int pos = a.st->pos;                              // first: the K/V loads depend on it
    AttPre pre;                                             // RoPE inputs ahead of the K/V loads (a CU returns loads in request order)
    pre.q0 = a.q[h * HS + j]; pre.q1 = a.q[h * HS + j + half];
    pre.k0 = a.k_raw[kvh * HS + j]; pre.k1 = a.k_raw[kvh * HS + j + half];
    pre.cs = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + j) * 2);
    ATT_STAMP(1);
    attention_body<HS, kAttF4, false, true, GEMMA>(a, h, pos, smem, etab, pre);
}

static size_t attention_smem(int head_size, int chunk, int seq_len) {
    return (size_t)(3 * head_size + 16 + (size_t)(chunk + 32) * (head_size + 4) + ((seq_len + 3) & ~3) + 64) * 4;   // +16 tile rows / +32 floats: serial_sum16's read-ahead
}

int attention_chunk(int head_size) {
    int ch = (16384 / head_size) & ~31;
    if (ch > 256) ch = 256;                 // (smaller chunks / LDS footprints were measured: no effect on the launch gap)
    return ch < 32 ? 32 : ch;
}

template <int HS, bool GEMMA>
static hipError_t launch_attention_hsg(const AttnArgs& a, size_t smem, hipStream_t s) {
    allow_big_lds(reinterpret_cast<const void*>(attention_kernel<HS, GEMMA>));
    LMRS_LAUNCH_GRID((attention_kernel<HS, GEMMA>), dim3(a.n_heads), kBlock, smem, s, a);
    return hipGetLastError();
}
template <int HS>
static hipError_t launch_attention_hs(const AttnArgs& a, size_t smem, hipStream_t s) {
    return a.gemma ? launch_attention_hsg<HS, true>(a, smem, s) : launch_attention_hsg<HS, false>(a, smem, s);
}

hipError_t launch_attention(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    a.chunk = attention_chunk(a.head_size);
    if (a.chunk * (a.head_size / 4) > kAttF4 * kBlock) return hipErrorInvalidValue;
    const size_t smem = attention_smem(a.head_size, a.chunk, a.seq_len);
    switch (a.head_size) {                                      // head sizes of the supported model families
        case 64: return launch_attention_hs<64>(a, smem, s);    // Llama-3.2-1B, tiny test models
        case 96: return launch_attention_hs<96>(a, smem, s);    // Phi-3.5
        case 128: return launch_attention_hs<128>(a, smem, s);  // Llama-3.2-3B
        case 256: return launch_attention_hs<256>(a, smem, s);  // Gemma-2
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// qkv projection + attention of one layer in ONE launch (short contexts).
// As two launches this edge costs a dependent kernel boundary (1.6 us) plus the attention kernel's own ramp (position -> K / V
// loads -> q: 1.2 us before its RoPE step) although only the 4 query heads + 1 key + 1 value head of a kv group meet in an
// attention workgroup - a few-to-one edge, not an all-to-all one.  Here the n_heads attention workgroups are the FIRST workgroups
// of the grid: they issue every K / V load of the earlier positions, then poll the 3 x HS {value, tag} granules they need
// (Guideline-16 "R2": one aligned 8-byte write-through store per value, the data is the flag - no counter, no fence, correct under
// any workgroup placement); the GEMV workgroups behind them are the unchanged static kernel body with the EPI_QKV_TAG epilogue.
// The attention workgroups never block a GEMV workgroup (nobody waits for them inside the launch), so the launch cannot deadlock
// whatever the residency; every poll loop is bounded (err).  Arithmetic: the bodies of the separate kernels - bit-identical.
// ------------------------------------------------------------------------------------------------
// ---- one WAVE per query head (merged launch, the shortest contexts: T = pos + 1 <= TW) -------------------------------------------
// A whole workgroup per head pays a workgroup barrier (and an LDS round trip) between any two phases for a context of a few dozen
// keys, where each phase is a handful of instructions.  Here a single wave runs the head start to finish with no barrier at all:
//   before q arrives (free time, the qkv workgroups are still streaming): the keys of the earlier positions go into an LDS tile
//   kt[dim group][t][4] (the blocked cache layout: lane t later reads its own key 16 bytes at a time, conflict-free), the value rows
//   into REGISTERS, v[t] = v_t[lane] (one coalesced 256-byte row per load): the chain lane of output dim d already holds its column;
//   then: poll the granules (lane d: q[d], k[d], v[d]), RoPE through LDS, scores one lane per key (pass p: keys 64p .. 64p+63),
//   max by DPP, exp, the sum chain run redundantly by every lane (no broadcast), divide, and the V chain o += a_t * v[t] with the
//   weights read back four at a time as LDS broadcasts.  Slots past the sequence carry weight +0.0 and value 0.0: exact no-ops.
// Same operations in the same order per value as attention_body: bit-identical.
template <int HS> struct WaveGeom {
    static constexpr int HS4 = HS / 4, ND = (HS + 63) / 64, NH2 = (HS / 2 + 63) / 64;
    static constexpr int TW = HS <= 64 ? 128 : 64;                  // longest context: ND * TW value registers, HS * TW * 4 bytes of keys in LDS
    static constexpr int NPASS = TW / 64;
    static constexpr size_t SMEM = (size_t)(2 * HS + TW + 16) * 4 + (size_t)HS4 * TW * 16;
};
// longest context (pos + 1) of the wave forms; 0: no wave class (Gemma's 256-wide heads: the four-wave form measured the same as the workgroup form there)
constexpr int qa_wave_T(int hs) { return hs == 64 ? 64 * 4 : ((hs == 96 || hs == 128) ? 128 : 0); }

template <int HS, bool GEMMA>
__device__ __forceinline__ void attention_wave_tag(const AttnArgs& a, const int h, const int pos, char* smem, const uint64_t etab, const AttTag& tg) {
    using W = WaveGeom<HS>;
    constexpr int HS4 = W::HS4, ND = W::ND, NH2 = W::NH2, TW = W::TW, NPASS = W::NPASS, half = HS / 2;
    const int lane = threadIdx.x & 63;                              // one wave per head (the caller retired the workgroup's other waves)
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul, kv_dim = a.n_kv_heads * HS;
    const int T = pos + 1, S = a.seq_len;
    float* qs = reinterpret_cast<float*>(smem);                     // HS: rotated query
    float* kn = qs + HS;                                            // HS: raw, then rotated key of this position
    float* att = kn + HS;                                           // TW + 16: exponentials, then weights
    float4* kt = reinterpret_cast<float4*>(att + TW + 16);          // [HS4][TW] x 16 bytes
    float* kT = att_k_head(a, kvh, HS);
    const float* vbase = a.v_cache + (size_t)a.layer * S * kv_dim + kvh * HS;
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[0] = wall_clock64();

    // ---- RoPE terms of this position, keys and values of the earlier ones: all before the first poll
    float2 cs[NH2];
#pragma unroll
    for (int i = 0; i < NH2; ++i) { const int j = lane + 64 * i; cs[i] = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + (j < half ? j : 0)) * 2); }
    // Keys: LDS-DMA (global_load_lds_dwordx4: lane t's 16 bytes land at tile + 16 t, no registers, nothing to wait for until the
    // scores).  Rows past the sequence are read as they lie in the cache (the host guarantees TW <= seq_len): those lanes' scores are
    // replaced below, and slot `pos` is overwritten with the rotated new key once the DMA has landed.
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (64 * p < pos) {                                         // wave-uniform
#pragma unroll
            for (int g = 0; g < HS4; ++g)
                __builtin_amdgcn_global_load_lds((const LMRS_GLOBAL void*)(kT + ((size_t)g * S + 64 * p + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(kt + g * TW + 64 * p), 16, 0, 0);
        }
    }
    // Values: v[t] = v_t[this lane's dims], one 256-byte row per load (buffer loads: the row offset is a scalar), blocks of 16 rows
    // NESTED so that the whole prefetch has ONE join (a join after every block made hipcc drain the loads there: 5.8 us for 100 rows).
    float v[ND][TW];
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vbase), 0, 0x7fffffff, 0x00020000);
    const int vrow = kv_dim * 4;
    auto vblock = [&](auto self, auto t0c) __attribute__((always_inline)) -> void {
        constexpr int t0 = decltype(t0c)::value;
        if constexpr (t0 < TW) {
            if (t0 < pos) {                                         // wave-uniform
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < ND; ++i) {
                        const int d = lane + 64 * i;
                        v[i][t0 + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, (d < HS ? d : 0) * 4, (t0 + u) * vrow, 0));
                    }
                self(self, std::integral_constant<int, t0 + 16>());
            }
        }
    };
    vblock(vblock, std::integral_constant<int, 0>());
    // slots past the sequence: 0.0 (what lies in the cache there must not meet its +0.0 weight as Inf / NaN); pinned HERE, ahead of the
    // polls (left alone the selects sink into the V chain, one scalar compare + select per add on the critical path)
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int i = 0; i < ND; ++i) { v[i][t] = t < pos ? v[i][t] : 0.0f; asm volatile("" : "+v"(v[i][t])); }
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[1] = wall_clock64();

    // ---- q, raw k, v of this position: poll the granules (two sweeps in flight: a sweep's latency, not twice it, after the store)
    unsigned long long xg[3 * ND];
    {
        const unsigned long long* gp[3 * ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = lane + 64 * i, dc = d < HS ? d : 0;
            gp[3 * i] = tg.gran + h * HS + dc; gp[3 * i + 1] = tg.gran + tg.att_dim + kvh * HS + dc; gp[3 * i + 2] = tg.gran + tg.att_dim + tg.kv_dim + kvh * HS + dc;
        }
        auto sweep = [&](unsigned long long (&x)[3 * ND]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 3 * ND; ++k) x[k] = __hip_atomic_load(gp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto fresh = [&](const unsigned long long (&x)[3 * ND]) __attribute__((always_inline)) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 3 * ND; ++k) ok = ok && (unsigned)(x[k] >> 32) == tg.tag;
            return __all(ok) != 0;
        };
        unsigned long long xa[3 * ND], xb[3 * ND];
        sweep(xa);
        for (unsigned spins = 0;; ++spins) {
            sweep(xb);
            if (fresh(xa)) {
#pragma unroll
                for (int k = 0; k < 3 * ND; ++k) xg[k] = xa[k];
                break;
            }
            sweep(xa);
            if (fresh(xb)) {
#pragma unroll
                for (int k = 0; k < 3 * ND; ++k) xg[k] = xb[k];
                break;
            }
            if (spins > kTagSpinMax || (spins & 1023) == 1023) {     // bounded: report and finish with garbage instead of hanging
                const int e = __hip_atomic_load(tg.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e != 0 || spins > kTagSpinMax) {
                    if (e == 0) __hip_atomic_store(tg.err, a.layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < 3 * ND; ++k) xg[k] = xb[k];
                    break;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (loads return in order: the key tile's DMA landed before the granules did)
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[2] = wall_clock64();
    float vnew[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = lane + 64 * i;
        vnew[i] = __uint_as_float((unsigned)xg[3 * i + 2]);
        if (d < HS) { qs[d] = __uint_as_float((unsigned)xg[3 * i]); kn[d] = __uint_as_float((unsigned)xg[3 * i + 1]); }
    }
    // RoPE (transformer.rs:480-491): this lane owns both halves of pair j; the rotated key goes to the cache (for the later steps)
    // and into the LDS tile at t = pos like every other key
#pragma unroll
    for (int i = 0; i < NH2; ++i) {
        const int j = lane + 64 * i;
        if (j < half) {
            const float fcr = cs[i].x, fci = cs[i].y;
            {
                const float v0 = qs[j], v1 = qs[j + half];
                const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
                qs[j] = a0 - a1; qs[j + half] = b0 + b1;
            }
            const float v0 = kn[j], v1 = kn[j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            const float r0 = a0 - a1, r1 = b0 + b1;
            kT[(((size_t)(j >> 2) * S + pos) << 2) + (j & 3)] = r0;
            kT[(((size_t)((j + half) >> 2) * S + pos) << 2) + ((j + half) & 3)] = r1;
            reinterpret_cast<float*>(kt)[(((j >> 2) * TW + pos) << 2) + (j & 3)] = r0;
            reinterpret_cast<float*>(kt)[((((j + half) >> 2) * TW + pos) << 2) + ((j + half) & 3)] = r1;
        }
    }
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[3] = wall_clock64();

    // ---- scores (transformer.rs:507-529): one lane per key, sequential dot over the head dims
    int wpos = pos;
    if constexpr (GEMMA) { const int wb = a.st->win_base; wpos = wb >= 0 ? wb : pos; }
    const float sqrt_hs = sqrtf((float)HS), ninf = __uint_as_float(0xff800000u);
    float sc[NPASS];
    float lmax = ninf;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        sc[p] = ninf;
        if (64 * p < T) {                                           // wave-uniform
            const int t = 64 * p + lane;
            float score = 0.0f;
            // batches of 4 dim groups (8 x 16-byte LDS reads) ping-pong one ahead of the adds; the running sum passes through an
            // opaque asm per batch so that the next batch's reads are ISSUED before this batch's arithmetic
            constexpr int GB = 4;
            static_assert(HS4 % (2 * GB) == 0, "head size");
            float4 ka[GB], qa[GB], kb[GB], qb[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) { ka[u] = kt[u * TW + t]; qa[u] = reinterpret_cast<const float4*>(qs)[u]; }
#pragma unroll
            for (int g0 = 0; g0 < HS4; g0 += 2 * GB) {
#pragma unroll
                for (int u = 0; u < GB; ++u) { kb[u] = kt[(g0 + GB + u) * TW + t]; qb[u] = reinterpret_cast<const float4*>(qs)[g0 + GB + u]; }
                asm volatile("" : "+v"(score) : : "memory");
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    float pr;
                    pr = qa[u].x * ka[u].x; score = score + pr;
                    pr = qa[u].y * ka[u].y; score = score + pr;
                    pr = qa[u].z * ka[u].z; score = score + pr;
                    pr = qa[u].w * ka[u].w; score = score + pr;
                }
                if (g0 + 2 * GB < HS4) {
#pragma unroll
                    for (int u = 0; u < GB; ++u) { ka[u] = kt[(g0 + 2 * GB + u) * TW + t]; qa[u] = reinterpret_cast<const float4*>(qs)[g0 + 2 * GB + u]; }
                }
                asm volatile("" : "+v"(score) : : "memory");
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    float pr;
                    pr = qb[u].x * kb[u].x; score = score + pr;
                    pr = qb[u].y * kb[u].y; score = score + pr;
                    pr = qb[u].z * kb[u].z; score = score + pr;
                    pr = qb[u].w * kb[u].w; score = score + pr;
                }
            }
            score = score / sqrt_hs;
            if constexpr (GEMMA) {                                  // transformer.rs:518-526
                score = score / 50.0f;
                score = (float)tanh((double)score);
                score = score * 50.0f;
                score = score + (((unsigned)(wpos - t) <= 4096u) ? 0.0f : -2.3819763e38f);
            }
            sc[p] = t < T ? score : ninf;
            lmax = fmaxf(lmax, sc[p]);
        }
    }
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[4] = wall_clock64();
    // ---- softmax (functional.rs:122-140): max (order-free), exp, sequential sum, divide
    const float mx = wave64_max(lmax);
    float ex[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        ex[p] = 0.0f;
        if (64 * p < T) {                                           // wave-uniform: whole waves call expf together (it shuffles)
            const int t = 64 * p + lane;
            const float e = expf_glibc_t(t < T ? sc[p] - mx : 0.0f, etab);
            ex[p] = t < T ? e : 0.0f;
        }
    }
    float sum = 0.0f;                                               // sequential over t: in registers, lane by lane (wave_serial_sum)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (64 * p < T) {                                           // wave-uniform; the lanes past T hold +0.0 (exact: the running sum is >= +0)
            const int left = T - 64 * p;
            sum = wave_serial_sum(sum, ex[p], left >= 64 ? 4 : (left + 15) >> 4);
        }
    }
    float ap_lane = 0.0f;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (64 * p < T) {
            const int t = 64 * p + lane;
            const float w = ex[p] / sum;
            att[t] = t < pos ? w : 0.0f;                             // the chain below covers the earlier positions; this one follows from registers
            if ((pos >> 6) == p) ap_lane = w;
        }
    }
    const float a_pos = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ap_lane), pos & 63));
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[5] = wall_clock64();
    // ---- weighted sum of values (transformer.rs:533-541), t ascending; this lane's dims
    float o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = 0.0f;
#pragma unroll
    for (int t0 = 0; t0 < TW; t0 += 16) {
        if (t0 < pos) {                                             // wave-uniform
            float4 w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = reinterpret_cast<const float4*>(att)[t0 / 4 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    float pr;
                    pr = w4[u].x * v[i][t0 + 4 * u + 0]; o[i] = o[i] + pr;
                    pr = w4[u].y * v[i][t0 + 4 * u + 1]; o[i] = o[i] + pr;
                    pr = w4[u].z * v[i][t0 + 4 * u + 2]; o[i] = o[i] + pr;
                    pr = w4[u].w * v[i][t0 + 4 * u + 3]; o[i] = o[i] + pr;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = lane + 64 * i;
        const float pr = a_pos * vnew[i];
        o[i] = o[i] + pr;
        if (d < HS) a.out[h * HS + d] = o[i];
    }
    if (a.dbg && lane == 0 && blockIdx.x == 0) a.dbg[7] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// The wave form for 64-wide heads with TWO waves per head from position 64 on (contexts up to 128: Llama-3.2-1B).  Below 64 positions wave 1
// leaves at once and wave 0 runs the head exactly as attention_wave_tag does.  From 64 on the head's work is cut by KEY RANGE - wave w owns keys
// 64 w .. 64 w + 63: their K tile rows (LDS-DMA), their value rows (registers: 64 buffer loads per wave instead of 128 in one - the prefetch of a
// 100-position context used to land 2.6 us after launch, later than the last qkv row), their scores and exponentials - and the reference's
// sequential chains cross the waves in order through LDS: the maximum (order-free), the softmax sum (wave 0's 64 terms, then wave 1 continues
// from its partial), and the value chain o += a_t v_t: wave 0 adds its 64 terms while wave 1 forms the PRODUCTS a_t v_t of its keys (the
// multiplies are independent - only the adds are the chain) and parks them in LDS; wave 0 then adds them in key order.  Both waves poll q / k / v
// and rotate q for themselves (no hand-off before the scores); wave 0 alone writes the rotated key to the cache.  Four lds_barrier()s between
// two waves when both are alive, none otherwise.  Same adds in the same order as every other form: bit-equal.
// ------------------------------------------------------------------------------------------------
constexpr size_t kPairSmem = (size_t)(2 * 2 * 64 + 128 + 16 + 16) * 4 + (size_t)16 * 128 * 16 + (size_t)64 * 64 * 4;   // q / k per wave, weights, hand-off words, K tile, products
template <bool GEMMA>
__device__ __forceinline__ void attention_pair_tag(const AttnArgs& a, const int h, const int pos, char* smem, const uint64_t etab, const AttTag& tg) {
    constexpr int HS = 64, HS4 = 16, TW = 128, half = 32;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool two = pos >= 64;                                     // (uniform over the workgroup)
    if (wv > (two ? 1 : 0)) return;
    const int tb = two ? 64 * wv : 0;                               // first key of this wave
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul, kv_dim = a.n_kv_heads * HS;
    const int T = pos + 1, S = a.seq_len;
    float* base = reinterpret_cast<float*>(smem);
    float* qs = base + wv * 2 * HS;                                 // this wave's rotated query
    float* kn = qs + HS;                                            // ... raw, then rotated key of this position
    float* att = base + 4 * HS;                                     // TW + 16 weights (shared)
    float* red = att + TW + 16;                                     // 16 hand-off words: [0..1] maxima, [2] wave 0's sum, [3] the sum, [4] a_pos
    float4* kt = reinterpret_cast<float4*>(red + 16);               // [HS4][TW] x 16 bytes (shared)
    float* prod = reinterpret_cast<float*>(kt + HS4 * TW);          // [64 keys of wave 1][HS]: a_t * v_t
    float* kT = att_k_head(a, kvh, HS);
    const float* vbase = a.v_cache + (size_t)a.layer * S * kv_dim + kvh * HS;
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[0] = wall_clock64();

    const float2 cs = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + (lane < half ? lane : 0)) * 2);
    if (tb < pos) {                                                 // this wave's 64 keys by LDS-DMA (rows past the sequence are read as they lie in the cache: seq_len >= 128)
#pragma unroll
        for (int g = 0; g < HS4; ++g)
            __builtin_amdgcn_global_load_lds((const LMRS_GLOBAL void*)(kT + ((size_t)g * S + tb + lane) * 4),
                                             (__attribute__((address_space(3))) void*)(kt + g * TW + tb), 16, 0, 0);
    }
    // this wave's value rows: v[u] = v_{tb + u}[lane], blocks of 16 rows nested so that the prefetch has one join (see attention_wave_tag)
    float v[64];
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vbase), 0, 0x7fffffff, 0x00020000);
    const int vrow = kv_dim * 4, vb0 = tb * vrow;
    auto vblock = [&](auto self, auto u0c) __attribute__((always_inline)) -> void {
        constexpr int u0 = decltype(u0c)::value;
        if constexpr (u0 < 64) {
            if (tb + u0 < pos) {                                    // wave-uniform
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u0 + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, lane * 4, vb0 + (u0 + u) * vrow, 0));
                self(self, std::integral_constant<int, u0 + 16>());
            }
        }
    };
    vblock(vblock, std::integral_constant<int, 0>());
#pragma unroll
    for (int u = 0; u < 64; ++u) { v[u] = tb + u < pos ? v[u] : 0.0f; asm volatile("" : "+v"(v[u])); }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[1] = wall_clock64();

    // ---- q, raw k, v of this position: poll the granules (both waves: no hand-off before the scores)
    unsigned long long xg[3];
    {
        const unsigned long long* gp[3] = {tg.gran + h * HS + lane, tg.gran + tg.att_dim + kvh * HS + lane, tg.gran + tg.att_dim + tg.kv_dim + kvh * HS + lane};
        auto sweep = [&](unsigned long long (&x)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = __hip_atomic_load(gp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto fresh = [&](const unsigned long long (&x)[3]) __attribute__((always_inline)) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) ok = ok && (unsigned)(x[k] >> 32) == tg.tag;
            return __all(ok) != 0;
        };
        unsigned long long xa[3], xb[3];
        sweep(xa);
        for (unsigned spins = 0;; ++spins) {
            sweep(xb);
            if (fresh(xa)) { xg[0] = xa[0]; xg[1] = xa[1]; xg[2] = xa[2]; break; }
            sweep(xa);
            if (fresh(xb)) { xg[0] = xb[0]; xg[1] = xb[1]; xg[2] = xb[2]; break; }
            if (spins > kTagSpinMax || (spins & 1023) == 1023) {     // bounded: report and finish with garbage instead of hanging
                const int e = __hip_atomic_load(tg.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e != 0 || spins > kTagSpinMax) {
                    if (e == 0) __hip_atomic_store(tg.err, a.layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    xg[0] = xb[0]; xg[1] = xb[1]; xg[2] = xb[2];
                    break;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (loads return in order: the key tile's DMA landed before the granules did)
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[2] = wall_clock64();
    const float vnew = __uint_as_float((unsigned)xg[2]);
    qs[lane] = __uint_as_float((unsigned)xg[0]); kn[lane] = __uint_as_float((unsigned)xg[1]);
    // RoPE (transformer.rs:480-491): lane j < 32 owns pair (j, j + 32); wave 0 stores the rotated key into the cache, the wave that owns
    // position `pos` into the LDS tile
    if (lane < half) {
        const int j = lane;
        const float fcr = cs.x, fci = cs.y;
        {
            const float v0 = qs[j], v1 = qs[j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            qs[j] = a0 - a1; qs[j + half] = b0 + b1;
        }
        const float v0 = kn[j], v1 = kn[j + half];
        const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
        const float r0 = a0 - a1, r1 = b0 + b1;
        if (wv == 0) {
            kT[(((size_t)(j >> 2) * S + pos) << 2) + (j & 3)] = r0;
            kT[(((size_t)((j + half) >> 2) * S + pos) << 2) + ((j + half) & 3)] = r1;
        }
        if ((pos >> 6) == (two ? wv : 0)) {
            reinterpret_cast<float*>(kt)[(((j >> 2) * TW + pos) << 2) + (j & 3)] = r0;
            reinterpret_cast<float*>(kt)[((((j + half) >> 2) * TW + pos) << 2) + ((j + half) & 3)] = r1;
        }
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[3] = wall_clock64();

    // ---- scores (transformer.rs:507-529): one lane per key of this wave, sequential dot over the head dims
    int wpos = pos;
    if constexpr (GEMMA) { const int wb = a.st->win_base; wpos = wb >= 0 ? wb : pos; }
    const float sqrt_hs = sqrtf((float)HS), ninf = __uint_as_float(0xff800000u);
    const int t = tb + lane;
    float sc;
    {
        float score = 0.0f;
        constexpr int GB = 4;
        float4 ka[GB], qa[GB], kb[GB], qb[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) { ka[u] = kt[u * TW + t]; qa[u] = reinterpret_cast<const float4*>(qs)[u]; }
#pragma unroll
        for (int g0 = 0; g0 < HS4; g0 += 2 * GB) {
#pragma unroll
            for (int u = 0; u < GB; ++u) { kb[u] = kt[(g0 + GB + u) * TW + t]; qb[u] = reinterpret_cast<const float4*>(qs)[g0 + GB + u]; }
            asm volatile("" : "+v"(score) : : "memory");
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                float pr;
                pr = qa[u].x * ka[u].x; score = score + pr;
                pr = qa[u].y * ka[u].y; score = score + pr;
                pr = qa[u].z * ka[u].z; score = score + pr;
                pr = qa[u].w * ka[u].w; score = score + pr;
            }
            if (g0 + 2 * GB < HS4) {
#pragma unroll
                for (int u = 0; u < GB; ++u) { ka[u] = kt[(g0 + 2 * GB + u) * TW + t]; qa[u] = reinterpret_cast<const float4*>(qs)[g0 + 2 * GB + u]; }
            }
            asm volatile("" : "+v"(score) : : "memory");
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                float pr;
                pr = qb[u].x * kb[u].x; score = score + pr;
                pr = qb[u].y * kb[u].y; score = score + pr;
                pr = qb[u].z * kb[u].z; score = score + pr;
                pr = qb[u].w * kb[u].w; score = score + pr;
            }
        }
        score = score / sqrt_hs;
        if constexpr (GEMMA) {                                      // transformer.rs:518-526
            score = score / 50.0f;
            score = (float)tanh((double)score);
            score = score * 50.0f;
            score = score + (((unsigned)(wpos - t) <= 4096u) ? 0.0f : -2.3819763e38f);
        }
        sc = t < T ? score : ninf;
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[4] = wall_clock64();
    // ---- softmax (functional.rs:122-140): max (order-free), exp, sequential sum, divide
    float mx = wave64_max(sc);
    if (two) {
        if (lane == 0) red[wv] = mx;
        lds_barrier();
        mx = fmaxf(red[0], red[1]);
    }
    const float e0 = expf_glibc_t(t < T ? sc - mx : 0.0f, etab);
    const float ex = t < T ? e0 : 0.0f;                             // the lanes past T hold +0.0 (exact: the running sum is >= +0)
    float sum;
    if (!two) {
        sum = wave_serial_sum(0.0f, ex, (T + 15) >> 4);
    } else {
        if (wv == 0) { sum = wave_serial_sum(0.0f, ex, 4); if (lane == 0) red[2] = sum; }
        lds_barrier();
        if (wv == 1) { sum = wave_serial_sum(red[2], ex, (T - 64 + 15) >> 4); if (lane == 0) red[3] = sum; }
        lds_barrier();
        sum = red[3];
    }
    const float w = ex / sum;
    att[t] = t < pos ? w : 0.0f;                                    // the chains below cover the earlier positions; this one follows from registers
    const bool own_pos = (pos >> 6) == (two ? wv : 0);
    float a_pos = 0.0f;
    if (own_pos) a_pos = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), pos & 63));
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[5] = wall_clock64();
    // ---- weighted sum of values (transformer.rs:533-541), t ascending; lane = output dim
    if (two && wv == 1) {                                           // the products of this wave's keys, for wave 0 to add in order
        if (lane == 0) red[4] = a_pos;
#pragma unroll
        for (int u0 = 0; u0 < 64; u0 += 16) {
            if (64 + u0 < pos) {                                    // wave-uniform
                float4 w4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w4[u] = reinterpret_cast<const float4*>(att)[(64 + u0) / 4 + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    prod[(u0 + 4 * u + 0) * HS + lane] = w4[u].x * v[u0 + 4 * u + 0];
                    prod[(u0 + 4 * u + 1) * HS + lane] = w4[u].y * v[u0 + 4 * u + 1];
                    prod[(u0 + 4 * u + 2) * HS + lane] = w4[u].z * v[u0 + 4 * u + 2];
                    prod[(u0 + 4 * u + 3) * HS + lane] = w4[u].w * v[u0 + 4 * u + 3];
                }
            }
        }
        lds_barrier();
        return;
    }
    float o = 0.0f;
#pragma unroll
    for (int u0 = 0; u0 < 64; u0 += 16) {
        if (u0 < pos) {                                             // wave-uniform
            float4 w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = reinterpret_cast<const float4*>(att)[u0 / 4 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float pr;
                pr = w4[u].x * v[u0 + 4 * u + 0]; o = o + pr;
                pr = w4[u].y * v[u0 + 4 * u + 1]; o = o + pr;
                pr = w4[u].z * v[u0 + 4 * u + 2]; o = o + pr;
                pr = w4[u].w * v[u0 + 4 * u + 3]; o = o + pr;
            }
        }
    }
    if (two) {
        lds_barrier();
        a_pos = red[4];
#pragma unroll
        for (int u0 = 0; u0 < 64; u0 += 16) {
            if (64 + u0 < pos) {                                    // wave-uniform; slots past pos hold +-0.0 products (weight +0.0, value 0.0): exact no-ops
                float pb[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pb[u] = prod[(u0 + u) * HS + lane];
#pragma unroll
                for (int u = 0; u < 16; ++u) o = o + pb[u];
            }
        }
    }
    {
        const float pr = a_pos * vnew;
        o = o + pr;
        a.out[h * HS + lane] = o;
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[7] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// The wave form for 64-wide heads at positions 128 .. 255 (Llama-3.2-1B), with ONE WAVE PER 64 KEYS: wave w of the head's workgroup owns keys
// 64 w .. 64 w + 63 and runs only if the context reaches them.  (Below 128 positions attention_pair_tag above runs: this form, with its keys in
// registers instead of an LDS tile, measured 1.5 % slower per step there - profiles/r5_ab_attention_pair.txt - and 2 % faster than the workgroup
// form from 128 on.)  A wave keeps
// its keys AND its value rows in registers - lane l holds key 64 w + l (16 x 16-byte loads from the blocked K cache: no LDS tile) and, as an
// output lane, v_t[l] for its 64 rows (64 buffer loads per wave: the prefetch of a 100-position context in one wave landed 2.6 us after
// launch, later than the last qkv row) - and computes their scores and exponentials.  The reference's sequential chains cross the waves IN
// ORDER through LDS: the maximum (order-free), the softmax sum (wave 0's 64 terms, then wave 1 continues from its partial, ...), and the value
// chain o += a_t v_t: wave 0 adds its 64 terms while the other waves form the PRODUCTS a_t v_t of their keys (the multiplies are independent -
// only the adds are the chain) and park them in LDS; wave 0 then adds them in key order.  Every wave polls q / k / v and rotates q for itself
// (no hand-off before the scores); wave 0 alone writes the rotated key to the cache.  nw + 2 lds_barrier()s between the nw live waves, none
// when there is one.  Same adds in the same order as every other form: bit-equal.
// ------------------------------------------------------------------------------------------------
constexpr int kMultiWaves = 4;
constexpr size_t kMultiSmem = (size_t)(kMultiWaves * 2 * 64 + 64 * kMultiWaves + 16 + 32) * 4 + (size_t)(kMultiWaves - 1) * 64 * 64 * 4;   // q / k per wave, weights, hand-off words, products
template <bool GEMMA>
__device__ __forceinline__ void attention_multi_tag(const AttnArgs& a, const int h, const int pos, char* smem, const uint64_t etab, const AttTag& tg) {
    constexpr int HS = 64, HS4 = 16, half = 32;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nw = (pos >> 6) + 1 < kMultiWaves ? (pos >> 6) + 1 : kMultiWaves;      // live waves (uniform over the workgroup); the host keeps pos < 64 * kMultiWaves
    if (wv >= nw) return;
    const int tb = 64 * wv, t = tb + lane;                          // this wave's first key, this lane's key
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul, kv_dim = a.n_kv_heads * HS;
    const int T = pos + 1, S = a.seq_len;
    float* base = reinterpret_cast<float*>(smem);
    float* qs = base + wv * 2 * HS;                                 // this wave's rotated query
    float* kn = qs + HS;                                            // ... raw, then rotated key of this position
    float* att = base + kMultiWaves * 2 * HS;                       // 64 * kMultiWaves + 16 weights (shared)
    float* red = att + 64 * kMultiWaves + 16;                       // hand-off words: [0..3] maxima, [4] a_pos, [8..11] running sums
    float* prod = red + 32;                                         // [keys of waves 1..][HS]: a_t * v_t
    float* kT = att_k_head(a, kvh, HS);
    const float* vbase = a.v_cache + (size_t)a.layer * S * kv_dim + kvh * HS;
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[0] = wall_clock64();

    const float2 cs = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + (lane < half ? lane : 0)) * 2);
    // this lane's key: dims 4g .. 4g+3 at ((g * S + t) * 4) of the blocked cache - 16 bytes per lane, a wave reads 1 KiB per load (lanes past the
    // sequence clamp to its last row: their scores are replaced below; the new key's lane is patched after the RoPE step)
    f32x4v kreg[HS4];
    {
        const int tk = t < S ? t : S - 1;
        if (tb < pos) {                                             // wave-uniform
#pragma unroll
            for (int g = 0; g < HS4; ++g) kreg[g] = *reinterpret_cast<const f32x4v*>(kT + ((size_t)g * S + tk) * 4);
        } else {
#pragma unroll
            for (int g = 0; g < HS4; ++g) kreg[g] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    // this wave's value rows: v[u] = v_{tb + u}[lane], blocks of 16 rows nested so that the prefetch has one join (see attention_wave_tag)
    float v[64];
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vbase), 0, 0x7fffffff, 0x00020000);
    const int vrow = kv_dim * 4;
    auto vblock = [&](auto self, auto u0c) __attribute__((always_inline)) -> void {
        constexpr int u0 = decltype(u0c)::value;
        if constexpr (u0 < 64) {
            if (tb + u0 < pos) {                                    // wave-uniform
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int row = tb + u0 + u < S ? tb + u0 + u : S - 1;                       // (scalar: rows past the sequence re-read its last row, their weight is +0.0)
                    v[u0 + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, lane * 4, row * vrow, 0));
                }
                self(self, std::integral_constant<int, u0 + 16>());
            }
        }
    };
    vblock(vblock, std::integral_constant<int, 0>());
#pragma unroll
    for (int u = 0; u < 64; ++u) { v[u] = tb + u < pos ? v[u] : 0.0f; asm volatile("" : "+v"(v[u])); }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[1] = wall_clock64();

    // ---- q, raw k, v of this position: poll the granules (every wave: no hand-off before the scores)
    unsigned long long xg[3];
    {
        const unsigned long long* gp[3] = {tg.gran + h * HS + lane, tg.gran + tg.att_dim + kvh * HS + lane, tg.gran + tg.att_dim + tg.kv_dim + kvh * HS + lane};
        auto sweep = [&](unsigned long long (&x)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = __hip_atomic_load(gp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto fresh = [&](const unsigned long long (&x)[3]) __attribute__((always_inline)) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) ok = ok && (unsigned)(x[k] >> 32) == tg.tag;
            return __all(ok) != 0;
        };
        unsigned long long xa[3], xb[3];
        sweep(xa);
        for (unsigned spins = 0;; ++spins) {
            sweep(xb);
            if (fresh(xa)) { xg[0] = xa[0]; xg[1] = xa[1]; xg[2] = xa[2]; break; }
            sweep(xa);
            if (fresh(xb)) { xg[0] = xb[0]; xg[1] = xb[1]; xg[2] = xb[2]; break; }
            if (spins > kTagSpinMax || (spins & 1023) == 1023) {     // bounded: report and finish with garbage instead of hanging
                const int e = __hip_atomic_load(tg.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e != 0 || spins > kTagSpinMax) {
                    if (e == 0) __hip_atomic_store(tg.err, a.layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    xg[0] = xb[0]; xg[1] = xb[1]; xg[2] = xb[2];
                    break;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[2] = wall_clock64();
    const float vnew = __uint_as_float((unsigned)xg[2]);
    qs[lane] = __uint_as_float((unsigned)xg[0]); kn[lane] = __uint_as_float((unsigned)xg[1]);
    // RoPE (transformer.rs:480-491): lane j < 32 owns pair (j, j + 32); wave 0 stores the rotated key into the cache, the wave that owns
    // position `pos` patches it into that key's lane
    const bool own_pos = (pos >> 6) == wv;
    if (lane < half) {
        const int j = lane;
        const float fcr = cs.x, fci = cs.y;
        {
            const float v0 = qs[j], v1 = qs[j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            qs[j] = a0 - a1; qs[j + half] = b0 + b1;
        }
        const float v0 = kn[j], v1 = kn[j + half];
        const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
        const float r0 = a0 - a1, r1 = b0 + b1;
        kn[j] = r0; kn[j + half] = r1;
        if (wv == 0) {
            kT[(((size_t)(j >> 2) * S + pos) << 2) + (j & 3)] = r0;
            kT[(((size_t)((j + half) >> 2) * S + pos) << 2) + ((j + half) & 3)] = r1;
        }
    }
    if (own_pos) {                                                  // wave-uniform
#pragma unroll
        for (int g = 0; g < HS4; ++g) {
            const float4 n4 = reinterpret_cast<const float4*>(kn)[g];
            const bool me = lane == (pos & 63);
            kreg[g][0] = me ? n4.x : kreg[g][0]; kreg[g][1] = me ? n4.y : kreg[g][1]; kreg[g][2] = me ? n4.z : kreg[g][2]; kreg[g][3] = me ? n4.w : kreg[g][3];
        }
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[3] = wall_clock64();

    // ---- scores (transformer.rs:507-529): one lane per key of this wave, sequential dot over the head dims
    int wpos = pos;
    if constexpr (GEMMA) { const int wb = a.st->win_base; wpos = wb >= 0 ? wb : pos; }
    const float sqrt_hs = sqrtf((float)HS), ninf = __uint_as_float(0xff800000u);
    float sc;
    {
        float score = 0.0f;
#pragma unroll
        for (int g = 0; g < HS4; ++g) {
            const float4 q4 = reinterpret_cast<const float4*>(qs)[g];
            float pr;
            pr = q4.x * kreg[g][0]; score = score + pr;
            pr = q4.y * kreg[g][1]; score = score + pr;
            pr = q4.z * kreg[g][2]; score = score + pr;
            pr = q4.w * kreg[g][3]; score = score + pr;
        }
        score = score / sqrt_hs;
        if constexpr (GEMMA) {                                      // transformer.rs:518-526
            score = score / 50.0f;
            score = (float)tanh((double)score);
            score = score * 50.0f;
            score = score + (((unsigned)(wpos - t) <= 4096u) ? 0.0f : -2.3819763e38f);
        }
        sc = t < T ? score : ninf;
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[4] = wall_clock64();
    // ---- softmax (functional.rs:122-140): max (order-free), exp, sequential sum, divide
    float mx = wave64_max(sc);
    if (nw > 1) {
        if (lane == 0) red[wv] = mx;
        lds_barrier();
        mx = red[0];
        for (int w = 1; w < nw; ++w) mx = fmaxf(mx, red[w]);
    }
    const float e0 = expf_glibc_t(t < T ? sc - mx : 0.0f, etab);
    const float ex = t < T ? e0 : 0.0f;                             // the lanes past T hold +0.0 (exact: the running sum is >= +0)
    float sum;
    if (nw == 1) {
        sum = wave_serial_sum(0.0f, ex, (T + 15) >> 4);
    } else {
        for (int s = 0; s < nw; ++s) {                              // the chain, wave after wave
            if (wv == s) {
                const int left = T - 64 * s;
                const float part = wave_serial_sum(s ? red[8 + s - 1] : 0.0f, ex, left >= 64 ? 4 : (left + 15) >> 4);
                if (lane == 0) red[8 + s] = part;
            }
            lds_barrier();
        }
        sum = red[8 + nw - 1];
    }
    const float w = ex / sum;
    att[t] = t < pos ? w : 0.0f;                                    // the chains below cover the earlier positions; this one follows from registers
    float a_pos = 0.0f;
    if (own_pos) a_pos = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), pos & 63));
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[5] = wall_clock64();
    // ---- weighted sum of values (transformer.rs:533-541), t ascending; lane = output dim
    if (wv > 0) {                                                   // the products of this wave's keys, for wave 0 to add in order
        if (own_pos && lane == 0) red[4] = a_pos;
        float* mine = prod + (size_t)(wv - 1) * 64 * HS;
#pragma unroll
        for (int u0 = 0; u0 < 64; u0 += 16) {
            if (tb + u0 < pos) {                                    // wave-uniform
                float4 w4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w4[u] = reinterpret_cast<const float4*>(att)[(tb + u0) / 4 + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mine[(u0 + 4 * u + 0) * HS + lane] = w4[u].x * v[u0 + 4 * u + 0];
                    mine[(u0 + 4 * u + 1) * HS + lane] = w4[u].y * v[u0 + 4 * u + 1];
                    mine[(u0 + 4 * u + 2) * HS + lane] = w4[u].z * v[u0 + 4 * u + 2];
                    mine[(u0 + 4 * u + 3) * HS + lane] = w4[u].w * v[u0 + 4 * u + 3];
                }
            }
        }
        lds_barrier();
        return;
    }
    float o = 0.0f;
#pragma unroll
    for (int u0 = 0; u0 < 64; u0 += 16) {
        if (u0 < pos) {                                             // wave-uniform
            float4 w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = reinterpret_cast<const float4*>(att)[u0 / 4 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float pr;
                pr = w4[u].x * v[u0 + 4 * u + 0]; o = o + pr;
                pr = w4[u].y * v[u0 + 4 * u + 1]; o = o + pr;
                pr = w4[u].z * v[u0 + 4 * u + 2]; o = o + pr;
                pr = w4[u].w * v[u0 + 4 * u + 3]; o = o + pr;
            }
        }
    }
    if (nw > 1) {
        lds_barrier();
        a_pos = red[4];
        for (int k0 = 0; k0 < 64 * (kMultiWaves - 1); k0 += 16) {
            if (64 + k0 < pos) {                                    // wave-uniform; slots past pos hold +-0.0 products (weight +0.0, value 0.0): exact no-ops
                float pb[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pb[u] = prod[(k0 + u) * HS + lane];
#pragma unroll
                for (int u = 0; u < 16; ++u) o = o + pb[u];
            }
        }
    }
    {
        const float pr = a_pos * vnew;
        o = o + pr;
        a.out[h * HS + lane] = o;
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[7] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// The wave form for the WIDE heads (96 / 128: Phi-3.5, Llama-3.2-3B; 256: Gemma-2), round 6: FOUR waves per head.
// What the one-wave form (attention_wave_tag) lost its time to was not arithmetic: one wave issued the whole prefetch of its head - 32 LDS-DMA
// key loads + up to 128 value-row loads of 256 bytes at HS = 128 - which took 3 us to ISSUE, and the granule poll queued behind it: q / k / v
// were seen 3.7 us after the last qkv row (profiles/r6_timeline_*: polled at 7.4 us, rows done at 3.6).  Here the workgroup's four waves share
// the head by KEY RANGE: wave w owns keys kpw w .. kpw w + kpw - 1 with kpw = 32 for positions 64 .. 127, where the kernel uses this form (16 below 64
// positions, kept general) - their value rows in registers (ND * kpw loads per wave instead of ND * 64), their scores (one lane per key), their
// exponentials and their products a_t v_t - and a quarter of the dim groups of the K tile's LDS-DMA for ALL keys.  The reference's sequential
// chains stay whole: every wave runs the softmax sum over all keys itself (from the exponentials in LDS: identical bits, one barrier less than a
// hand-over), wave 0 runs the value chain - its own keys from registers, then the other waves' products from LDS (they alias the K tile, dead
// after the scores) in key order, then the new position's.  The new key never enters the tile: its lane reads it from the wave's own rotated
// copy, so no wave waits for another's DMA to patch it.  Four lds_barrier()s.  Same operations in the same order per value as
// attention_body: bit-identical.
// ------------------------------------------------------------------------------------------------
template <int HS> struct QuadGeom {
    static constexpr int HS4 = HS / 4, ND = (HS + 63) / 64, NH2 = (HS / 2 + 63) / 64;
    static constexpr int KPW = HS <= 128 ? 32 : 16, TWK = 4 * KPW;             // keys per wave at most, longest context
    static constexpr int NPASS = TWK / 64;
    static constexpr size_t SMEM = (size_t)(4 * 2 * HS + TWK + 16 + 32) * 4 + (size_t)HS4 * TWK * 16;
    static_assert((size_t)3 * KPW * ND * 64 * 4 <= (size_t)HS4 * TWK * 16, "the products alias the K tile");
};
template <int HS, bool GEMMA>
__device__ __forceinline__ void attention_quad_tag(const AttnArgs& a, const int h, const int pos, char* smem, const uint64_t etab, const AttTag& tg) {
    using Q = QuadGeom<HS>;
    constexpr int HS4 = Q::HS4, ND = Q::ND, NH2 = Q::NH2, KPW = Q::KPW, TWK = Q::TWK, half = HS / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int kpw = (KPW == 32 && pos >= 64) ? 32 : 16;             // keys per wave (uniform over the workgroup)
    const int tb = kpw * wv;                                        // this wave's first key
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul, kv_dim = a.n_kv_heads * HS;
    const int T = pos + 1, S = a.seq_len;
    float* base = reinterpret_cast<float*>(smem);
    float* qs = base + wv * 2 * HS;                                 // this wave's rotated query
    float* kn = qs + HS;                                            // ... raw, then rotated key of this position
    float* att = base + 4 * 2 * HS;                                 // TWK + 16 exponentials (shared)
    float* red = att + TWK + 16;                                    // hand-off words: [0..3] maxima, [4] a_pos
    float4* kt = reinterpret_cast<float4*>(red + 32);               // [HS4][TWK] x 16 bytes (shared)
    float* prod = reinterpret_cast<float*>(kt);                     // after the scores: [key - kpw][ND][64] products a_t * v_t of waves 1..3
    float* kT = att_k_head(a, kvh, HS);
    const float* vbase = a.v_cache + (size_t)a.layer * S * kv_dim + kvh * HS;
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[0] = wall_clock64();

    // ---- RoPE terms; this wave's quarter of the K tile (all keys) and its own value rows: all before the first poll
    float2 cs[NH2];
#pragma unroll
    for (int i = 0; i < NH2; ++i) { const int j = lane + 64 * i; cs[i] = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + (j < half ? j : 0)) * 2); }
#pragma unroll
    for (int p = 0; p < Q::NPASS; ++p) {
        if (64 * p < pos) {                                         // wave-uniform (rows past the sequence are read as they lie in the cache: seq_len >= TWK)
#pragma unroll
            for (int gg = 0; gg < HS4 / 4; ++gg) {
                const int g = wv * (HS4 / 4) + gg;
                __builtin_amdgcn_global_load_lds((const LMRS_GLOBAL void*)(kT + ((size_t)g * S + 64 * p + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(kt + g * TWK + 64 * p), 16, 0, 0);
            }
        }
    }
    float v[ND][KPW];
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vbase), 0, 0x7fffffff, 0x00020000);
    const int vrow = kv_dim * 4, vb0 = tb * vrow;
    auto vblock = [&](auto self, auto u0c) __attribute__((always_inline)) -> void {
        constexpr int u0 = decltype(u0c)::value;
        if constexpr (u0 < KPW) {
            if (u0 < kpw && tb + u0 < pos) {                        // wave-uniform
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < ND; ++i) {
                        const int d = lane + 64 * i;
                        v[i][u0 + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, (d < HS ? d : 0) * 4, vb0 + (u0 + u) * vrow, 0));
                    }
                self(self, std::integral_constant<int, u0 + 16>());
            }
        }
    };
    vblock(vblock, std::integral_constant<int, 0>());
#pragma unroll
    for (int u = 0; u < KPW; ++u)
#pragma unroll
        for (int i = 0; i < ND; ++i) { v[i][u] = (u < kpw && tb + u < pos) ? v[i][u] : 0.0f; asm volatile("" : "+v"(v[i][u])); }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[1] = wall_clock64();

    // ---- q, raw k, v of this position: poll the granules (every wave: nothing is handed over before the scores)
    unsigned long long xg[3 * ND];
    {
        const unsigned long long* gp[3 * ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = lane + 64 * i, dc = d < HS ? d : 0;
            gp[3 * i] = tg.gran + h * HS + dc; gp[3 * i + 1] = tg.gran + tg.att_dim + kvh * HS + dc; gp[3 * i + 2] = tg.gran + tg.att_dim + tg.kv_dim + kvh * HS + dc;
        }
        auto sweep = [&](unsigned long long (&x)[3 * ND]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 3 * ND; ++k) x[k] = __hip_atomic_load(gp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto fresh = [&](const unsigned long long (&x)[3 * ND]) __attribute__((always_inline)) {
            unsigned bad = 0u;
#pragma unroll
            for (int k = 0; k < 3 * ND; ++k) bad |= (unsigned)(x[k] >> 32) ^ tg.tag;
            return __all(bad == 0u) != 0;
        };
        unsigned long long xa[3 * ND], xb[3 * ND];
        sweep(xa);
        for (unsigned spins = 0;; ++spins) {
            sweep(xb);
            if (fresh(xa)) {
#pragma unroll
                for (int k = 0; k < 3 * ND; ++k) xg[k] = xa[k];
                break;
            }
            sweep(xa);
            if (fresh(xb)) {
#pragma unroll
                for (int k = 0; k < 3 * ND; ++k) xg[k] = xb[k];
                break;
            }
            if (spins > kTagSpinMax || (spins & 1023) == 1023) {     // bounded: report and finish with garbage instead of hanging
                const int e = __hip_atomic_load(tg.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e != 0 || spins > kTagSpinMax) {
                    if (e == 0) __hip_atomic_store(tg.err, a.layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < 3 * ND; ++k) xg[k] = xb[k];
                    break;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (loads return in order: this wave's share of the key tile landed before the granules did)
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[2] = wall_clock64();
    float vnew[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = lane + 64 * i;
        vnew[i] = __uint_as_float((unsigned)xg[3 * i + 2]);
        if (d < HS) { qs[d] = __uint_as_float((unsigned)xg[3 * i]); kn[d] = __uint_as_float((unsigned)xg[3 * i + 1]); }
    }
    // RoPE (transformer.rs:480-491): this lane owns both halves of pair j; every wave keeps its own rotated q and key, wave 0 writes the
    // key to the cache (for the later steps)
#pragma unroll
    for (int i = 0; i < NH2; ++i) {
        const int j = lane + 64 * i;
        if (j < half) {
            const float fcr = cs[i].x, fci = cs[i].y;
            {
                const float v0 = qs[j], v1 = qs[j + half];
                const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
                qs[j] = a0 - a1; qs[j + half] = b0 + b1;
            }
            const float v0 = kn[j], v1 = kn[j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            const float r0 = a0 - a1, r1 = b0 + b1;
            kn[j] = r0; kn[j + half] = r1;
            if (wv == 0) {
                kT[(((size_t)(j >> 2) * S + pos) << 2) + (j & 3)] = r0;
                kT[(((size_t)((j + half) >> 2) * S + pos) << 2) + ((j + half) & 3)] = r1;
            }
        }
    }
    lds_barrier();                                                  // every wave's share of the K tile has landed
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[3] = wall_clock64();

    // ---- scores (transformer.rs:507-529): one lane per key of this wave, sequential dot over the head dims; the key of this position
    // comes from the wave's own rotated copy (kn), every other key from the tile
    int wpos = pos;
    if constexpr (GEMMA) { const int wb = a.st->win_base; wpos = wb >= 0 ? wb : pos; }
    const float sqrt_hs = sqrtf((float)HS), ninf = __uint_as_float(0xff800000u);
    const bool mine = lane < kpw;
    const int t = tb + (mine ? lane : 0);
    float sc;
    {
        const float4* kb4 = (t == pos) ? reinterpret_cast<const float4*>(kn) : kt + t;
        const int ks = (t == pos) ? 1 : TWK;
        float score = 0.0f;
        constexpr int GB = 4;
        static_assert(HS4 % (2 * GB) == 0, "head size");
        float4 ka[GB], qa[GB], kb[GB], qb[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) { ka[u] = kb4[u * ks]; qa[u] = reinterpret_cast<const float4*>(qs)[u]; }
#pragma unroll
        for (int g0 = 0; g0 < HS4; g0 += 2 * GB) {
#pragma unroll
            for (int u = 0; u < GB; ++u) { kb[u] = kb4[(g0 + GB + u) * ks]; qb[u] = reinterpret_cast<const float4*>(qs)[g0 + GB + u]; }
            asm volatile("" : "+v"(score) : : "memory");
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                float pr;
                pr = qa[u].x * ka[u].x; score = score + pr;
                pr = qa[u].y * ka[u].y; score = score + pr;
                pr = qa[u].z * ka[u].z; score = score + pr;
                pr = qa[u].w * ka[u].w; score = score + pr;
            }
            if (g0 + 2 * GB < HS4) {
#pragma unroll
                for (int u = 0; u < GB; ++u) { ka[u] = kb4[(g0 + 2 * GB + u) * ks]; qa[u] = reinterpret_cast<const float4*>(qs)[g0 + 2 * GB + u]; }
            }
            asm volatile("" : "+v"(score) : : "memory");
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                float pr;
                pr = qb[u].x * kb[u].x; score = score + pr;
                pr = qb[u].y * kb[u].y; score = score + pr;
                pr = qb[u].z * kb[u].z; score = score + pr;
                pr = qb[u].w * kb[u].w; score = score + pr;
            }
        }
        score = score / sqrt_hs;
        if constexpr (GEMMA) {                                      // transformer.rs:518-526
            score = score / 50.0f;
            score = (float)tanh((double)score);
            score = score * 50.0f;
            score = score + (((unsigned)(wpos - t) <= 4096u) ? 0.0f : -2.3819763e38f);
        }
        sc = (mine && t < T) ? score : ninf;
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[4] = wall_clock64();
    // ---- softmax (functional.rs:122-140): max (order-free), exp, sequential sum over ALL keys (run by every wave), divide
    float mx = wave64_max(sc);
    if (lane == 0) red[wv] = mx;
    lds_barrier();                                                  // (also: every wave is done with the K tile)
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float e0 = expf_glibc_t((mine && t < T) ? sc - mx : 0.0f, etab);
    const float ex = (mine && t < T) ? e0 : 0.0f;                   // the slots past T hold +0.0 (exact: the running sum is >= +0)
    if (mine) att[t] = ex;
    lds_barrier();
    float sum;
    {
        const float ea = att[lane];                                 // keys 0 .. 63 (kpw = 16: all four waves' slots; the slots past 4 kpw are never added)
        sum = wave_serial_sum(0.0f, ea, T >= 64 ? 4 : (T + 15) >> 4);
        if constexpr (TWK > 64) {
            if (T > 64) { const float eb = att[64 + lane]; sum = wave_serial_sum(sum, eb, (T - 64 + 15) >> 4); }
        }
    }
    const float w = ex / sum;
    const float wz = (mine && t < pos) ? w : 0.0f;                  // the chains below cover the earlier positions; this one follows from registers
    const bool own_pos = pos >= tb && pos < tb + kpw;
    if (own_pos) {                                                  // wave-uniform
        const float ap = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), pos - tb));
        if (lane == 0) red[4] = ap;
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[5] = wall_clock64();
    // ---- weighted sum of values (transformer.rs:533-541), t ascending; this lane's dims
    if (wv > 0) {                                                   // the products of this wave's keys, for wave 0 to add in order
        float* dst = prod + (size_t)(tb - kpw) * ND * 64;
#pragma unroll
        for (int u0 = 0; u0 < KPW; u0 += 16) {
            if (u0 < kpw && tb + u0 < pos) {                        // wave-uniform
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float wt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wz), u0 + u));
#pragma unroll
                    for (int i = 0; i < ND; ++i) dst[((u0 + u) * ND + i) * 64 + lane] = wt * v[i][u0 + u];
                }
            }
        }
        lds_barrier();
        return;
    }
    float o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = 0.0f;
#pragma unroll
    for (int u0 = 0; u0 < KPW; u0 += 16) {
        if (u0 < kpw && u0 < pos) {                                 // wave-uniform
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float wt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wz), u0 + u));
#pragma unroll
                for (int i = 0; i < ND; ++i) { const float pr = wt * v[i][u0 + u]; o[i] = o[i] + pr; }
            }
        }
    }
    lds_barrier();
    const float a_pos = red[4];
    for (int k0 = 0; k0 < 3 * kpw; k0 += 8) {
        if (kpw + k0 < pos) {                                       // wave-uniform; slots past pos hold +-0.0 products (weight +0.0, value 0.0): exact no-ops
            float pb[8][ND];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < ND; ++i) pb[u][i] = prod[((k0 + u) * ND + i) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < ND; ++i) o[i] = o[i] + pb[u][i];
        }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = lane + 64 * i;
        const float pr = a_pos * vnew[i];
        o[i] = o[i] + pr;
        if (d < HS) a.out[h * HS + d] = o[i];
    }
    if (a.dbg && lane == 0 && wv == 0 && blockIdx.x == 0) a.dbg[7] = wall_clock64();
}

constexpr int qa_chunk(int hs) { return (8192 / hs) & ~31; }        // V rows per LDS tile: 64 -> 128, 96 -> 64, 128 -> 64, 256 -> 32
template <int HS> struct QaGeom { static constexpr int CH = qa_chunk(HS), NF = (CH * (HS / 4) + kBlock - 1) / kBlock; };

template <int N, int L, int PRO, bool Q4, int HS, bool GEMMA, bool WAVE>
__global__ __launch_bounds__(kBlock) void qkv_attn_kernel(LMRS_HOT_PARAMS, const QkvAttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    QkvAttnArgs a = a0;                                              // (kernel-argument preload: see gemv_static_kernel)
    a.g = with_hot(a0.g, h_xin, h_wq, h_ws, h_rms_w, h_st, h_seq, h_out); a.t.st = h_st;
    const int nh = a.t.n_heads;
    if ((int)blockIdx.x < nh) {
        // (wave forms: the workgroup's waves share the head by key range - attention_pair_tag / attention_multi_tag for 64-wide heads, attention_quad_tag beyond)
        const uint64_t etab = exp2f_tab_lane();
        const int pos = a.t.st->pos;
        const AttTag tg{a.g.gran, *a.g.seq + 1u, a.g.att_dim, a.g.kv_dim, a.err};
        // workgroup -> head: the query heads that share a kv head sit n_kv_heads workgroups apart - workgroup b runs on XCD b % 8, so with 8 (or 32)
        // kv heads they share an XCD and its L2 fetches their K / V history once, not once per query head (the launch's counter traffic was 1.25 x its
        // algorithmic bytes with head = b: profiles/r4b_traffic_llama1b_q8.json against r5_traffic_llama1b_q8.json)
        const int nkv = a.t.n_kv_heads, bq = (int)blockIdx.x / nkv, br = (int)blockIdx.x - bq * nkv;
        const int head = br * (nh / nkv) + bq;
        if constexpr (WAVE && HS == 64) { if (pos < 128) attention_pair_tag<GEMMA>(a.t, head, pos, smem, etab, tg); else attention_multi_tag<GEMMA>(a.t, head, pos, smem, etab, tg); }
        else if constexpr (WAVE) {
            // 96 / 128-wide heads: ONE wave per head below 64 positions (no barrier at all: 950 against 981 us per step on Llama-3.2-3B at the driver's
            // 20 steps), the workgroup's four waves by key range from 64 to 127 (against the workgroup form there: -3 % per step) - profiles/r6_ab_attention_quad.txt
            if (pos < 64) { if (threadIdx.x >= 64) return; attention_wave_tag<HS, GEMMA>(a.t, head, pos, smem, etab, tg); }
            else attention_quad_tag<HS, GEMMA>(a.t, head, pos, smem, etab, tg);
        }
        else attention_body<HS, QaGeom<HS>::NF, false, false, GEMMA, false, true>(a.t, head, pos, smem, etab, AttPre(), tg);
    } else {
        gemv_static_body<N, L, PRO, EPI_QKV_TAG, kBlock, Q4>(a.g, smem, (int)blockIdx.x - nh, (int)gridDim.x - nh);
    }
}

// the merged classes: (N, L, PRO, Q4) of the model's qkv launch x (head size, Gemma score path)
#define LMRS_QA_TABLE(X)                                                                                                \
    X(2048, 32, PRO_RMS_QUANT, false, 64, false)      /* Llama-3.2-1B Q8_0 */                                            \
    X(3072, 32, PRO_RMS_QUANT, false, 128, false)     /* Llama-3.2-3B Q8_0 */                                            \
    X(3072, 16, PRO_RMS_QUANT, false, 96, false) X(3072, 32, PRO_RMS_QUANT, false, 96, false)     /* Phi-3.5 Q8_0 */          \
    X(2304, 16, PRO_RMS_QUANT, false, 256, true) X(2304, 16, PRO_ADD_RMS_QUANT, false, 256, true)   /* Gemma-2-2B Q8_0 */  \
    X(2304, 8, PRO_RMS_QUANT, true, 256, true) X(2304, 8, PRO_ADD_RMS_QUANT, true, 256, true)       /* Gemma-2-2B Q4_0 */  \
    X(2048, 16, PRO_RMS_QUANT, true, 64, false)       /* Llama-3.2-1B Q4_0 */

bool qkv_attn_supported(const GemvArgs& g, int pro, const AttnArgs& t) {
    const StaticClass sc = static_class(g, pro, EPI_QKV);
    if (!sc.L || sc.nt != kBlock || t.n_heads <= 0) return false;
#define X(n_, l_, p_, q_, hs_, gm_) if (g.n == n_ && sc.L == l_ && pro == p_ && (g.q4 != 0) == q_ && t.head_size == hs_ && (t.gemma != 0) == gm_) return true;
    LMRS_QA_TABLE(X)
#undef X
    return false;
}
int qkv_attn_wave_T(int head_size) { return qa_wave_T(head_size); }

// The merged launch in ONE round of workgroups.  The GEMV part takes any grid <= its row passes (a workgroup walks its passes with a
// double-buffered tile and runs the prologue once), but a workgroup that only STARTS when another one has left pays the whole prologue - 3 us
// of norm chain and quantiser at dim 3072 - behind the first round: Llama-3.2-3B's 640 + 24 and Phi-3.5's 576 + 32 workgroups on 512 resident
// slots (two per CU: registers) finished their rows at 6.8 us instead of 3.7, and the heads polled until then (profiles/r6_timeline_*).  The grid is
// therefore capped at what is resident at once (occupancy query, cached per kernel and LDS size); the surplus passes go to workgroups as second passes.
static int resident_grid_cap(const void* fn, size_t smem) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({fn, smem});
    if (it != cache.end()) return it->second;
    int dev = 0, cus = 0, occ = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kBlock, smem) != hipSuccess || occ <= 0) { (void)hipGetLastError(); occ = 0; }
    const int cap = occ * cus;                                      // 0: unknown - no cap
    cache[{fn, smem}] = cap;
    return cap;
}
static int qkv_attn_grid(const void* fn, size_t smem, int grid, int n_heads) {
    static const int on = env_flag("LMRS_QKV_ONE_ROUND", 1);
    const int cap = on ? resident_grid_cap(fn, smem) : 0;
    return (cap > n_heads + 64 && grid > cap) ? cap : grid;
}

template <int N, int L, int PRO, bool Q4, int HS, bool GEMMA>
static hipError_t launch_qkv_attn_class(const QkvAttnArgs& a, int grid, size_t gsmem, int max_T, bool wave, hipStream_t s) {
    if constexpr (qa_wave_T(HS) > 0) {
        if (wave) {
            constexpr size_t wsm = HS == 64 ? (kMultiSmem > kPairSmem ? kMultiSmem : kPairSmem) : (QuadGeom<HS>::SMEM > WaveGeom<HS>::SMEM ? QuadGeom<HS>::SMEM : WaveGeom<HS>::SMEM);
            size_t smem = wsm > gsmem ? wsm : gsmem;
            if (smem > 64 * 1024) allow_big_lds(reinterpret_cast<const void*>(qkv_attn_kernel<N, L, PRO, Q4, HS, GEMMA, true>));
            grid = qkv_attn_grid(reinterpret_cast<const void*>(qkv_attn_kernel<N, L, PRO, Q4, HS, GEMMA, true>), smem, grid, a.t.n_heads);
            LMRS_LAUNCH_GRID((qkv_attn_kernel<N, L, PRO, Q4, HS, GEMMA, true>), dim3(grid), kBlock, smem, s, LMRS_HOT_OF(a.g), a);
            return hipGetLastError();
        }
    }
    if (wave) return hipErrorNotSupported;
    size_t smem = attention_smem(HS, qa_chunk(HS), max_T);
    if (gsmem > smem) smem = gsmem;
    if (smem > 64 * 1024) allow_big_lds(reinterpret_cast<const void*>(qkv_attn_kernel<N, L, PRO, Q4, HS, GEMMA, false>));
    grid = qkv_attn_grid(reinterpret_cast<const void*>(qkv_attn_kernel<N, L, PRO, Q4, HS, GEMMA, false>), smem, grid, a.t.n_heads);
    LMRS_LAUNCH_GRID((qkv_attn_kernel<N, L, PRO, Q4, HS, GEMMA, false>), dim3(grid), kBlock, smem, s, LMRS_HOT_OF(a.g), a);
    return hipGetLastError();
}

// wave: one wave per head (contexts up to qkv_attn_wave_T(head_size)); else one workgroup per head (contexts up to max_T)
hipError_t launch_qkv_attn(const GemvArgs& g0, int pro, const AttnArgs& t0, int* err, int max_T, bool wave, hipStream_t s) {
    static const int order_barrier = env_flag("LMRS_ORDER_BARRIER", 1);
    if (!qkv_attn_supported(g0, pro, t0) || !g0.gran || !g0.seq || !err) return hipErrorNotSupported;
    QkvAttnArgs a{g0, t0, err};
    a.g.order_barrier = order_barrier; a.g.chain_spread = env_flag("LMRS_CHAIN_SPREAD", 1);
    a.t.chunk = qa_chunk(t0.head_size);
    const StaticClass sc = static_class(a.g, pro, EPI_QKV);
    const int grid = t0.n_heads + gemv_grid(a.g, pro, EPI_QKV);
    const size_t gsmem = gemv_smem(a.g, pro);
#define X(n_, l_, p_, q_, hs_, gm_)                                                                                      \
    if (a.g.n == n_ && sc.L == l_ && pro == p_ && (a.g.q4 != 0) == q_ && t0.head_size == hs_ && (t0.gemma != 0) == gm_)  \
        return launch_qkv_attn_class<n_, l_, p_, q_, hs_, gm_>(a, grid, gsmem, max_T, wave, s);
    LMRS_QA_TABLE(X)
#undef X
    return hipErrorNotSupported;
}

// ------------------------------------------------------------------------------------------------
// Attention of one new token at LONG contexts, as two launches.  One workgroup per head streams the whole K and V of its
// kv head through one CU (≈24 GB/s per CU): 89 us per layer at 4000 positions, of which only the softmax sum and the V
// accumulation - one chain of T adds each - are inherently serial.  Here the scores are cut by (head, 256-key chunk), all
// independent, into S[head][t]; then one workgroup per (head, quarter of the head dims) redoes the cheap softmax of its head
// (max / exp / the sum chain / divide - the four copies run concurrently) and runs the V chains of its HS/4 dims, streaming
// a quarter of the V bytes.  Same operations in the same order per value as attention_body: bit-identical.
// ------------------------------------------------------------------------------------------------
template <int HS, bool GEMMA>
__global__ __launch_bounds__(kBlock) void attention_split_scores_kernel(const AttnArgs a, float* S) {
    __shared__ __attribute__((aligned(16))) float q[HS];
    __shared__ __attribute__((aligned(16))) float kn[HS];
    constexpr int half = HS / 2;
    const int h = blockIdx.x, tid = threadIdx.x;
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul;
    const int pos = a.st->pos, T = pos + 1;
    const int t0 = blockIdx.y * kBlock;
    if (t0 >= T) return;
    const int SQ = a.seq_len;
    float* kT = att_k_head(a, kvh, HS);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(kT, 0, HS * SQ * 4, 0x00020000);
    constexpr int KG = HS / 4 <= 32 ? HS / 4 : 16;
    const int t = t0 + tid, tc = t < T ? t : T - 1;
    f32x4v kk[KG];
    att_kload<KG>(kk, krs, tc * 16, 0, SQ);
    const bool mine = pos >= t0 && pos < t0 + kBlock;       // this chunk holds the new key: rotate it, store it (every head of the kv head: same values)
    for (int j = tid; j < half; j += kBlock) {              // RoPE (transformer.rs:480-491), as in attention_body
        const float2 cs = *reinterpret_cast<const float2*>(a.rope + ((size_t)pos * half + j) * 2);
        const float fcr = cs.x, fci = cs.y;
        {
            const float v0 = a.q[h * HS + j], v1 = a.q[h * HS + j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            q[j] = a0 - a1; q[j + half] = b0 + b1;
        }
        if (mine) {
            const float v0 = a.k_raw[kvh * HS + j], v1 = a.k_raw[kvh * HS + j + half];
            const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
            const float r0 = a0 - a1, r1 = b0 + b1;
            kn[j] = r0; kn[j + half] = r1;
            kT[(((size_t)(j >> 2) * SQ + pos) << 2) + (j & 3)] = r0;
            kT[(((size_t)((j + half) >> 2) * SQ + pos) << 2) + ((j + half) & 3)] = r1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (tc == pos) {                                        // batch 0 was loaded before the new key existed: patch it from LDS
#pragma unroll
        for (int u = 0; u < KG; ++u) { const float4 t4 = reinterpret_cast<const float4*>(kn)[u]; kk[u] = f32x4v{t4.x, t4.y, t4.z, t4.w}; }
    }
    int wpos = pos;
    if constexpr (GEMMA) { const int wb = a.st->win_base; wpos = wb >= 0 ? wb : pos; }
    const float sqrt_hs = sqrtf((float)HS);
    float score = att_score_chain<HS, KG>(kk, krs, tc * 16, q, SQ);
    score = score / sqrt_hs;
    if constexpr (GEMMA) {                                  // transformer.rs:518-526
        score = score / 50.0f;
        score = (float)tanh((double)score);
        score = score * 50.0f;
        score = score + (((unsigned)(wpos - t) <= 4096u) ? 0.0f : -2.3819763e38f);
    }
    if (t < T) S[(size_t)h * SQ + t] = score;
}

// Round 6: the value phase as a PIPELINE with specialised waves.  Waves 1-3 load a chunk of 256 value rows (the workgroup's slice of the head
// dims), scale them by the weights and store the products TRANSPOSED - tile[d][t] - while wave 0, lane d, walks down row d of the PREVIOUS
// chunk's tile with 16-byte reads (4 terms per read, the next batch of 16 in flight under the adds of the current one); one barrier per chunk,
// two tiles.  Before: rows stored [t][d], the chain lane read one 4-byte term at a time, and the loads / stores / chains of a chunk ran one
// after the other - 16.0 us per layer at 1024-1087 positions (profiles/r5_long_decode_kernel_stats.csv), 14.3 now.  The softmax sum runs in
// registers (wave_serial_sum: 64 exponentials in the 64 lanes, one DPP-fed add per term; the next 64 are read under the adds).
// What bounds it (profiles/r6_long_decode.txt, stamps at 1056 positions): the two chains, 2 x 1056 dependent adds, at 6.9 cycles per term
// DPP-fed and 8-10 LDS-fed (a ds_read_b128 costs the issuing wave ~14 cycles per 4 terms on top of the adds) = 3.4 + 5.3 of the
// workgroup's 12 us.  Measured and not kept: every value chain DPP-fed too, one output dim per wave - 4 waves x 512 workgroups: the
// workgroup 10.6 us but the step +12 us (two chain waves per SIMD); 16 waves x 128 workgroups: value phase 9.4 us (four chain waves per SIMD
// share its issue port).  Same operations in the same order per value: bit-identical.
// One lane's chain over a row of float4 terms in LDS, 16 terms per batch, THREE batches of reads in flight under the adds of the current one (a ds_read_b128
// returns after ~130-200 cycles, 16 dependent adds take ~64-80: one batch ahead left the chain waiting on every batch - 671 -> 664 us per step at 1032..1063
// positions, 808 -> 800 at 1800; profiles/r6_ab_long_decode.txt, which also holds what did NOT help: the softmax sum as a one-lane LDS-fed chain, +4..+25 us
// against the DPP-fed wave_serial_sum, and the chain lanes alone under EXEC, +-0).  nb batches are added; reads run up to three batches past nb.
__device__ __forceinline__ float chain_rows16(float o, const float4* row, int nb) {
    float4 R[4][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) R[k][u] = row[k * 4 + u];
    for (int b = 0; b < nb; b += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (b + k >= nb) break;
#pragma unroll
            for (int u = 0; u < 4; ++u) R[(k + 3) & 3][u] = row[(b + k + 3) * 4 + u];
            asm volatile("" : "+v"(o) : : "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) { o = o + R[k][u].x; o = o + R[k][u].y; o = o + R[k][u].z; o = o + R[k][u].w; }
        }
    }
    return o;
}
template <int HS> struct SplitGeom {
    static constexpr int NSL = 4, HP = HS / NSL, HP4 = HP / 4, CHK = HS <= 128 ? 256 : 128, PITCH = CHK + 4;   // workgroups per head, dims per workgroup, keys per chunk (two tiles + 8192 weights within 160 KB), floats per tile row
    static constexpr int NLD = 3 * 64, NSLOT = (CHK * HP4 + NLD - 1) / NLD;                                 // loader threads (waves 1-3), float4 slots per loader thread and chunk
    static_assert(HP % 4 == 0 && HP <= 64, "dim slice");
};
template <int HS, bool GEMMA>
__global__ __launch_bounds__(kBlock) void attention_split_values_kernel(const AttnArgs a, const float* S) {
    using G = SplitGeom<HS>;
    constexpr int HP = G::HP, HP4 = G::HP4, CHK = G::CHK, PITCH = G::PITCH, NLD = G::NLD, NSLOT = G::NSLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul, kv_dim = a.n_kv_heads * HS;
    const int pos = a.st->pos, T = pos + 1;
    const uint64_t etab = exp2f_tab_lane();
    const bool stamp = a.dbg && tid == 0 && h == 0 && sl == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    float* red = reinterpret_cast<float*>(smem);            // 16 floats of reduction scratch
    float* tile = red + 16;                                 // 2 tiles of HP rows x PITCH floats: products a_t * v_t[d], [d][t - t0]
    float* att = tile + 2 * HP * PITCH;                     // T exponentials, then weights (+ 128 floats of zero padding)
    const float* vbase = a.v_cache + (size_t)a.layer * a.seq_len * kv_dim + kvh * HS + sl * HP;
    const int nchunks = (T + CHK - 1) / CHK;
    // loader threads: slot i of loader thread lt is float4 number f = lt + i * NLD of the chunk (row f / HP4, dims 4 (f % HP4) ..)
    const int lt = tid - 64;
    float4 vreg[NSLOT];
    auto vload = [&](int c) __attribute__((always_inline)) {
        const int t0 = c * CHK, ct = (T - t0) < CHK ? (T - t0) : CHK;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int f = lt + i * NLD, row = f / HP4, c4 = f - row * HP4;
            const int rr = row < ct ? row : ct - 1;                                     // (rows past the chunk: a harmless duplicate load, their products are zeroed)
            vreg[i] = ld_f32x4<false>(vbase + (size_t)(t0 + rr) * kv_dim + c4 * 4);
        }
    };
    auto vstore = [&](int c) __attribute__((always_inline)) {
        const int t0 = c * CHK, ct = (T - t0) < CHK ? (T - t0) : CHK, nrows = (ct + 15) & ~15;       // the chain runs in batches of 16: rows ct .. nrows-1 hold +0.0
        float* tl = tile + (c & 1) * HP * PITCH;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int f = lt + i * NLD, row = f / HP4, c4 = f - row * HP4;
            if (row < nrows) {
                const bool live = row < ct;
                const float w = live ? att[t0 + row] : 0.0f;
                float4 v = vreg[i];
                v.x = live ? w * v.x : 0.0f; v.y = live ? w * v.y : 0.0f; v.z = live ? w * v.z : 0.0f; v.w = live ? w * v.w : 0.0f;
                tl[(c4 * 4 + 0) * PITCH + row] = v.x; tl[(c4 * 4 + 1) * PITCH + row] = v.y;
                tl[(c4 * 4 + 2) * PITCH + row] = v.z; tl[(c4 * 4 + 3) * PITCH + row] = v.w;
            }
        }
    };
    if (wave > 0) vload(0);                                 // the first V chunk is in flight across the softmax
    // softmax (functional.rs:122-140): max (order-free), exp, sequential sum, divide - as in attention_body
    float lmax = __uint_as_float(0xff800000u);
    for (int t = tid; t < T; t += kBlock) { const float sc = S[(size_t)h * a.seq_len + t]; att[t] = sc; lmax = fmaxf(lmax, sc); }
    lmax = wave64_max(lmax);
    if (lane == 0) red[wave] = lmax;
    lds_barrier();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (stamp) a.dbg[1] = wall_clock64();
    for (int t0 = 0; t0 < T; t0 += kBlock) {                // whole waves call expf together (it shuffles)
        const int t = t0 + tid;
        const float e = expf_glibc_t(t < T ? att[t] - mx : 0.0f, etab);
        if (t < T) att[t] = e;
    }
    if (tid < 128) att[T + tid] = 0.0f;                     // +0.0 past the sequence: exact for a running sum that is >= +0 (128: the sum chain reads one block ahead)
    lds_barrier();
    if (stamp) a.dbg[2] = wall_clock64();
    if (wave == 0) {                                        // the sum chain, 64 keys per step in the 64 lanes; the next 64 are read under the adds
        float sum = 0.0f;
        float e = att[lane];
        for (int t0 = 0; t0 < T; t0 += 64) {
            const float en = att[t0 + 64 + lane];
            const int left = T - t0;
            sum = wave_serial_sum(sum, e, left >= 64 ? 4 : (left + 15) >> 4);
            e = en;
        }
        if (lane == 0) red[4] = sum;
    }
    lds_barrier();
    const float sum = red[4];
    if (stamp) a.dbg[4] = wall_clock64();
    for (int t = tid; t < T; t += kBlock) att[t] = att[t] / sum;
    lds_barrier();
    if (stamp) a.dbg[5] = wall_clock64();
    // weighted sum of values (transformer.rs:533-541) for this workgroup's HP dims: wave 0 chains chunk c while waves 1-3 prepare chunk c + 1
    if (wave > 0) { vstore(0); if (nchunks > 1) vload(1); }
    lds_barrier();
    if (stamp) a.dbg[6] = wall_clock64();
    float o = 0.0f;
    for (int c = 0; c < nchunks; ++c) {
        if (wave == 0) {
          {
            const int t0 = c * CHK, ct = (T - t0) < CHK ? (T - t0) : CHK, nb = (ct + 15) >> 4;
            const float4* row = reinterpret_cast<const float4*>(tile + (c & 1) * HP * PITCH + (lane < HP ? lane : 0) * PITCH);
            o = chain_rows16(o, row, nb);                   // (reads up to three batches past nb: inside the shared memory of the kernel, never added)
          }
        } else if (c + 1 < nchunks) {
            vstore(c + 1);
            if (c + 2 < nchunks) vload(c + 2);
        }
        lds_barrier();
    }
    if (wave == 0 && lane < HP) a.out[h * HS + sl * HP + lane] = o;
    if (stamp) a.dbg[7] = wall_clock64();
}

size_t attention_split_scratch_floats(int n_heads, int seq_len) { return (size_t)n_heads * seq_len; }

template <int HS, bool GEMMA>
static hipError_t launch_attention_split_hsg(const AttnArgs& a0, float* S, int n_key_chunks, hipStream_t s) {
    AttnArgs a = a0;
    using G = SplitGeom<HS>;
    a.chunk = G::CHK;
    const size_t smem = (size_t)(16 + 2 * G::HP * G::PITCH + ((a.seq_len + 3) & ~3) + 128) * 4;
    allow_big_lds(reinterpret_cast<const void*>(attention_split_values_kernel<HS, GEMMA>));
    LMRS_LAUNCH_GRID((attention_split_scores_kernel<HS, GEMMA>), dim3(a.n_heads, n_key_chunks), kBlock, 0, s, a, S);
    LMRS_LAUNCH_GRID((attention_split_values_kernel<HS, GEMMA>), dim3(a.n_heads, G::NSL), kBlock, smem, s, a, (const float*)S);
    return hipGetLastError();
}
// n_key_chunks: 256-key chunks covering the longest context this launch (graph) will see
hipError_t launch_attention_split(const AttnArgs& a, float* S, int n_key_chunks, hipStream_t s) {
#define AS(HS_) return a.gemma ? launch_attention_split_hsg<HS_, true>(a, S, n_key_chunks, s) : launch_attention_split_hsg<HS_, false>(a, S, n_key_chunks, s);
    switch (a.head_size) { case 64: AS(64) case 96: AS(96) case 128: AS(128) case 256: AS(256) default: return hipErrorInvalidValue; }
#undef AS
}

// ------------------------------------------------------------------------------------------------
// Embedding row (transformer.rs:324-332; quantization.rs:25-42) — dequantised on the fly, which is
// bit-identical to reading the reference's load-time f32 copy of the table.
// ------------------------------------------------------------------------------------------------
__global__ void embed_kernel(const EmbedArgs a) {
    const int pos = a.st->pos;
    const uint32_t token = a.tokens[pos];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.dim; i += gridDim.x * blockDim.x) {
        float v = dequant_elem(a.emb_q, a.emb_s, a.q4, (size_t)token * a.dim + i);
        if (a.do_scale) v = v * a.scale;
        a.x[i] = v;
    }
}

hipError_t launch_embed(const EmbedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3((a.dim + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void dequant_rows_kernel(const void* q, const float* s, int q4, const uint32_t* tokens, int dim, float* out) {
    const uint32_t token = tokens[blockIdx.x];
    for (int i = threadIdx.x; i < dim; i += blockDim.x) out[(size_t)blockIdx.x * dim + i] = dequant_elem(q, s, q4, (size_t)token * dim + i);
}

hipError_t launch_dequant_rows(const void* q, const float* s, int q4, const uint32_t* tokens, int n_tok, int dim, float* out, hipStream_t st) {
    hipLaunchKernelGGL(dequant_rows_kernel, dim3(n_tok), dim3(256), 0, st, q, s, q4, tokens, dim, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sample_argmax (sampler.rs:29-41) over the per-workgroup partials, token feedback, position advance.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void argmax_final_kernel(const ArgmaxArgs a) {
    LMRS_STAMP(0);
    if (a.dbg && threadIdx.x == 0) a.dbg[1] = clock64();          // shader-clock cycles, to derive the running clock
    float best = __uint_as_float(0xff800000u); int best_i = 0x7fffffff;
    int nan0 = 0;                                                  // index -1: the logit at index 0 is NaN (cls_flag_nan_at_zero)
    // partials: n_groups shards x n_part entries; shard g holds [values | indices] at part_val + g * group_stride
    const size_t pofs = a.part_par ? (size_t)(*a.part_par & 1u) * a.part_par_floats : 0;        // the half the last exchange delivered
    for (int i = threadIdx.x; i < a.n_part * a.n_groups; i += kBlock) {
        const int g = i / a.n_part, k = i - g * a.n_part;
        const float v = a.part_val[pofs + (size_t)g * a.group_stride + k]; const int idx = a.part_idx[pofs + (size_t)g * a.group_stride + k];
        if (idx < 0) { nan0 = 1; continue; }
        if (v > best || (v == best && idx < best_i)) { best = v; best_i = idx; }
    }
    argmax_tail(a, best, best_i, nan0, tail_preload(a));
    LMRS_STAMP(3);
    if (a.dbg && threadIdx.x == 0) a.dbg[2] = clock64();
}

hipError_t launch_argmax_final(const ArgmaxArgs& a, hipStream_t s) {
    LMRS_LAUNCH_GRID(argmax_final_kernel, dim3(1), kBlock, 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The parallel part of Sampler::sample (sampler.rs:109-129, temperature != 0) on the device:
//   1. logits[i] /= temperature (:115), per-workgroup maxima                                   [grid]
//   2. logits[i] = exp(logits[i] - max) (functional.rs:126-133: max starts at x[0], strict >)   [grid]
// The softmax sum (one sequential chain over all n exponentials, functional.rs:134), the division and sample_mult's running cdf /
// sample_topp's sort run on the host (lmrs_sampler_sample_exps, lmrs_text.cpp): round 4 ran the chains here in one wave, lane by lane
// through the add's DPP operand (~2.5 ns per term against ~1 ns on a host core) and measured it slower than copying the logits.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sample_scale_max_kernel(const SampleArgs a) {
    __shared__ float red[kBlock / 64];
    float m = __uint_as_float(0xff800000u);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += gridDim.x * kBlock) {
        const float v = a.logits[i] / a.temperature;
        a.logits[i] = v;
        if (i == 0) a.part[gridDim.x] = v;                        // x[0]: where the reference's max scan starts (step 2 needs it after block 0 has overwritten it)
        m = fmaxf(m, v);                                          // (NaNs are skipped here; a NaN at index 0 is put back in step 2)
    }
    m = wave64_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) a.part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(kBlock) void sample_exp_kernel(const SampleArgs a, int n_part) {
    __shared__ float red[kBlock / 64];
    float m = __uint_as_float(0xff800000u);
    for (int i = threadIdx.x; i < n_part; i += kBlock) m = fmaxf(m, a.part[i]);
    m = wave64_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float x0 = a.part[n_part];
    if (!(x0 == x0)) mx = x0;                                    // max_val starts at x[0] and only moves on a strict `>`: a NaN there stays
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += gridDim.x * kBlock) a.logits[i] = expf_glibc(a.logits[i] - mx);
}
// ------------------------------------------------------------------------------------------------
// sample_topp's sort (sampler.rs:67-81) on the device, for distributions where most of the vocabulary passes the cutoff (a flat one: > 64 k
// candidates, 7 ms per token in a host stable sort).  The reference fills its candidates in INDEX order and sorts them stably by descending
// probability; (prob, index) is therefore a total order, and any sort of the unique 64-bit keys (~bits(prob) << 32 | index) ascending gives
// exactly that permutation - no stability needed, so an unordered compaction and a bitonic network do: bit-exact by construction.
// prob = exp / sum is the IEEE quotient on both sides; the sum itself is the reference's sequential chain and stays on the host (the
// caller passes it in).  keys: N = the power of two >= n0 entries, padded with ~0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sample_keys_kernel(const float* exps, int n, float sum, float cutoff, unsigned long long* keys, unsigned* count) {
    const int lane = threadIdx.x & 63;
    for (int i0 = (blockIdx.x * kBlock + (int)threadIdx.x - lane); i0 < n; i0 += gridDim.x * kBlock) {      // whole waves stay together (ballot)
        const int i = i0 + lane;
        float p = 0.0f; bool in = false;
        if (i < n) { p = exps[i] / sum; in = p >= cutoff; }                                                   // sampler.rs:75 (NaN: never a candidate)
        const unsigned long long m = __ballot(in);
        if (m) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(m));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (in) keys[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(~__float_as_uint(p)) << 32) | (unsigned)i;
        }
    }
}
__global__ __launch_bounds__(kBlock) void sample_pad_kernel(unsigned long long* keys, const unsigned* count, int N) {
    for (int i = (int)*count + blockIdx.x * kBlock + (int)threadIdx.x; i < N; i += gridDim.x * kBlock) keys[i] = ~0ull;
}
constexpr int kSortBlock = 8192, kSortThreads = 1024;             // keys per workgroup in LDS (64 KB)
// TAIL false: sorts every block of kSortBlock keys (stages k = 2 .. kSortBlock); TAIL true: the steps j = kSortBlock / 2 .. 1 of stage k
template <bool TAIL>
__global__ __launch_bounds__(kSortThreads) void sample_bitonic_local_kernel(unsigned long long* keys, int k_outer) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* t = reinterpret_cast<unsigned long long*>(smem);
    const int base = blockIdx.x * kSortBlock;
    for (int i = threadIdx.x; i < kSortBlock; i += kSortThreads) t[i] = keys[base + i];
    __syncthreads();
    for (int k = TAIL ? k_outer : 2; k <= (TAIL ? k_outer : kSortBlock); k <<= 1) {
        for (int j = (TAIL ? kSortBlock : k) >> 1; j > 0; j >>= 1) {
            for (int q = threadIdx.x; q < kSortBlock / 2; q += kSortThreads) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), l = i | j;        // the pair (i, i + j)
                const bool up = ((base + i) & k) == 0;
                const unsigned long long a = t[i], b = t[l];
                if ((a > b) == up) { t[i] = b; t[l] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < kSortBlock; i += kSortThreads) keys[base + i] = t[i];
}
__global__ __launch_bounds__(kBlock) void sample_bitonic_global_kernel(unsigned long long* keys, int N, int j, int k) {
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < N / 2; q += gridDim.x * kBlock) {
        const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), l = i | j;
        const bool up = (i & k) == 0;
        const unsigned long long a = keys[i], b = keys[l];
        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
    }
}
__global__ __launch_bounds__(kBlock) void sample_pairs_kernel(const unsigned long long* keys, const unsigned* count, float* pairs) {     // -> {prob, index} as sampler.rs:4-8
    const int n0 = (int)*count;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n0; i += gridDim.x * kBlock) {
        const unsigned long long key = keys[i];
        pairs[2 * i] = __uint_as_float(~(unsigned)(key >> 32));
        reinterpret_cast<unsigned*>(pairs)[2 * i + 1] = (unsigned)key;
    }
}
// exps: the n exponentials on the device; keys: room for N = the power of two >= max(n0_host, kSortBlock) keys; count: one zeroed word.
// pairs_out (device): the n0 sorted {prob, index} pairs.  All asynchronous on `s`.
hipError_t launch_sample_topp_sort(const float* exps, int n, float sum, float cutoff, int N, unsigned long long* keys, unsigned* count, float* pairs_out, hipStream_t s) {
    if (N < kSortBlock || (N & (N - 1)) != 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(count, 0, 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sample_keys_kernel, dim3(128), dim3(kBlock), 0, s, exps, n, sum, cutoff, keys, count);
    hipLaunchKernelGGL(sample_pad_kernel, dim3(64), dim3(kBlock), 0, s, keys, (const unsigned*)count, N);
    allow_big_lds(reinterpret_cast<const void*>(sample_bitonic_local_kernel<false>));
    allow_big_lds(reinterpret_cast<const void*>(sample_bitonic_local_kernel<true>));
    const size_t smem = (size_t)kSortBlock * 8;
    hipLaunchKernelGGL(sample_bitonic_local_kernel<false>, dim3(N / kSortBlock), dim3(kSortThreads), smem, s, keys, 0);
    for (int k = 2 * kSortBlock; k <= N; k <<= 1) {
        for (int j = k >> 1; j >= kSortBlock; j >>= 1) hipLaunchKernelGGL(sample_bitonic_global_kernel, dim3(N / 2 / kBlock < 256 ? N / 2 / kBlock : 256), dim3(kBlock), 0, s, keys, N, j, k);
        hipLaunchKernelGGL(sample_bitonic_local_kernel<true>, dim3(N / kSortBlock), dim3(kSortThreads), smem, s, keys, k);
    }
    hipLaunchKernelGGL(sample_pairs_kernel, dim3(128), dim3(kBlock), 0, s, (const unsigned long long*)keys, (const unsigned*)count, pairs_out);
    return hipGetLastError();
}

int sample_sort_min_n() { return kSortBlock; }
hipError_t launch_sample_exps(const SampleArgs& a, hipStream_t s) {
    if (a.n <= 0 || !a.logits || !a.part) return hipErrorInvalidValue;
    LMRS_LAUNCH_GRID(sample_scale_max_kernel, dim3(kSampleGrid), kBlock, 0, s, a);
    LMRS_LAUNCH_GRID(sample_exp_kernel, dim3(kSampleGrid), kBlock, 0, s, a, (int)kSampleGrid);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Thin kernels for the lmrs_op_* entry points: the same device functions as the fused path.
// ------------------------------------------------------------------------------------------------
template <bool Q4>
__global__ __launch_bounds__(kBlock) void quantize_kernel(const float* x, void* q, float* s, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t* xq = reinterpret_cast<int8_t*>(smem);
    float* xs = reinterpret_cast<float*>(smem + ((n + 15) & ~15));
    float4 v[kMaxP];
    load_vec(v, x, n);
    quantize_to_lds<Q4, kMaxP>(v, n, xq, xs, q, s);
}

// Batched prefill on row shards: token t's slice of n values (whole 128-groups) -> [n int8] at q + t * n, its scales at s + t * (n / 128);
// the arithmetic of quantize_to_lds, i.e. bit for bit what quantising the gathered vector gives for these groups.  Q4: the reference's Q4_0
// activation quantiser, the values (q - 8) as int8, de-interleaved within every 8 elements - what the batched matmul_q4 reads (rows_prologue_kernel).
template <bool Q4>
__global__ __launch_bounds__(kBlock) void quantize_rows_kernel(const float* x, int8_t* q, float* s, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t* xq = reinterpret_cast<int8_t*>(smem);
    float* xs = reinterpret_cast<float*>(smem + ((n + 15) & ~15));
    const size_t t = blockIdx.x;
    float4 v[kMaxP];
    load_vec(v, x + t * n, n);
    if constexpr (!Q4) quantize_to_lds<false, kMaxP>(v, n, xq, xs, q + t * n, s + t * (n / kGS));
    else {
        quantize_to_lds<true, kMaxP>(v, n, xq, xs, nullptr, nullptr);
        lds_barrier();
        for (int e = threadIdx.x * 16; e < n; e += kBlock * 16) *reinterpret_cast<int4*>(q + t * n + e) = *reinterpret_cast<const int4*>(xq + e);   // (n is a multiple of 128)
        for (int g = threadIdx.x; g < n / kGS; g += kBlock) s[t * (n / kGS) + g] = xs[g];
    }
}
hipError_t launch_quantize_rows(const float* x, int n, int n_tok, int q4, int8_t* q, float* s, hipStream_t st) {
    if (n % kGS || n > kMaxP * 1024 || n_tok <= 0) return hipErrorInvalidValue;
    const size_t smem = ((n + 15) & ~15) + (size_t)(n / kGS + 4) * 4;
    if (q4) hipLaunchKernelGGL(quantize_rows_kernel<true>, dim3(n_tok), dim3(kBlock), smem, st, x, q, s, n);
    else hipLaunchKernelGGL(quantize_rows_kernel<false>, dim3(n_tok), dim3(kBlock), smem, st, x, q, s, n);
    return hipGetLastError();
}
// ... and the gathered blocks of all shards ([w]: n_tok x n_l int8, then at s_off n_tok x n_l / 128 scales) as the activation operand of the
// GEMM that follows: xq [n_tok][world * n_l], xs [n_tok][world * n_l / 128].  One workgroup per token; 16 bytes per lane.
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const char* blocks, size_t blk_stride, size_t s_off, int world, int n_l, int8_t* xq, float* xs, int xs_ld) {
    const size_t t = blockIdx.x;
    const int n = world * n_l, gl = n_l / kGS;
    for (int e = threadIdx.x * 16; e < n; e += kBlock * 16) {
        const int w = e / n_l, j = e - w * n_l;
        *reinterpret_cast<i32x4*>(xq + t * n + e) = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(blocks + (size_t)w * blk_stride + t * n_l + j));
    }
    for (int g = threadIdx.x; g < n / kGS; g += kBlock) {
        const int w = g / gl, j = g - w * gl;
        const float v = __builtin_nontemporal_load(reinterpret_cast<const float*>(blocks + (size_t)w * blk_stride + s_off) + t * gl + j);
        if (xs_ld) xs[(size_t)g * xs_ld + t] = v; else xs[t * (n / kGS) + g] = v;
    }
}
hipError_t launch_gather_rows(const char* blocks, size_t blk_stride, size_t s_off, int world, int n_l, int n_tok, int8_t* xq, float* xs, hipStream_t st, int xs_ld) {
    if (n_l % kGS || s_off % 16 || blk_stride % 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_tok), dim3(kBlock), 0, st, blocks, blk_stride, s_off, world, n_l, xq, xs, xs_ld);
    return hipGetLastError();
}

// wsT[g * rows + r] = ws[r * groups + g]: the layers' weight scales, once at create, for the ring GEMMs of the batched path (GemmArgs::ws_ld)
__global__ __launch_bounds__(kBlock) void transpose_scales_kernel(const float* __restrict__ ws, int rows, int groups, float* __restrict__ wsT) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, g0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
    for (int k = ty; k < 32; k += 8) { const int r = r0 + k, g = g0 + tx; tile[k][tx] = r < rows && g < groups ? ws[(size_t)r * groups + g] : 0.0f; }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) { const int g = g0 + k, r = r0 + tx; if (g < groups && r < rows) wsT[(size_t)g * rows + r] = tile[tx][k]; }
}
hipError_t launch_transpose_scales(const float* ws, int rows, int groups, float* wsT, hipStream_t s) {
    hipLaunchKernelGGL(transpose_scales_kernel, dim3((rows + 31) / 32, (groups + 31) / 32), dim3(kBlock), 0, s, ws, rows, groups, wsT);
    return hipGetLastError();
}

// Row shards that split wo / w2 too (the split-out plan): every shard's [n_tok x n_l] f32 slice of the projection's output, gathered, becomes columns
// w * n_l .. of the full rows: dst[t][w * n_l + j] = blk_w[t][j] (store: Gemma's branch buffer) or dst + blk (the residual add, transformer.rs:574 / :652 -
// the same single addition per element as the GEMM's own epilogue).  One workgroup per token; 16 bytes per lane.
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(const char* blocks, size_t blk_stride, int world, int n_l, float* dst, int add) {
    const size_t t = blockIdx.x;
    const int n = world * n_l;
    for (int e = threadIdx.x * 4; e < n; e += kBlock * 4) {
        const int w = e / n_l, j = e - w * n_l;
        typedef float f32x4s __attribute__((ext_vector_type(4)));
        const f32x4s v = __builtin_nontemporal_load(reinterpret_cast<const f32x4s*>(blocks + (size_t)w * blk_stride) + (t * n_l + j) / 4);
        float4* d = reinterpret_cast<float4*>(dst + t * n + e);
        if (add) { float4 x = *d; x.x = x.x + v.x; x.y = x.y + v.y; x.z = x.z + v.z; x.w = x.w + v.w; *d = x; }
        else *d = make_float4(v.x, v.y, v.z, v.w);
    }
}
hipError_t launch_scatter_rows(const char* blocks, size_t blk_stride, int world, int n_l, int n_tok, float* dst, int add, hipStream_t st) {
    if (n_l % 4 || blk_stride % 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n_tok), dim3(kBlock), 0, st, blocks, blk_stride, world, n_l, dst, add);
    return hipGetLastError();
}

// The static kernels' quantisers (vec_quantize_q8 / vec_quantize_q4: the prologue code of the decode launches, grouped passes and the
// ragged Gemma lengths included) behind the same entry point, for the shapes they are built for.
template <int N, int NTH, bool Q4>
__global__ __launch_bounds__(NTH) void quantize_static_kernel(const float* x, int8_t* q, float* s) {
    __shared__ __attribute__((aligned(16))) int8_t xq[N];
    __shared__ float xs[N / 128];
    float4 v[(VecGeom<N, NTH>::NP)];
    vec_load<N, false, NTH>(v, x);
    if constexpr (Q4) vec_quantize_q4<N, NTH>(v, xq, xs); else vec_quantize_q8<N, NTH>(v, xq, xs);
    lds_barrier();
    // Q4_0: the LDS image holds every nibble XOR 8 (lmrs_stage.h); the reference's packing is the plain nibbles
    for (int e = threadIdx.x * 4; e < (Q4 ? N / 2 : N); e += NTH * 4) *reinterpret_cast<unsigned*>(q + e) = *reinterpret_cast<const unsigned*>(xq + e) ^ (Q4 ? 0x88888888u : 0u);
    for (int g = threadIdx.x; g < N / 128; g += NTH) s[g] = xs[g];
}

hipError_t launch_quantize(const float* x, void* q, float* s, int n, int q4, hipStream_t st) {
    if (n % kGS || n > kMaxP * 1024) return hipErrorInvalidValue;
    int8_t* q8 = static_cast<int8_t*>(q);
#define QS(N_, NT_) if (n == N_) { if (q4) LMRS_LAUNCH_GRID((quantize_static_kernel<N_, NT_, true>), dim3(1), NT_, 0, st, x, q8, s); \
                                   else LMRS_LAUNCH_GRID((quantize_static_kernel<N_, NT_, false>), dim3(1), NT_, 0, st, x, q8, s); return hipGetLastError(); }
    QS(2048, 256) QS(3072, 256) QS(8192, 512) QS(2304, 256) QS(9216, 512)
#undef QS
    const size_t smem = ((n + 15) & ~15) + (size_t)(n / kGS + 4) * 4;
    if (q4) LMRS_LAUNCH_GRID(quantize_kernel<true>, dim3(1), kBlock, smem, st, x, q, s, n);
    else LMRS_LAUNCH_GRID(quantize_kernel<false>, dim3(1), kBlock, smem, st, x, q, s, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void rmsnorm_kernel(const float* x, const float* w, float* o, int n, float eps, int add_unit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 v[kMaxP], nw[kMaxP];
    load_vec(v, x, n);
    load_vec(nw, w, n);
    rmsnorm_inplace(v, nw, n, eps, add_unit, reinterpret_cast<float*>(smem));
    const int P = (n + 1023) >> 10;
#pragma unroll
    for (int i = 0; i < kMaxP; ++i)
        if (i < P) {
            const int e = i * 1024 + threadIdx.x * 4;
            if (e < n) *reinterpret_cast<float4*>(o + e) = v[i];
        }
}

hipError_t launch_rmsnorm(const float* x, const float* w, float* o, int n, float eps, int add_unit, hipStream_t st) {
    if (n % 32 || n > kMaxP * 1024) return hipErrorInvalidValue;
    const size_t smem = (size_t)(8 * (n / 8 + 4) + 4) * 4;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(1), dim3(kBlock), smem, st, x, w, o, n, eps, add_unit);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Gemma glue (transformer.rs:563-568, 643-650): x += rmsnorm(delta, w, add_unit_offset = true).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void addnorm_kernel(float* x, const float* delta, const float* w, int n, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 v[kMaxP], nw[kMaxP];
    load_vec(v, delta, n);
    load_vec(nw, w, n);
    rmsnorm_inplace(v, nw, n, eps, 1, reinterpret_cast<float*>(smem));
    const int P = (n + 1023) >> 10;
#pragma unroll
    for (int i = 0; i < kMaxP; ++i)
        if (i < P) {
            const int e = i * 1024 + threadIdx.x * 4;
            if (e < n) {
                float4 xv = *reinterpret_cast<const float4*>(x + e);
                xv.x = xv.x + v[i].x; xv.y = xv.y + v[i].y; xv.z = xv.z + v[i].z; xv.w = xv.w + v[i].w;
                *reinterpret_cast<float4*>(x + e) = xv;
            }
        }
}

hipError_t launch_addnorm(float* x, const float* delta, const float* w, int n, float eps, hipStream_t st) {
    if (n % 32 || n > kMaxP * 1024) return hipErrorInvalidValue;
    const size_t smem = (size_t)(8 * (n / 8 + 4) + 4) * 4;
    LMRS_LAUNCH_GRID(addnorm_kernel, dim3(1), kBlock, smem, st, x, delta, w, n, eps);
    return hipGetLastError();
}

// x[i] += d[i]  (row-sharded path: the residual add after the all-gather of a projection's slices)
__global__ void addvec_kernel(float* x, const float* d, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < n) {
        float4 a = *reinterpret_cast<const float4*>(x + i); const float4 b = *reinterpret_cast<const float4*>(d + i);
        a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
        *reinterpret_cast<float4*>(x + i) = a;
    }
}
hipError_t launch_addvec(float* x, const float* d, int n, hipStream_t st) {
    LMRS_LAUNCH_GRID(addvec_kernel, dim3((n / 4 + 255) / 256), 256, 0, st, x, d, n);
    return hipGetLastError();
}

// softmax of functional.rs:122-140 on a vector in global memory (same code shape as attention_kernel)
__global__ __launch_bounds__(kBlock) void softmax_kernel(float* x, int n) {
    __shared__ float red[8];
    const int tid = threadIdx.x;
    float lmax = __uint_as_float(0xff800000u);
    for (int t = tid; t < n; t += kBlock) lmax = fmaxf(lmax, x[t]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int t = tid; t < n; t += kBlock) x[t] = expf_glibc(x[t] - mx);
    __syncthreads();
    if (tid == 0) {
        float sum = 0.0f;
        for (int t = 0; t < n; ++t) sum = sum + x[t];
        red[4] = sum;
    }
    __syncthreads();
    const float sum = red[4];
    for (int t = tid; t < n; t += kBlock) x[t] = x[t] / sum;
}

hipError_t launch_softmax(float* x, int n, hipStream_t st) {
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(kBlock), 0, st, x, n);
    return hipGetLastError();
}

__global__ void expf_kernel(const float* x, float* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = expf_glibc(x[i]);
}

// (float)tanh(c * (double)x): the f64 tanh of Gemma's score / logit soft-caps (c = 1: transformer.rs:520-522, 377-379, after the f32
// division by the cap) and of the tanh-GELU (c = 0.7978845608028654 on the f32 cubic: transformer.rs:614) exactly as the kernels
// evaluate it (ocml's f64 tanh, rounded to f32) - exported so that it can be compared with the host libm the reference calls.
__global__ void tanh_cast_kernel(const float* x, float* y, size_t n, double c) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)tanh(c * (double)x[i]);
}
hipError_t launch_tanh_cast(const float* x, float* y, size_t n, double c, hipStream_t st) {
    hipLaunchKernelGGL(tanh_cast_kernel, dim3(1024), dim3(256), 0, st, x, y, n, c);
    return hipGetLastError();
}

hipError_t launch_expf(const float* x, float* y, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(expf_kernel, dim3(1024), dim3(256), 0, st, x, y, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Peer-to-peer exchange of the row-sharded step (the alternative to ncclAllGather for these latency-bound, few-KB messages).
// One workgroup: [quantise my f32 slice into my block (quantization.rs:44-67; whole 128-groups, so bit-identical to quantising the
// gathered vector) -] copy my block into the same place of every peer's exchange arena (stores that leave over xGMI; in the
// single-device verification modes the "peers" are other contexts / processes on the same GPU), make them visible system-wide,
// raise my flag in every peer's flag row with this exchange's sequence number, then wait until every peer's flag in MY row has
// reached it.  The arena is fine-grained (uncached in L2) memory, so the kernels that follow read what the peers wrote.
// Flags are monotonic per (exchange slot of the step, source shard): no reset, no ABA; all shards run the same sequence of steps.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void exchange_push_kernel(const ExchangeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_seq;
    const int tid = threadIdx.x;
    if (tid == 0) { s_seq = *a.my_seq + 1u; *a.my_seq = s_seq; }
    __syncthreads();
    const size_t pofs = (size_t)(s_seq & 1u) * (size_t)a.par_bytes;    // double-buffered block: this exchange's half
    if (a.qsrc) {                                          // my slice, quantised on its way out: [qn int8 | qn / 128 scales]
        int8_t* xq = reinterpret_cast<int8_t*>(smem);
        float* xs = reinterpret_cast<float*>(smem + ((a.qn + 15) & ~15));
        float4 v[kMaxP];
        load_vec(v, a.qsrc, a.qn);
        char* blk = const_cast<char*>(a.local) + pofs;
        quantize_to_lds<false, kMaxP>(v, a.qn, xq, xs, blk, reinterpret_cast<float*>(blk + a.qn));
        __threadfence();                                   // the block is re-read below by other lanes of this workgroup
    }
    __syncthreads();
    for (int w = 0; w < a.world; ++w) {
        if (w == a.rank) continue;
        for (int off = tid * 16; off < a.bytes; off += kBlock * 16)
            __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const i32x4*>(a.local + pofs + off)), reinterpret_cast<i32x4*>(a.peer_dst[w] + pofs + off));
    }
    __threadfence_system();
    __syncthreads();
    const unsigned seq = s_seq;
    if (tid < a.world && tid != a.rank) {
        __hip_atomic_store(a.peer_flag[tid], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        const bool dead = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // an earlier exchange gave up: do not wait again
        for (; !dead;) {
            const unsigned v = __hip_atomic_load(a.my_flags + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - seq) >= 0) break;
            if (wall_clock64() - t0 > a.timeout_ticks) { *a.err = a.slot + 1; break; }    // a peer is gone: report (check_err), do not hang
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __threadfence_system();
    __syncthreads();
}

// Blocks of a token batch (hundreds of KB): the copy by as many workgroups as it takes, then the push kernel with nothing left to copy
// raises the flags (the kernel boundary orders the two; the stores are system-scope visible when the copy kernel has ended).
__global__ __launch_bounds__(kBlock) void exchange_copy_kernel(const ExchangeArgs a) {
    const size_t stride = (size_t)gridDim.x * kBlock * 16;
    for (int w = 0; w < a.world; ++w) {
        if (w == a.rank) continue;
        for (size_t off = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 16; off < (size_t)a.bytes; off += stride)
            __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const i32x4*>(a.local + off)), reinterpret_cast<i32x4*>(a.peer_dst[w] + off));
    }
    __threadfence_system();
}
hipError_t launch_exchange_copy_push(const ExchangeArgs& a, hipStream_t s) {
    if (a.qsrc || a.par_bytes || a.bytes % 16) return hipErrorInvalidValue;
    const int grid = (int)std::min<size_t>(256, ((size_t)a.bytes + kBlock * 16 - 1) / (kBlock * 16));
    if (grid > 0) LMRS_LAUNCH_GRID(exchange_copy_kernel, dim3(grid), kBlock, 0, s, a);
    ExchangeArgs f = a; f.bytes = 0;
    return launch_exchange_push(f, s);
}

hipError_t launch_exchange_push(const ExchangeArgs& a, hipStream_t s) {
    if (a.qsrc && (a.qn % kGS || a.qn > kMaxP * 1024)) return hipErrorInvalidValue;
    const size_t smem = a.qsrc ? ((a.qn + 15) & ~15) + (size_t)(a.qn / kGS + 4) * 4 : 16;
    LMRS_LAUNCH_GRID(exchange_push_kernel, dim3(1), kBlock, smem, s, a);
    return hipGetLastError();
}

#include "lmrs_prefill.inc"
#include "lmrs_f32.inc"        // (its batched kernel uses the epilogues of lmrs_prefill.inc)
#include "lmrs_vision_att.h"
#include "lmrs_vision.inc"

}  // namespace lmrs
