// lmrs_vision_att.hip - attention of the CLIP tower (reference src/vision.rs:383-429), a translation unit of its own: these kernels are hand-ordered
// around scalar loads (s_load_dwordx16 rows, asm-pinned waits), and LLVM's max-ilp scheduling strategy - which the rest of the library is built with
// (Makefile: SCHED) - re-orders them for the worse: vis_att_output_kernel 51.7 -> 84.9 us per layer (profiles/r4_vision_kernel_stats.csv against the
// first round-5 profile).  Built WITHOUT that flag.  Declarations: lmrs_kernels.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lmrs_device_math.h"
#include "lmrs_kernels.h"
#include "lmrs_vision_att.h"

namespace lmrs {

// ------------------------------------------------------------------------------------------------
// Attention of the tower (vision.rs:383-429): non-causal, T = 577 tokens, head size 64.  Scores are matmul_rest(q_t, K) (8
// lane sums over the 8 chunks of the head dims + tree), softmax per row (functional.rs:122-140: sequential sum), outputs are
// matmul_rest(row, V_d): 8 lane sums over the 72 chunks of the keys (lane sum r adds the keys r, r+8, r+16, .. in order),
// the tree, and the scalar tail for key 576.
// A lane owns one query of a block of 64; the order-free work is cut into waves (lmrs_vision_att.h lists the four launches): the keys of the
// score phase by chunk, the exps by an eighth of the keys per wave of the softmax workgroup (whose wave 0 then runs the row's sequential sum, one
// chain per lane), the 8 lane sums of the output phase by residue and half of the dims (a wave visits its 72 keys in order and keeps 32 output
// dims in registers).  K / V rows reach the lanes as scalar operands (s_load_dwordx16: every lane multiplies the same row); the T x 64 score
// slab and the 8 x 64 x 64 partial sums live in global scratch, [key][query] and [residue][dim][query], so that the lanes' accesses coalesce.
// ------------------------------------------------------------------------------------------------
// (shared helpers - slab layout, scalar-row loads: lmrs_vision_att.h)
size_t vis_attention_scratch_floats(int num_crops, int n_heads, int T) { return (size_t)num_crops * n_heads * ((T + kVisQB - 1) / kVisQB) * vis_slab_floats(T); }


// ------------------------------------------------------------------------------------------------
// Stray queries (round 6).  A query block with ONE live lane costs every phase what a full block does, and T = 577 = 9 x 64 + 1: a tenth of the
// attention's waves worked on dead lanes.  The last T % 64 queries (when there are at most kVisStrayMax of them) are taken off the blocked phases -
// those run over the whole blocks only - and get a workgroup each that rides in the softmax launch (whose 288 blocks leave most CUs idle in their second round): threads = keys for the scores (the 8 lane sums over
// the head's dims + tree, vision.rs:391-400), the maximum, exp and the divide (functional.rs:122-140; the row's sequential sum by one thread), then
// threads = (lane sum r, dim d) pairs for matmul_rest(row, V_d) (r adds the keys r, r + 8, .. in order), the tree and the scalar tail.  The same
// operations per value in the same order as the blocked phases: bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int kVisStrayMax = 4, kVisStrayT = 1024;
template <int NT>
__device__ __forceinline__ void vis_att_stray(const float* __restrict__ qkv, float* __restrict__ out, int T, int dim, int t, int head, int crop) {
    __shared__ float P[kVisStrayT];
    __shared__ float part[8 * kVisHS];
    __shared__ float red[NT / 64];
    __shared__ float rowsum;
    const int tid = threadIdx.x;
    const float* base = qkv + (size_t)crop * T * dim * 3;
    const size_t rstride = (size_t)dim * 3;
    float4 q[kVisHS / 4];
#pragma unroll
    for (int u = 0; u < kVisHS / 4; ++u) q[u] = *reinterpret_cast<const float4*>(base + (size_t)t * rstride + head * kVisHS + u * 4);
    float mx = __uint_as_float(0xff800000u);
    for (int k = tid; k < T; k += NT) {
        const float* kr = base + (size_t)k * rstride + dim + head * kVisHS;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kVisHS / 8; ++j) {
            const float4 k0 = *reinterpret_cast<const float4*>(kr + j * 8), k1 = *reinterpret_cast<const float4*>(kr + j * 8 + 4);
            const float4 q0 = q[2 * j], q1 = q[2 * j + 1];
            float pr;
            pr = k0.x * q0.x; s[0] = s[0] + pr; pr = k0.y * q0.y; s[1] = s[1] + pr; pr = k0.z * q0.z; s[2] = s[2] + pr; pr = k0.w * q0.w; s[3] = s[3] + pr;
            pr = k1.x * q1.x; s[4] = s[4] + pr; pr = k1.y * q1.y; s[5] = s[5] + pr; pr = k1.z * q1.z; s[6] = s[6] + pr; pr = k1.w * q1.w; s[7] = s[7] + pr;
        }
        float fs = 0.0f;
        fs = fs + reduce_add8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
        P[k] = fs;
        mx = fmaxf(mx, fs);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, red[w]);
    for (int k = tid; k < T; k += NT) P[k] = expf_glibc(P[k] - mx);      // (a thread's own entries)
    __syncthreads();
    if (tid == 0) {                                                           // functional.rs:134: the sequential sum
        float sum = 0.0f;
        int k = 0;
        for (; k + 16 <= T; k += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = P[k + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) sum = sum + v[u];
        }
        for (; k < T; ++k) sum = sum + P[k];
        rowsum = sum;
    }
    __syncthreads();
    const float sum = rowsum;
    for (int k = tid; k < T; k += NT) P[k] = P[k] / sum;
    __syncthreads();
    const int ns = T / 8;
    for (int idx = tid; idx < 8 * kVisHS; idx += NT) {
        const int r = idx / kVisHS, d = idx % kVisHS;
        const float* v = base + 2 * dim + head * kVisHS + d;
        float ls = 0.0f;
        int j = 0;
        for (; j + 24 <= ns; j += 24) {                                       // 24 value loads in flight (the adds are the chain, the loads are not)
            float vv[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) vv[u] = v[(size_t)((j + u) * 8 + r) * rstride];
#pragma unroll
            for (int u = 0; u < 24; ++u) { const float pr = vv[u] * P[(j + u) * 8 + r]; ls = ls + pr; }
        }
        for (; j < ns; ++j) {
            const int key = j * 8 + r;
            const float pr = v[(size_t)key * rstride] * P[key];
            ls = ls + pr;
        }
        part[idx] = ls;
    }
    __syncthreads();
    if (tid < kVisHS) {
        const int d = tid;
        const float* v = base + 2 * dim + head * kVisHS + d;
        float fs = 0.0f;
        fs = fs + reduce_add8(part[d], part[kVisHS + d], part[2 * kVisHS + d], part[3 * kVisHS + d], part[4 * kVisHS + d], part[5 * kVisHS + d], part[6 * kVisHS + d], part[7 * kVisHS + d]);
        for (int k = ns * 8; k < T; ++k) {
            const float pr = v[(size_t)k * rstride] * P[k];
            fs = fs + pr;
        }
        out[((size_t)crop * T + t) * dim + head * kVisHS + d] = fs;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void vis_att_softmax_kernel(float* __restrict__ scratch, const float* __restrict__ qkv, float* __restrict__ out, int n_heads, int nqb, int T, int dim) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qb = blockIdx.x, head = blockIdx.y, crop = blockIdx.z;
    if (qb >= nqb) { vis_att_stray<64 * NW>(qkv, out, T, dim, nqb * kVisQB + (qb - nqb), head, crop); return; }     // (workgroup-uniform)
    const uint64_t etab = exp2f_tab_lane();
    const VisSlab sl = vis_slab(scratch, crop, n_heads, head, nqb, qb, T, lane);
    float* S = sl.S;
    const int n_chunks = vis_key_chunks(T);
    float mx = __uint_as_float(0xff800000u);
    for (int c = 0; c < n_chunks; c += 8) {                                      // (eight loads in flight; past the last chunk: the last one again)
        float m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = sl.M[(size_t)(c + u < n_chunks ? c + u : n_chunks - 1) * kVisQB];
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, m[u]);
    }
    // ---- exp of this wave's share of the keys: contiguous shares of ceil(T / NW) keys (round 4: by 64-key chunk two of the eight waves
    // had two chunks, 128 double-precision exps per lane against 64 - the launch waited for them)
    const int per = (T + NW - 1) / NW, k_lo = wave * per, k_hi = k_lo + per < T ? k_lo + per : T;
    for (int k8 = k_lo; k8 < k_hi; k8 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = S[(size_t)(k8 + (k8 + u < k_hi ? u : 0)) * kVisQB];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = expf_glibc_t(v[u] - mx, etab);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k8 + u < k_hi) S[(size_t)(k8 + u) * kVisQB] = v[u];
    }
    __syncthreads();                                                              // (drains the S stores: hipcc's barrier waits for vmcnt(0))
    // ---- the row's sequential sum (functional.rs:134): one chain per lane, wave 0.  The adds are one dependent chain, the loads are not:
    // four batches of 16 in flight (round 4: with one batch at a time the chain waited for a memory round trip per 16 terms - 36 of them)
    if (wave == 0) {
        float sum = 0.0f;
        const int nb = T / 16;
        float A[16], B[16], Cc[16], D[16];
        auto ld = [&](float (&v)[16], int b) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = S[(size_t)((b < nb ? b : nb - 1) * 16 + u) * kVisQB];
        };
        auto add = [&](const float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 16; ++u) sum = sum + v[u];
        };
        if (nb > 0) {
            ld(A, 0); ld(B, 1); ld(Cc, 2);
            int b = 0;
            for (; b + 4 <= nb; b += 4) {
                ld(D, b + 3); asm volatile("" ::: "memory"); add(A);
                ld(A, b + 4); asm volatile("" ::: "memory"); add(B);
                ld(B, b + 5); asm volatile("" ::: "memory"); add(Cc);
                ld(Cc, b + 6); asm volatile("" ::: "memory"); add(D);
            }
            // up to three whole batches left: they are already in A, B, Cc (loads past the end were clamped duplicates, never added)
            if (b < nb) { add(A); ++b; }
            if (b < nb) { add(B); ++b; }
            if (b < nb) { add(Cc); ++b; }
        }
        for (int k = nb * 16; k < T; ++k) sum = sum + S[(size_t)k * kVisQB];
        sl.SUM[0] = sum;
    }
}

__global__ __launch_bounds__(64) void vis_att_output_kernel(const float* __restrict__ qkv, float* __restrict__ scratch, int n_heads, int nqb, int T, int dim) {
    // One wave per (query block, head, crop, lane sum r, HALF of the head's dims) - round 6.  With all 64 dims a wave was 72 keys x 64 dims and a launch 2304
    // of them, every one resident from the start: 2.25 per SIMD on average, three on most - the launch took three waves' time.  Halves are 4608 shorter
    // waves, more than fit at once (five per SIMD): the dispatcher fills SIMDs as they drain.  (The weights' divides are done by both halves.)
    constexpr int HD = kVisHS / 2;
    const int lane = threadIdx.x, r = blockIdx.x & 7, half = blockIdx.x >> 3, qb = blockIdx.y % nqb, head = blockIdx.y / nqb, crop = blockIdx.z;
    const float* base = qkv + (size_t)crop * T * dim * 3;
    const size_t rstride = (size_t)dim * 3;
    const VisSlab sl = vis_slab(scratch, crop, n_heads, head, nqb, qb, T, lane);
    const float sum = sl.SUM[0];
    const int n_simd = T / 8, rest = n_simd * 8;
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.0f;
    auto macv = [&](const f32x16v& r0, const f32x16v& r1, float p) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { float pr; pr = r0[u] * p; acc[u] = acc[u] + pr; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { float pr; pr = r1[u] * p; acc[16 + u] = acc[16 + u] + pr; }
    };
    const float* row = base + (size_t)r * rstride + 2 * dim + head * kVisHS + half * HD;   // this half of the value row of key r (workgroup-uniform); the residue's keys are 8 rows apart
    f32x16v A0, A1, B0, B1;
    srow_first(A0, A1, row);
    for (int c0 = 0; c0 < rest; c0 += 64) {
        float p[8];                                                                // this lane's weights of the chunk's 8 keys of the residue: exp / sum (functional.rs:137-139)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int key = c0 + r + 8 * j; p[j] = sl.S[(size_t)(key < rest ? key : rest - 8 + r) * kVisQB]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = p[j] / sum;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                                           // keys in pairs: A holds the even one's half row, B the odd one's
            const int key = c0 + r + 8 * j;
            if (key < rest) {                                                      // workgroup-uniform
                row += key + 8 < rest ? 8 * rstride : 0;
                srow_request(B0, B1, row, A0, A1);
                macv(A0, A1, p[j]);
                srow_wait(B0, B1);
                if (key + 8 < rest) {
                    row += key + 16 < rest ? 8 * rstride : 0;
                    srow_request(A0, A1, row, B0, B1);
                    macv(B0, B1, p[j + 1]);
                    srow_wait(A0, A1);
                }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) sl.P2[(size_t)(r * kVisHS + half * HD + d) * kVisQB] = acc[d];
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void vis_att_tree_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ scratch, int n_heads, int nqb, int T, int dim) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qb = blockIdx.x, head = blockIdx.y, crop = blockIdx.z;
    int t = qb * kVisQB + lane; const bool live = t < T; t = live ? t : T - 1;
    const float* base = qkv + (size_t)crop * T * dim * 3;
    const VisSlab sl = vis_slab(scratch, crop, n_heads, head, nqb, qb, T, lane);
    const float sum = sl.SUM[0];
    const int rest = (T / 8) * 8;
    float* ob = out + ((size_t)crop * T + t) * dim + head * kVisHS;
    constexpr int DPW = kVisHS / NW;
    for (int d = wave * DPW; d < wave * DPW + DPW; ++d) {
        float s[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) s[r] = sl.P2[(size_t)(r * kVisHS + d) * kVisQB];
        float fs = 0.0f;
        fs = fs + reduce_add8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
        for (int k = rest; k < T; ++k) {
            const float vv = base[(size_t)k * dim * 3 + 2 * dim + head * kVisHS + d];
            const float w = sl.S[(size_t)k * kVisQB] / sum;
            const float pr = vv * w;
            fs = fs + pr;
        }
        if (live) ob[d] = fs;
    }
}

hipError_t launch_vis_attention(const float* qkv, float* out, float* scratch, int num_crops, int n_heads, int T, int dim, bool stray_workgroups, hipStream_t s) {
    if (dim != n_heads * kVisHS || T < 64) return hipErrorInvalidValue;
    // one launch per phase, wave-granular grids (round 4: 7.5 -> 6.9 ms for the tower against the single-launch forms, which are gone from the library)
    const bool no_stray = !stray_workgroups;                                   // (LMRS_VIS_NO_STRAY at lmrs_vision_create - A/B and tests: the last block with its dead lanes, as before round 6)
    const int rem = T % kVisQB, n_stray = (!no_stray && rem && rem <= kVisStrayMax && T <= kVisStrayT) ? rem : 0;      // vis_att_stray
    const int nqb = (T - n_stray + kVisQB - 1) / kVisQB;
    if (const hipError_t e = launch_vis_att_scores(qkv, scratch, num_crops, n_heads, nqb, T, dim, s)) return e;     // (lmrs_vision.inc: that kernel gains from the max-ilp strategy, 58.7 -> 51.5 us)
    hipLaunchKernelGGL((vis_att_softmax_kernel<8>), dim3(nqb + n_stray, n_heads, num_crops), dim3(512), 0, s, scratch, qkv, out, n_heads, nqb, T, dim);
    hipLaunchKernelGGL(vis_att_output_kernel, dim3(16, nqb * n_heads, num_crops), dim3(64), 0, s, qkv, scratch, n_heads, nqb, T, dim);
    hipLaunchKernelGGL((vis_att_tree_kernel<4>), dim3(nqb, n_heads, num_crops), dim3(256), 0, s, qkv, out, scratch, n_heads, nqb, T, dim);
    return hipGetLastError();
}

}  // namespace lmrs
