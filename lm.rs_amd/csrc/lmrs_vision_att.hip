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
// One workgroup (4 waves) per (crop, head, block of 64 queries).  A lane owns one query in every wave; the waves split the
// order-free work: the keys of the score / exp / divide phases by 64-key chunk, the 8 lane sums of the output phase by
// residue (each visits its 72 keys in order and keeps all 64 output dims in registers).  The row's sequential softmax sum is
// one chain per lane in wave 0.  K / V rows are staged through LDS 64 keys at a time per wave (every lane reads the same row:
// broadcast); the T x 64 score slab and the 8 x 64 x 64 partial sums live in global scratch, [key][query] and
// [residue][dim][query], so that the lanes' accesses coalesce.
// ------------------------------------------------------------------------------------------------
// (shared helpers - slab layout, scalar-row loads: lmrs_vision_att.h)
size_t vis_attention_scratch_floats(int num_crops, int n_heads, int T) { return (size_t)num_crops * n_heads * ((T + kVisQB - 1) / kVisQB) * vis_slab_floats(T); }


template <int NW>
__global__ __launch_bounds__(64 * NW) void vis_att_softmax_kernel(float* __restrict__ scratch, int n_heads, int T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qb = blockIdx.x, head = blockIdx.y, crop = blockIdx.z;
    const uint64_t etab = exp2f_tab_lane();
    const VisSlab sl = vis_slab(scratch, crop, n_heads, head, gridDim.x, qb, T, lane);
    float* S = sl.S;
    const int n_chunks = (T + 63) / 64;
    float mx = sl.M[0];
    for (int c = 1; c < n_chunks; ++c) mx = fmaxf(mx, sl.M[(size_t)c * kVisQB]);
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
    const int lane = threadIdx.x, r = blockIdx.x, qb = blockIdx.y % nqb, head = blockIdx.y / nqb, crop = blockIdx.z;
    const float* base = qkv + (size_t)crop * T * dim * 3;
    const size_t rstride = (size_t)dim * 3;
    const VisSlab sl = vis_slab(scratch, crop, n_heads, head, nqb, qb, T, lane);
    const float sum = sl.SUM[0];
    const int n_simd = T / 8, rest = n_simd * 8;
    float acc[kVisHS];
#pragma unroll
    for (int d = 0; d < kVisHS; ++d) acc[d] = 0.0f;
    auto macv = [&](const f32x16v& r0, const f32x16v& r1, float p, int d0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { float pr; pr = r0[u] * p; acc[d0 + u] = acc[d0 + u] + pr; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { float pr; pr = r1[u] * p; acc[d0 + 16 + u] = acc[d0 + 16 + u] + pr; }
    };
    const float* row = base + (size_t)r * rstride + 2 * dim + head * kVisHS;      // value row of key r (workgroup-uniform); the residue's keys are 8 rows apart
    f32x16v A0, A1, B0, B1;
    srow_first(A0, A1, row);
    for (int c0 = 0; c0 < rest; c0 += 64) {
        float p[8];                                                                // this lane's weights of the chunk's 8 keys of the residue: exp / sum (functional.rs:137-139)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int key = c0 + r + 8 * j; p[j] = sl.S[(size_t)(key < rest ? key : rest - 8 + r) * kVisQB]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = p[j] / sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = c0 + r + 8 * j;
            if (key < rest) {                                                      // workgroup-uniform
                srow_request(B0, B1, row + 32, A0, A1);
                macv(A0, A1, p[j], 0);
                srow_wait(B0, B1);
                row += key + 8 < rest ? 8 * rstride : 0;
                srow_request(A0, A1, row, B0, B1);
                macv(B0, B1, p[j], 32);
                srow_wait(A0, A1);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < kVisHS; ++d) sl.P2[(size_t)(r * kVisHS + d) * kVisQB] = acc[d];
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void vis_att_tree_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ scratch, int n_heads, int T, int dim) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qb = blockIdx.x, head = blockIdx.y, crop = blockIdx.z;
    int t = qb * kVisQB + lane; const bool live = t < T; t = live ? t : T - 1;
    const float* base = qkv + (size_t)crop * T * dim * 3;
    const VisSlab sl = vis_slab(scratch, crop, n_heads, head, gridDim.x, qb, T, lane);
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

hipError_t launch_vis_attention(const float* qkv, float* out, float* scratch, int num_crops, int n_heads, int T, int dim, hipStream_t s) {
    if (dim != n_heads * kVisHS || T < 64) return hipErrorInvalidValue;
    // one launch per phase, wave-granular grids (round 4: 7.5 -> 6.9 ms for the tower against the single-launch forms, which are gone from the library)
    const dim3 grid((T + kVisQB - 1) / kVisQB, n_heads, num_crops);
    const int nqb = (T + kVisQB - 1) / kVisQB;
    if (const hipError_t e = launch_vis_att_scores(qkv, scratch, num_crops, n_heads, T, dim, s)) return e;     // (lmrs_vision.inc: that kernel gains from the max-ilp strategy, 58.7 -> 51.5 us)
    hipLaunchKernelGGL((vis_att_softmax_kernel<8>), grid, dim3(512), 0, s, scratch, n_heads, T);
    hipLaunchKernelGGL(vis_att_output_kernel, dim3(8, nqb * n_heads, num_crops), dim3(64), 0, s, qkv, scratch, n_heads, nqb, T, dim);
    hipLaunchKernelGGL((vis_att_tree_kernel<4>), grid, dim3(256), 0, s, qkv, out, scratch, n_heads, T, dim);
    return hipGetLastError();
}

}  // namespace lmrs
