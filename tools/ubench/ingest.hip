// ingest.hip — what bounds the batched GEMMs' ingest?  (every tiling of gemm_q8_* takes in ~5.5-6 TB/s chip-wide, ~21-24 GB/s per CU,
// whatever it has in flight, with the matrix cores 12 % busy and 88 % of the requests hitting L2.)  This is the LOADER of gemm_q8_lds_kernel alone - 128 x 128
// tiles, 256 threads, per quantisation group 4 x 16 B of the weight tile and 4 x 16 B of the token tile per thread, two workgroups per CU, the grid and
// block -> XCD placement of the w1/w3 projection at 512 tokens (128 x 4 workgroups, K = 2048) - with nothing behind the loads but an XOR.  Variants:
//   depth      register stages per workgroup (the kernel: 2 = one group ahead; here 2 and 4), pinned with sched_barrier - left alone hipcc
//              sinks the loads to their uses and nothing is in flight
//   skew       every workgroup starts its K walk at another group (tests whether the lock-step walk camps on a few L2 channels: at a
//              given group all 128 rows of every tile are 2048 or 8192 bytes apart)
//   pad        row stride K + 128 bytes instead of K (same question, answered by the layout)
//   same       every workgroup reads tile (0, 0): the L2 -> CU rate with everything hot in every XCD
// Output: GB/s chip-wide and per CU.  If skew / pad lift the rate, the GEMMs are bound by channel conflicts, not by bytes per MAC.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void ingest(const char* __restrict__ A, const char* __restrict__ B, int strideA, int strideB, int G, int skew_on, int same, int* sink) {
    const int tid = threadIdx.x;
    const int bx = same ? 0 : blockIdx.x, by = same ? 0 : blockIdx.y;
    const char* pa[4]; const char* pb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = tid + p * 256, row = idx >> 3, ch = idx & 7;
        pa[p] = A + (size_t)(bx * 128 + row) * strideA + ch * 16;
        pb[p] = B + (size_t)(by * 128 + row) * strideB + ch * 16;
    }
    const int skew = skew_on ? (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)G) : 0;
    auto gof = [&](int g) { g = g < G ? g : G - 1; int q = g + skew; return (q >= G ? q - G : q) * 128; };
    i32x4 st[DEPTH][8];
    int acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
        const int o = gof(d);
#pragma unroll
        for (int p = 0; p < 4; ++p) { st[d][p] = *reinterpret_cast<const i32x4*>(pa[p] + o); st[d][4 + p] = *reinterpret_cast<const i32x4*>(pb[p] + o); }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int g = 0; g < G; g += DEPTH) {                       // G % DEPTH == 0
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int slot = (d + DEPTH - 1) % DEPTH;           // the stage freed by step g + d - 1
            const int o = gof(g + d + DEPTH - 1);               // (past the end: the last group again - unconditional loads keep the waits counted)
#pragma unroll
            for (int p = 0; p < 4; ++p) { st[slot][p] = *reinterpret_cast<const i32x4*>(pa[p] + o); st[slot][4 + p] = *reinterpret_cast<const i32x4*>(pb[p] + o); }
            __builtin_amdgcn_sched_barrier(0);                  // the loads stay HERE, ahead of the consumption of the stage loaded DEPTH - 1 steps ago
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += st[d][u].x ^ st[d][u].y ^ st[d][u].z ^ st[d][u].w;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 0x12345678) *sink = acc;
}

template <int DEPTH>
static float run(const char* A, const char* B, int sA, int sB, int G, int skew, int same, int n_rt, int n_tt, int* sink, int reps) {
    hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(ingest<DEPTH>, dim3(n_rt, n_tt), dim3(256), 0, 0, A, B, sA, sB, G, skew, same, sink);
    HIPC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(ingest<DEPTH>, dim3(n_rt, n_tt), dim3(256), 0, 0, A, B, sA, sB, G, skew, same, sink);
    HIPC(hipEventRecord(e1, 0)); HIPC(hipEventSynchronize(e1));
    float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
    HIPC(hipEventDestroy(e0)); HIPC(hipEventDestroy(e1));
    return ms * 1e3f / reps;                                    // us per launch
}

int main() {
    HIPC(hipSetDevice(0));
    const int o = 16384, n_tok = 512, reps = 20;
    int* sink; HIPC(hipMalloc(&sink, 4));
    const int Ks[] = {2048, 8192};
    for (int K : Ks) {
        const int G = K / 128, n_rt = o / 128, n_tt = n_tok / 128;
        char *A, *B;
        HIPC(hipMalloc(&A, (size_t)o * (K + 128))); HIPC(hipMalloc(&B, (size_t)n_tok * (K + 128)));
        HIPC(hipMemset(A, 1, (size_t)o * (K + 128))); HIPC(hipMemset(B, 2, (size_t)n_tok * (K + 128)));
        const double bytes = (double)n_rt * n_tt * G * 32768.0;
        printf("K = %d: %d x %d workgroups of 256 threads (two per CU), %d groups, %.0f MB taken in per launch (%.1f MB unique)\n", K, n_rt, n_tt, G, bytes / 1e6,
               ((double)o * K + (double)n_tok * K) / 1e6);
        for (int same = 0; same <= 1; ++same)
            for (int pad = 0; pad <= 1; ++pad)
                for (int skew = 0; skew <= 1; ++skew) {
                    if (same && (pad || skew)) continue;
                    const int sA = K + (pad ? 128 : 0), sB = sA;
                    const float u2 = run<2>(A, B, sA, sB, G, skew, same, n_rt, n_tt, sink, reps);
                    const float u4 = run<4>(A, B, sA, sB, G, skew, same, n_rt, n_tt, sink, reps);
                    printf("  %-34s depth 2: %7.1f us = %6.0f GB/s (%5.1f per CU)   depth 4: %7.1f us = %6.0f GB/s (%5.1f per CU)\n",
                           same ? "every workgroup reads tile (0,0)" : pad ? (skew ? "row stride K+128, skewed K walk" : "row stride K+128, lock-step K walk") : (skew ? "row stride K, skewed K walk" : "row stride K, lock-step K walk (GEMM)"),
                           u2, bytes / u2 / 1e3, bytes / u2 / 1e3 / 256, u4, bytes / u4 / 1e3, bytes / u4 / 1e3 / 256);
                }
        HIPC(hipFree(A)); HIPC(hipFree(B));
    }
    return 0;
}
