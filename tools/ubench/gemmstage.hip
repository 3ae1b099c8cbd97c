// gemmstage.hip — where does a group iteration of gemm_q8_lds_kernel spend its time?  (tools/ubench/ingest.hip: the loader alone runs 3.4 x faster than
// the kernel, so the time is inside the workgroup.)  The kernel of lm.rs_amd/csrc/lmrs_prefill.inc, rebuilt stage by stage - same tiles (128 x 128, 4 waves
// of 64 x 64), same loaders, same LDS layout (row stride 144 B, double-buffered), same grid as the w1/w3 projection at 512 tokens:
//   stage 0   global loads only (one group ahead), folded into an XOR
//   stage 1   + LDS stores of the prefetched group and the barrier per group
//   stage 2   + the 16 fragment reads and the scale reads per wave and group (folded into the XOR)
//   stage 3   + the 32 MFMAs per wave and group (integer results folded into the XOR)
//   stage 4   + the float combine (cvt, two multiplies, one add per element): the whole kernel without its epilogue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4m __attribute__((ext_vector_type(4)));
typedef float f32x4m __attribute__((ext_vector_type(4)));
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ int fold(const i32x4m& v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int STAGE>
__global__ __launch_bounds__(256, 2) void gemm_stage(const int8_t* __restrict__ wq, const int8_t* __restrict__ xq, const float* __restrict__ ws,
                                                     const float* __restrict__ xs, int K, int o, int n_tok, int* sink) {
    constexpr int TM = 128, TN = 128, LD = 144;
    __shared__ __attribute__((aligned(16))) char As[2][TM * LD];
    __shared__ __attribute__((aligned(16))) char Bs[2][TN * LD];
    __shared__ __attribute__((aligned(16))) float Wsc[2][TM];
    __shared__ __attribute__((aligned(16))) float Xsc[2][TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, kb = lane >> 4;
    const int G = K / 128;
    const int r_base = blockIdx.x * TM, t_base = blockIdx.y * TN;
    const int8_t* gA[4]; const int8_t* gB[4]; int lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = tid + p * 256, row = idx >> 3, ch = idx & 7;
        int r = r_base + row; r = r < o ? r : o - 1;
        int t = t_base + row; t = t < n_tok ? t : n_tok - 1;
        gA[p] = wq + (size_t)r * K + ch * 16;
        gB[p] = xq + (size_t)t * K + ch * 16;
        lo[p] = row * LD + ch * 16;
    }
    const float* gS;
    {
        const int i = tid & 127;
        if (tid < 128) { int r = r_base + i; r = r < o ? r : o - 1; gS = ws + (size_t)r * G; }
        else { int t = t_base + i; t = t < n_tok ? t : n_tok - 1; gS = xs + (size_t)t * G; }
    }
    i32x4m ra[4], rb[4]; float rs;
    int iacc = 0;
    auto gload = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { ra[p] = *reinterpret_cast<const i32x4m*>(gA[p] + g * 128); rb[p] = *reinterpret_cast<const i32x4m*>(gB[p] + g * 128); }
        rs = gS[g];
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        if constexpr (STAGE >= 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { *reinterpret_cast<i32x4m*>(As[buf] + lo[p]) = ra[p]; *reinterpret_cast<i32x4m*>(Bs[buf] + lo[p]) = rb[p]; }
            if (tid < 128) Wsc[buf][tid] = rs; else Xsc[buf][tid - 128] = rs;
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) iacc ^= fold(ra[p]) ^ fold(rb[p]);
            iacc ^= __float_as_int(rs);
        }
    };
    f32x4m acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = f32x4m{0.f, 0.f, 0.f, 0.f};

    gload(0); lstore(0);
    if constexpr (STAGE >= 1) __syncthreads();
    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        if (g + 1 < G) gload(g + 1);
        __builtin_amdgcn_sched_barrier(0);          // (the prefetch is issued here in every stage - where the product kernel's schedule has it - not sunk to its use)
        if constexpr (STAGE >= 2) {
            i32x4m a0[4], a1[4], b0[4], b1[4]; f32x4m wsv[4]; float xsv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const char* pa = As[buf] + (wm * 64 + m * 16 + lr) * LD + kb * 16;
                a0[m] = *reinterpret_cast<const i32x4m*>(pa); a1[m] = *reinterpret_cast<const i32x4m*>(pa + 64);
                wsv[m] = *reinterpret_cast<const f32x4m*>(&Wsc[buf][wm * 64 + m * 16 + kb * 4]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* pb = Bs[buf] + (wn * 64 + j * 16 + lr) * LD + kb * 16;
                b0[j] = *reinterpret_cast<const i32x4m*>(pb); b1[j] = *reinterpret_cast<const i32x4m*>(pb + 64);
                xsv[j] = Xsc[buf][wn * 64 + j * 16 + lr];
            }
            if constexpr (STAGE == 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m) iacc ^= fold(a0[m]) ^ fold(a1[m]) ^ fold(b0[m]) ^ fold(b1[m]) ^ __float_as_int(wsv[m][0] + wsv[m][1] + wsv[m][2] + wsv[m][3]) ^ __float_as_int(xsv[m]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        i32x4m c = {0, 0, 0, 0};
                        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[m], b0[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[m], b1[j], c, 0, 0, 0);
                        if constexpr (STAGE == 3) {
                            iacc ^= fold(c);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float p = (float)c[i] * wsv[m][i];
                                p = p * xsv[j];
                                acc[m][j][i] = acc[m][j][i] + p;
                            }
                        }
                    }
                if constexpr (STAGE == 3) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) iacc ^= __float_as_int(wsv[m][0] + wsv[m][1] + wsv[m][2] + wsv[m][3]) ^ __float_as_int(xsv[m]);
                }
            }
        }
        if (g + 1 < G) lstore(buf ^ 1);
        if constexpr (STAGE >= 1) __syncthreads();
    }
    float fs = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) fs += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
    iacc ^= __float_as_int(fs);
    if (iacc == 0x12345678) *sink = iacc;
}

template <int STAGE>
static float run(const int8_t* wq, const int8_t* xq, const float* ws, const float* xs, int K, int o, int n_tok, int* sink, int reps) {
    hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    const dim3 grid((o + 127) / 128, (n_tok + 127) / 128);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gemm_stage<STAGE>, grid, dim3(256), 0, 0, wq, xq, ws, xs, K, o, n_tok, sink);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gemm_stage<STAGE>, grid, dim3(256), 0, 0, wq, xq, ws, xs, K, o, n_tok, sink);
    HIPC(hipEventRecord(e1, 0)); HIPC(hipEventSynchronize(e1));
    float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
    HIPC(hipEventDestroy(e0)); HIPC(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

int main() {
    HIPC(hipSetDevice(0));
    const int reps = 20;
    int* sink; HIPC(hipMalloc(&sink, 4));
    struct Shape { int K, o, n_tok; const char* what; };
    const Shape shapes[] = {{2048, 16384, 512, "w1/w3 of Llama-3.2-1B, 512 tokens"}, {2048, 16384, 2048, "the same, 2048 tokens"}, {8192, 16384, 512, "K = 8192"}};
    for (const Shape& sh : shapes) {
        const int G = sh.K / 128;
        int8_t *wq, *xq; float *ws, *xs;
        HIPC(hipMalloc(&wq, (size_t)sh.o * sh.K)); HIPC(hipMalloc(&xq, (size_t)sh.n_tok * sh.K));
        HIPC(hipMalloc(&ws, (size_t)sh.o * G * 4)); HIPC(hipMalloc(&xs, (size_t)sh.n_tok * G * 4));
        HIPC(hipMemset(wq, 3, (size_t)sh.o * sh.K)); HIPC(hipMemset(xq, 5, (size_t)sh.n_tok * sh.K));
        HIPC(hipMemset(ws, 0x3c, (size_t)sh.o * G * 4)); HIPC(hipMemset(xs, 0x3c, (size_t)sh.n_tok * G * 4));
        const double macs = (double)sh.o * sh.n_tok * sh.K;
        const int nwg = ((sh.o + 127) / 128) * ((sh.n_tok + 127) / 128);
        printf("%s: K = %d, o = %d, %d tokens: %d workgroups, %d groups\n", sh.what, sh.K, sh.o, sh.n_tok, nwg, G);
        const float u0 = run<0>(wq, xq, ws, xs, sh.K, sh.o, sh.n_tok, sink, reps);
        const float u1 = run<1>(wq, xq, ws, xs, sh.K, sh.o, sh.n_tok, sink, reps);
        const float u2 = run<2>(wq, xq, ws, xs, sh.K, sh.o, sh.n_tok, sink, reps);
        const float u3 = run<3>(wq, xq, ws, xs, sh.K, sh.o, sh.n_tok, sink, reps);
        const float u4 = run<4>(wq, xq, ws, xs, sh.K, sh.o, sh.n_tok, sink, reps);
        printf("  stage 0 loads only            %8.1f us\n  stage 1 + LDS stores, barrier %8.1f us\n  stage 2 + fragment reads      %8.1f us\n"
               "  stage 3 + MFMAs               %8.1f us\n  stage 4 + float combine       %8.1f us = %.0f int8 TOP/s\n", u0, u1, u2, u3, u4, 2 * macs / u4 / 1e6);
        HIPC(hipFree(wq)); HIPC(hipFree(xq)); HIPC(hipFree(ws)); HIPC(hipFree(xs));
    }
    return 0;
}
