// bcast.hip — how should a wave-uniform row reach the VALU?  (gfx950; the question behind the CLIP tower's attention kernel,
// lmrs_vision.inc: 64 queries = the lanes of a wave, the key / value rows are the same for every lane.)
// The loop is the score phase of vis_attention_kernel: per key 64 multiplies + 64 adds per lane (8 lane sums over the 8 chunks of
// the head dims, then the tree), the key row delivered
//   L : from the wave's LDS staging area, 16 x ds_read_b128 with every lane on the same address (what the kernel does today);
//   S : by scalar loads (s_load_dwordx8/16 through a constant-address-space pointer) straight into SGPR operands of the VALU;
//   N : not at all (operands already in registers): the VALU floor of the loop.
// Each with 8 waves per CU (one 512-thread workgroup) and 16 (two).  Output: ns per key per wave, and what the score + output
// phases of the real kernel (2 x 72 keys per wave) would take at that rate.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o bcast bcast.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define CONST_AS __attribute__((address_space(4)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HS = 64, KS = 32, RS = HS + 4;

__device__ __forceinline__ float tree8(const float (&s)[8]) {
    const float a = s[0] + s[4], b = s[1] + s[5], c = s[2] + s[6], d = s[3] + s[7];
    const float e = a + c, f = b + d;
    return e + f;
}

// (at most 128 VGPRs, like the real kernel's 125: two workgroups share a CU)
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_scores(const float* __restrict__ Q, const float* __restrict__ K, float* __restrict__ out, int nk) {
    __shared__ __attribute__((aligned(16))) float kv_all[NW][KS * RS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* kv = kv_all[wave];
    float4 q[HS / 4];
#pragma unroll
    for (int u = 0; u < HS / 4; ++u) q[u] = *reinterpret_cast<const float4*>(Q + ((size_t)blockIdx.x * 64 + lane) * HS + u * 4);
    f32x4 kreg[8];                                                  // MODE 2: a row that is already in registers
    if constexpr (MODE == 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) kreg[u] = *reinterpret_cast<const f32x4*>(K + (size_t)lane * HS + u * 4);
    }
    float mx = -1e30f;
    float* S = out + (size_t)blockIdx.x * (NW * nk + 8) * 64 + (size_t)wave * nk * 64 + lane;     // the workgroup's score slab, [key][query]
    const int key0 = (blockIdx.x & 7) * NW * nk + wave * nk;       // this wave's keys (8 different row sets across the grid)
    for (int c0 = 0; c0 < nk; c0 += KS) {
        if constexpr (MODE == 0) {
            constexpr int LPK = 64 / KS, F4 = HS / 4 / LPK;
            const int kl = lane / LPK, part = lane % LPK;
            const float* kp = K + (size_t)(key0 + c0 + kl) * HS + part * F4 * 4;
#pragma unroll
            for (int u = 0; u < F4; ++u) *reinterpret_cast<float4*>(kv + kl * RS + (part * F4 + u) * 4) = *reinterpret_cast<const float4*>(kp + u * 4);
            __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (MODE == 1) {
            // half a row (32 floats = 2 x s_load_dwordx16) ahead: the next half is requested before the current one is multiplied.  Scalar
            // loads return out of order, so every wait is lgkmcnt(0): a request has exactly one half-row of arithmetic to hide behind.
            // (Written with explicit s_load / s_waitcnt: left to itself hipcc moves all four loads of a row to the top of the iteration.)
            // (the half-row about to be multiplied rides through the request as an operand: its multiplies cannot be scheduled above it)
            auto sload2 = [&](f32x16& d0, f32x16& d1, const float* p, f32x16& k0, f32x16& k1) __attribute__((always_inline)) {
                asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40" : "=&s"(d0), "=&s"(d1), "+s"(k0), "+s"(k1) : "s"(p));
            };
            auto swait = [&](f32x16& a, f32x16& b) __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); };
            auto mac = [&](float (&s)[8], const f32x16& r0, const f32x16& r1, int j0) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x16& r = j < 2 ? r0 : r1;
                    const int o = (j & 1) * 8;
                    const float4 q0 = q[2 * (j0 + j)], q1 = q[2 * (j0 + j) + 1];
                    float pr;
                    pr = r[o + 0] * q0.x; s[0] = s[0] + pr; pr = r[o + 1] * q0.y; s[1] = s[1] + pr; pr = r[o + 2] * q0.z; s[2] = s[2] + pr; pr = r[o + 3] * q0.w; s[3] = s[3] + pr;
                    pr = r[o + 4] * q1.x; s[4] = s[4] + pr; pr = r[o + 5] * q1.y; s[5] = s[5] + pr; pr = r[o + 6] * q1.z; s[6] = s[6] + pr; pr = r[o + 7] * q1.w; s[7] = s[7] + pr;
                }
            };
            f32x16 A0, A1, B0, B1;
            const float* row = K + (size_t)(key0 + c0) * HS;              // (uniform: key0 comes from readfirstlane)
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(A0), "=&s"(A1) : "s"(row));
#pragma unroll 1
            for (int kk = 0; kk < KS; ++kk) {
                float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                sload2(B0, B1, row + 32, A0, A1);
                mac(s, A0, A1, 0);
                swait(B0, B1);
                row += (kk + 1 < KS ? HS : 0);
                sload2(A0, A1, row, B0, B1);
                mac(s, B0, B1, 4);
                swait(A0, A1);
                const float fs = tree8(s);
                S[(size_t)(c0 + kk) * 64] = fs;
                mx = fmaxf(mx, fs);
            }
        } else {
        const int nkk = nk - c0 < KS ? nk - c0 : KS;                   // (run-time trip count, as in the real kernel: a constant one is pipelined three keys deep and spills)
        for (int kk = 0; kk < nkk; ++kk) {
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float4* kr = reinterpret_cast<const float4*>(kv + kk * RS);
#pragma unroll
            for (int j = 0; j < HS / 8; ++j) {
                float4 k0, k1;
                if constexpr (MODE == 0) { k0 = kr[2 * j]; k1 = kr[2 * j + 1]; }
                else { const f32x4 a = kreg[(2 * j) & 7], b = kreg[(2 * j + 1) & 7]; k0 = make_float4(a.x, a.y, a.z, a.w); k1 = make_float4(b.x, b.y, b.z, b.w); }
                const float4 q0 = q[2 * j], q1 = q[2 * j + 1];
                float pr;
                pr = k0.x * q0.x; s[0] = s[0] + pr; pr = k0.y * q0.y; s[1] = s[1] + pr; pr = k0.z * q0.z; s[2] = s[2] + pr; pr = k0.w * q0.w; s[3] = s[3] + pr;
                pr = k1.x * q1.x; s[4] = s[4] + pr; pr = k1.y * q1.y; s[5] = s[5] + pr; pr = k1.z * q1.z; s[6] = s[6] + pr; pr = k1.w * q1.w; s[7] = s[7] + pr;
            }
            const float fs = tree8(s);
            S[(size_t)(c0 + kk) * 64] = fs;
            mx = fmaxf(mx, fs);
            if constexpr (MODE == 2) {                                  // keep the loop body from being hoisted
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(kreg[u]));
            }
        }
        }
        if constexpr (MODE == 0) { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    out[(size_t)blockIdx.x * (NW * nk + 8) * 64 + (size_t)NW * nk * 64 + threadIdx.x] = mx;
}

template <int MODE, int NW>
static double run(const float* Q, const float* K, float* out, int grid, int nk) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_scores<MODE, NW>), dim3(grid), dim3(64 * NW), 0, 0, Q, K, out, nk);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_scores<MODE, NW>), dim3(grid), dim3(64 * NW), 0, 0, Q, K, out, nk);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main() {
    const int nk = 96, NW = 8;                            // keys per wave per launch (the real kernel: 72 for 577 keys over 8 waves)
    const size_t nq = (size_t)2560 * 64 * HS, nkf = (size_t)8 * 8 * 4 * nk * HS;
    std::vector<float> h(nq > nkf ? nq : nkf);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    float *Q, *K, *out;
    CK(hipMalloc(&Q, nq * 4)); CK(hipMalloc(&K, nkf * 4)); CK(hipMalloc(&out, (size_t)2560 * (NW * nk + 8) * 64 * 4));
    CK(hipMemcpy(Q, h.data(), nq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(K, h.data(), nkf * 4, hipMemcpyHostToDevice));
    const char* names[3] = {"L (ds_read_b128 broadcast)", "S (scalar loads)", "N (registers: VALU floor)"};
    for (int grid : {256, 512}) {
        double us[3] = {run<0, 8>(Q, K, out, grid, nk), run<1, 8>(Q, K, out, grid, nk), run<2, 8>(Q, K, out, grid, nk)};
        for (int m = 0; m < 3; ++m)
            printf("%2d waves per CU  %-28s %8.1f us per launch  %6.1f ns per key per wave  -> score + output phases of the real kernel (2 x 72 keys per wave): %5.1f us\n",
                   grid / 256 * NW, names[m], us[m], us[m] * 1e3 / nk, us[m] / nk * 144.0);
    }
    // the real launch has 320 workgroups (10 query blocks x 16 heads x 2 crops) of 8 waves on 256 CUs: 64 CUs carry two.  The same 2560 waves
    // x nk keys as workgroups of 4, 2 and 1 waves - what the dispatcher spreads evenly: what is the imbalance worth?
    printf("\nthe same 2560 waves x %d keys as 320 x 8 (the real grid), 640 x 4, 1280 x 2 and 2560 x 1 waves (us per launch; x 144 / %d for score + output):\n", nk, nk);
    {
        double a[3] = {run<0, 8>(Q, K, out, 320, nk), run<1, 8>(Q, K, out, 320, nk), run<2, 8>(Q, K, out, 320, nk)};
        double b[3] = {run<0, 4>(Q, K, out, 640, nk), run<1, 4>(Q, K, out, 640, nk), run<2, 4>(Q, K, out, 640, nk)};
        double c[3] = {run<0, 2>(Q, K, out, 1280, nk), run<1, 2>(Q, K, out, 1280, nk), run<2, 2>(Q, K, out, 1280, nk)};
        double d[3] = {run<0, 1>(Q, K, out, 2560, nk), run<1, 1>(Q, K, out, 2560, nk), run<2, 1>(Q, K, out, 2560, nk)};
        for (int m = 0; m < 3; ++m)
            printf("  %-28s %6.1f (%5.1f)  %6.1f (%5.1f)  %6.1f (%5.1f)  %6.1f (%5.1f)\n", names[m], a[m], a[m] / nk * 144.0, b[m], b[m] / nk * 144.0,
                   c[m], c[m] / nk * 144.0, d[m], d[m] / nk * 144.0);
    }
    return 0;
}
