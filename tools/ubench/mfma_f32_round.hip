// Does a K = 1 f32 MFMA (D = C + A * B, one product per output) round like the reference's separate multiply and add, or like an FMA?
// (round 6: whether the batched attention's chains - sum += q[i] * k[i], out[d] += w[t] * v[t][d] - could run on the matrix pipe bit-exactly)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x1_2b_f32: 2 blocks; A: lane l holds A[block l/32][row l%32]; B: lane l holds B[block l/32][col l%32]; D 32x32 per block, 16 per lane.
__global__ void k32(const float* a, const float* b, const float* c, float* d) {
    const int l = threadIdx.x;
    f32x32 acc;
    for (int i = 0; i < 32; ++i) acc[i] = c[l * 32 + i];
    acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 32; ++i) d[l * 32 + i] = acc[i];
}
// 16x16x4: K = 4 products per output: how are they summed?
__global__ void k16(const float* a, const float* b, const float* c, float* d) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = c[l * 4 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}
static uint32_t rng = 12345;
static float rnd(int mode) {
    rng = rng * 1664525u + 1013904223u; uint32_t r = rng;
    float m = 1.0f + (float)(r & 0x7fffff) / 8388608.0f;
    int e;
    if (mode == 0) e = (int)((r >> 23) & 15) - 8; else if (mode == 1) e = -70 - (int)((r >> 23) & 7); else e = -130 + (int)((r >> 23) & 7);
    float v = ldexpf(m, e);
    return (r >> 31) ? -v : v;
}
int main() {
    for (int mode = 0; mode < 3; ++mode) {
        long n_eq_sep = 0, n_eq_fma = 0, n_tot = 0, n_differ = 0;
        for (int rep = 0; rep < 10; ++rep) {
            std::vector<float> a(64), b(64), c(64 * 32), d(64 * 32);
            for (auto& x : a) x = rnd(mode == 2 ? 1 : mode);
            for (auto& x : b) x = rnd(mode == 2 ? 1 : mode);
            for (auto& x : c) x = mode == 0 ? rnd(0) : (mode == 1 ? rnd(2) : 0.0f);
            float *da, *db, *dc, *dd;
            (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dc, 8192); (void)hipMalloc(&dd, 8192);
            hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 8192, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            hipMemcpy(d.data(), dd, 8192, hipMemcpyDeviceToHost);
            // generic: D[blk][r][cidx]; lane l = 32 * ((r / 4) % 2) + cidx, reg i = 4 * (r / 8) + r % 4; blocks: regs hold both? (32x32x1 2b: 32 regs?)  Use brute force matching.
            for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) {
                const float cc = c[l * 32 + i], dv = d[l * 32 + i];
                bool sep = false, fm = false;
                for (int la = 0; la < 64 && !(sep && fm); ++la) for (int lb = 0; lb < 64; ++lb) {
                    volatile float p = a[la] * b[lb];
                    volatile float s = p + cc;
                    const float f = fmaf(a[la], b[lb], cc);
                    uint32_t us, uf, ud; float sv = s; memcpy(&us, &sv, 4); memcpy(&uf, &f, 4); memcpy(&ud, &dv, 4);
                    if (us != uf) { if (ud == us) sep = true; if (ud == uf) fm = true; }
                }
                ++n_tot; if (sep && !fm) ++n_eq_sep; if (fm && !sep) ++n_eq_fma; if (!sep && !fm) ++n_differ;
            }
            hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
        }
        printf("mode %d (0: normal, 1: denormal products + tiny c, 2: denormal products + 0): outputs %ld  only-separate-rounding %ld  only-fma %ld  neither/ambiguous %ld\n", mode, n_tot, n_eq_sep, n_eq_fma, n_differ);
    }
    return 0;
}
