// dpp.hip — cost of a dependent add whose accumulator moves one lane per step (v_add_f32 ... row_ror:1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }

// MODE 0: compiler-built (update_dpp + add), row_ror:1 = 0x121
// MODE 1: hand asm without nops; MODE 2: hand asm with s_nop 1; MODE 3: hand asm, s_nop 0
template <int MODE>
__global__ void k(float* out, const float* in, long long* t, float* chk) {
    float a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = in[u * 64 + (threadIdx.x & 63)];
    float p = 0.0f;
    __syncthreads();
    long long best = 1ll << 60;
    for (int r = 0; r < 6; ++r) {
        p = 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long c0 = __builtin_readcyclecounter();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (MODE == 0) p = dpp_f<0x121>(p) + a[m];
                else if (MODE == 1) asm volatile("v_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(a[m]));
                else if (MODE == 2) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(a[m]));
                else asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(a[m]));
            }
        }
        long long c1 = __builtin_readcyclecounter();
        if (c1 - c0 < best) best = c1 - c0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x < 64 && blockIdx.x == 0) chk[threadIdx.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = best;
}

int main() {
    float *in, *out, *chk; long long* t;
    CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&t, 64)); CK(hipMalloc(&chk, 256));
    float h[16 * 64];
    srand(1);
    for (int i = 0; i < 16 * 64; ++i) h[i] = (float)rand() / RAND_MAX;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    // expected: row 0 chain: element j at lane j % 16, register j / 16; after 256 steps the value sits in lane 15
    float ref = 0.0f;
    for (int j = 0; j < 256; ++j) ref = ref + h[(j / 16) * 64 + (j % 16)];
    const char* nm[] = {"compiler (update_dpp + add)", "asm v_add_f32_dpp, no nop", "asm s_nop 1 + v_add_f32_dpp", "asm s_nop 0 + v_add_f32_dpp"};
    for (int mode = 0; mode < 4; ++mode) {
        long long c; float g[64];
        for (int i = 0; i < 2; ++i) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, in, t, chk);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, in, t, chk);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, in, t, chk);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, in, t, chk);
            CK(hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(g, chk, 256, hipMemcpyDeviceToHost));
        }
        printf("%-32s: %5lld cycles = %.2f / add; lane15 = %.9g ref %.9g %s\n", nm[mode], c, c / 256.0, g[15], ref, g[15] == ref ? "EXACT" : "DIFF");
    }
    return 0;
}
