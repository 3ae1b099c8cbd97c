// (every chain is ONE asm block: between separate asm statements hipcc inserts an s_nop, which costs an issue slot)
// addlat.hip — latency of a dependent f32 add by where the running sum sits in the instruction (src0 / src1 / DPP forms / fmac):
// the serial chains of the decode path (RMSNorm partials, row combines, softmax sums, V accumulation) are strings of exactly these.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define CHAIN(name, text)                                                                                   \
    __global__ __launch_bounds__(256) void name(float* out, const float* in, long long* t) {                \
        float p = in[threadIdx.x], a = in[256 + (threadIdx.x & 63)], b = in[512 + (threadIdx.x & 63)];       \
        (void)b;                                                                                             \
        long long best = 1ll << 60;                                                                          \
        for (int r = 0; r < 4; ++r) {                                                                        \
            if (threadIdx.x < 64) {                                                                          \
                const long long c0 = __builtin_readcyclecounter();                                           \
                asm volatile(".rept 256\n\t" text "\n\t.endr" : "+v"(p) : "v"(a), "v"(b)); \
                const long long c1 = __builtin_readcyclecounter();                                           \
                if (c1 - c0 < best) best = c1 - c0;                                                          \
            }                                                                                                \
            __syncthreads();                                                                                 \
        }                                                                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = p;                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = best;                                                \
    }

CHAIN(k_src0, "v_add_f32 %0, %0, %1")
CHAIN(k_src1, "v_add_f32 %0, %1, %0")
CHAIN(k_e64_src0, "v_add_f32_e64 %0, %0, %1")
CHAIN(k_e64_src1, "v_add_f32_e64 %0, %1, %0")
CHAIN(k_fmac, "v_fmac_f32 %0, 1.0, %1")
CHAIN(k_fma, "v_fma_f32 %0, %1, 1.0, %0")
CHAIN(k_dpp_acc_src1, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
CHAIN(k_dpp_acc_dpp, "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
CHAIN(k_two_src1, "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0")
CHAIN(k_mix44, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf")
CHAIN(k_nop_between, "v_add_f32 %0, %1, %0\n\ts_nop 0")
CHAIN(k_sub_src1, "v_sub_f32 %0, %1, %0")
CHAIN(k_mul_src1, "v_mul_f32 %0, %1, %0")
CHAIN(k_max_src1, "v_max_f32 %0, %1, %0")
CHAIN(k_add_u32, "v_add_u32 %0, %1, %0")

template <class K> static void run(const char* name, K k, float* out, float* in, long long* t, int per) {
    long long h = 0;
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); }
    printf("%-64s %6lld cycles per 256 = %.2f per instruction\n", name, h, h / (256.0 * per));
}

int main() {
    float *in, *out; long long* t;
    CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&t, 64));
    CK(hipMemset(in, 0, 1 << 16));
    run("v_add_f32 acc, acc, x      (sum in src0)", k_src0, out, in, t, 1);
    run("v_add_f32 acc, x, acc      (sum in src1)", k_src1, out, in, t, 1);
    run("v_add_f32_e64 acc, acc, x", k_e64_src0, out, in, t, 1);
    run("v_add_f32_e64 acc, x, acc", k_e64_src1, out, in, t, 1);
    run("v_fmac_f32 acc, 1.0, x", k_fmac, out, in, t, 1);
    run("v_fma_f32 acc, x, 1.0, acc", k_fma, out, in, t, 1);
    run("v_add_f32_dpp acc, x(dpp), acc   (sum in src1)", k_dpp_acc_src1, out, in, t, 1);
    run("s_nop 1; v_add_f32_dpp acc, acc(dpp), x", k_dpp_acc_dpp, out, in, t, 1);
    run("two adds per step, sum in src1", k_two_src1, out, in, t, 2);
    run("rms_chain8 pattern: 4 plain + 4 DPP-fed adds", k_mix44, out, in, t, 8);
    run("v_add_f32 + s_nop 0 (what hipcc puts between asm statements)", k_nop_between, out, in, t, 1);
    run("v_sub_f32 acc, x, acc", k_sub_src1, out, in, t, 1);
    run("v_mul_f32 acc, x, acc", k_mul_src1, out, in, t, 1);
    run("v_max_f32 acc, x, acc", k_max_src1, out, in, t, 1);
    run("v_add_u32 acc, x, acc", k_add_u32, out, in, t, 1);
    return 0;
}
