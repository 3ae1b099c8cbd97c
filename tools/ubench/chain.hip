// chain.hip — what does a dependent v_add_f32 cost in the configuration the GEMV prologues run it in?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: every wave runs the chain; 1: only wave 0, the others wait at the barrier; 2: as 1 + s_setprio 3
// 3: only wave 0, chain source = LDS reads (as in vec_rmsnorm)
template <int MODE, int ACTIVE>
__global__ void k(float* out, const float* in, long long* t) {
    __shared__ float lds[8 * 260];
    float a[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) a[u] = in[u * 64 + (threadIdx.x & 63)];
    for (int i = threadIdx.x; i < 8 * 260; i += blockDim.x) lds[i] = in[i];
    float p = in[4096 + threadIdx.x];
    __syncthreads();
    long long best = 1ll << 60;
    for (int r = 0; r < 6; ++r) {
        long long c0 = 0, c1 = 0;
        const bool me = (MODE == 0 || threadIdx.x < 64) && (threadIdx.x & 63) < ACTIVE;
        if (me) {
            if (MODE == 2) __builtin_amdgcn_s_setprio(3);
            c0 = __builtin_readcyclecounter();
            if (MODE == 4) {                      // software-pipelined: one 16-byte LDS read per 4 adds, D reads ahead
                const float4* row = reinterpret_cast<const float4*>(lds + (threadIdx.x & 7) * 260);
                constexpr int D = 8;
                float4 R[D];
#pragma unroll
                for (int u = 0; u < D; ++u) R[u] = row[u];
#pragma unroll
                for (int u = 0; u < 64; ++u) {
                    const float4 c = R[u % D];
                    p = p + c.x; p = p + c.y; p = p + c.z; p = p + c.w;
                    if (u + D < 64) R[u % D] = row[u + D];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (MODE == 5) {               // same with 8-byte reads, one per 2 adds
                const float2* row = reinterpret_cast<const float2*>(lds + (threadIdx.x & 7) * 260);
                constexpr int D = 16;
                float2 R[D];
#pragma unroll
                for (int u = 0; u < D; ++u) R[u] = row[u];
#pragma unroll
                for (int u = 0; u < 128; ++u) {
                    const float2 c = R[u % D];
                    p = p + c.x; p = p + c.y;
                    if (u + D < 128) R[u % D] = row[u + D];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (MODE == 3) {
                const float4* row = reinterpret_cast<const float4*>(lds + (threadIdx.x & 7) * 260);
#pragma unroll
                for (int rep = 0; rep < 4; ++rep) {
                    float4 A[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) A[u] = row[rep * 16 + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { p = p + A[u].x; p = p + A[u].y; p = p + A[u].z; p = p + A[u].w; }
                }
            } else {
#pragma unroll
                for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) p = p + a[u];
                    asm volatile("" : "+v"(p));
                }
            }
            c1 = __builtin_readcyclecounter();
            if (MODE == 2) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        if (c1 - c0 < best) best = c1 - c0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = best;
}

template <int MODE, int ACTIVE>
static void run(const char* name, int nt, int grid, float* out, float* in, long long* t) {
    long long h = 0;
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(grid), dim3(nt), 0, 0, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); }
    printf("%-64s threads %4d grid %4d: %5lld cycles = %.2f / add\n", name, nt, grid, h, h / 256.0);
}

int main() {
    float *in, *out; long long* t;
    CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&t, 64));
    CK(hipMemset(in, 0, 1 << 16));
    for (int grid : {1, 256, 512}) {
        run<0, 64>("all waves chain, 64 lanes", 64, grid, out, in, t);
        run<0, 8>("all waves chain, 8 lanes", 64, grid, out, in, t);
        run<0, 8>("all waves chain, 8 lanes", 256, grid, out, in, t);
        run<1, 8>("wave 0 chain (8 lanes), others at the barrier", 256, grid, out, in, t);
        run<1, 64>("wave 0 chain (64 lanes), others at the barrier", 256, grid, out, in, t);
        run<2, 8>("wave 0 chain (8 lanes) at prio 3, others at the barrier", 256, grid, out, in, t);
        run<1, 8>("wave 0 chain (8 lanes), others at the barrier", 512, grid, out, in, t);
        run<3, 8>("wave 0 chain from LDS float4 reads, others at the barrier", 256, grid, out, in, t);
        run<3, 8>("wave 0 chain from LDS float4 reads, others at the barrier", 512, grid, out, in, t);
        run<4, 8>("wave 0 chain, LDS b128 reads interleaved 1 per 4 adds, 8 ahead", 256, grid, out, in, t);
        run<4, 8>("wave 0 chain, LDS b128 reads interleaved 1 per 4 adds, 8 ahead", 512, grid, out, in, t);
        run<5, 8>("wave 0 chain, LDS b64 reads interleaved 1 per 2 adds, 16 ahead", 256, grid, out, in, t);
    }
    return 0;
}
