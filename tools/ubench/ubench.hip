// ubench.hip — micro-benchmarks that decide design questions of the decode path on MI355X (gfx950).
//   chain   : cycles per dependent v_add_f32 (the RMSNorm / softmax / V chains are strings of these)
//   l2keep  : does data read by kernel A stay in the XCD L2 for kernel B (same block -> same slice)?
//   boundary: dependent-launch cost inside a hipGraph as a function of the grid size
// build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- chain
template <int ACTIVE>
__global__ void k_chain(float* out, const float* in, long long* t) {
    float a[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) a[u] = in[u * 64 + (threadIdx.x & 63)];
    float p = in[4096 + threadIdx.x];
    long long c0 = 0, c1 = 0;
    if ((threadIdx.x & 63) < ACTIVE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c0 = __builtin_readcyclecounter();
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int u = 0; u < 32; ++u) p = p + a[u];
            asm volatile("" : "+v"(p));
        }
        c1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; }
}

// two independent chains interleaved in one lane (is the pipe the limit or the dependency?)
__global__ void k_chain2(float* out, const float* in, long long* t) {
    float a[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) a[u] = in[u * 64 + (threadIdx.x & 63)];
    float p = in[4096 + threadIdx.x], q = in[4200 + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long c0 = __builtin_readcyclecounter();
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
        for (int u = 0; u < 32; ++u) { p = p + a[u]; q = q + a[31 - u]; }
        asm volatile("" : "+v"(p), "+v"(q));
    }
    long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p + q;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; }
}

// ---------------------------------------------------------------- l2keep
template <bool NT, int SHIFT = 0>
__global__ void k_read(const i32x4* buf, size_t per_block_vec, int* sink, long long* tw) {
    const i32x4* p = buf + (size_t)((blockIdx.x + SHIFT) % gridDim.x) * per_block_vec;
    long long w0 = wall_clock64();
    i32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_block_vec; i += blockDim.x * 8) {
        i32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t j = i + (size_t)u * blockDim.x;
            const i32x4* q = p + (j < per_block_vec ? j : i);
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[0] = 1;
    __syncthreads();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) { tw[blockIdx.x * 2] = w0; tw[blockIdx.x * 2 + 1] = w1; }
}

__global__ void k_empty(int* p) { if (p && threadIdx.x == 1234567) p[0] = 1; }
__global__ void k_touch(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }

static double span_us(const std::vector<long long>& tw, int nb) {
    long long lo = tw[0], hi = tw[1];
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, tw[2 * b]); hi = std::max(hi, tw[2 * b + 1]); }
    return (hi - lo) / 100.0;
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    // ---- chain
    {
        float *in, *out; long long* t;
        CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&t, 64));
        CK(hipMemset(in, 0, 1 << 16));
        long long h;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_chain<64>, dim3(1), dim3(64), 0, s, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            if (rep) printf("chain: 256 dependent v_add_f32, 64 lanes, 1 wave : %lld cycles (%.2f / add)\n", h, h / 256.0);
            hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(64), 0, s, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            if (rep) printf("chain: 256 dependent v_add_f32,  8 lanes, 1 wave : %lld cycles (%.2f / add)\n", h, h / 256.0);
            hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(256), 0, s, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            if (rep) printf("chain: same, 4 waves in the workgroup (1 per SIMD)  : %lld cycles (%.2f / add)\n", h, h / 256.0);
            hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(512), 0, s, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            if (rep) printf("chain: same, 8 waves in the workgroup (2 per SIMD)  : %lld cycles (%.2f / add)\n", h, h / 256.0);
            hipLaunchKernelGGL(k_chain2, dim3(1), dim3(64), 0, s, out, in, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            if (rep) printf("chain2: 2 x 256 interleaved independent chains      : %lld cycles (%.2f / add pair)\n", h, h / 256.0);
        }
    }
    // ---- l2keep
    {
        const int nb = 256;
        const size_t big = (size_t)1 << 30;
        char *flush, *buf; int* sink; long long* tw;
        CK(hipMalloc(&flush, big)); CK(hipMalloc(&buf, 256 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&tw, nb * 16 * 4));
        CK(hipMemset(flush, 1, big)); CK(hipMemset(buf, 1, 256 << 20));
        std::vector<long long> h(nb * 2 * 4);
        for (size_t mb : {16, 32, 48, 64, 96, 128, 192}) {
            const size_t per = (mb << 20) / nb / 16;
            for (int mode = 0; mode < 5; ++mode) {
                // flush: read 1 GiB with 1024 blocks
                hipLaunchKernelGGL(k_read<false>, dim3(1024), dim3(256), 0, s, (const i32x4*)flush, big / 1024 / 16, sink, tw + nb * 2);
                if (mode == 1 || mode == 3) hipLaunchKernelGGL(k_read<false>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw + nb * 2);   // producer / prefetch
                if (mode == 4) hipLaunchKernelGGL((k_read<false, 3>), dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw + nb * 2);          // prefetch from ANOTHER XCD (block b reads slice b + 3)
                if (mode == 3) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, (float*)(buf + (250 << 20)));                           // an unrelated kernel in between
                {
                    if (mode == 2) hipLaunchKernelGGL(k_read<true>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw + nb * 2);   // nt prefetch
                    hipLaunchKernelGGL(k_read<true>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw);
                }
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h.data(), tw, nb * 16, hipMemcpyDeviceToHost));
                const char* nm[] = {"cold", "after plain prefetch kernel", "after nt prefetch kernel", "plain prefetch + unrelated kernel between", "after plain prefetch from another XCD"};
                printf("l2keep %2zu MB, 256 blocks, nt read, %-42s: %.2f us (%.0f GB/s)\n", mb, nm[mode], span_us(h, nb), (mb << 20) / span_us(h, nb) / 1e3);
            }
        }
    }
    // ---- boundary: N dependent launches in a graph
    {
        int* d; CK(hipMalloc(&d, 1 << 20));
        for (int grid : {32, 256, 384, 512, 1024}) {
            for (int nt : {256, 512}) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(nt), 0, s, d);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("boundary: empty kernel grid %4d x %3d threads: %.2f us per launch\n", grid, nt, ms * 1000.0 / 1000.0);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
        }
    }
    return 0;
}
