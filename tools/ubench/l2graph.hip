// l2graph.hip — does L2 retention across launches survive (a) hipGraph replay, (b) LDS-DMA prefetch loads?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: plain loads, 1: nt loads, 2: LDS-DMA loads
__global__ void k_read(const i32x4* buf, size_t per_block_vec, int* sink, long long* tw) {
    __shared__ char lds[1024];
    const i32x4* p = buf + (size_t)blockIdx.x * per_block_vec;
    long long w0 = wall_clock64();
    i32x4 acc = {0, 0, 0, 0};
    if (MODE == 2) {
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
        for (size_t i = threadIdx.x; i < per_block_vec; i += blockDim.x) {
            const i32x4* g = p + i; unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (size_t i = threadIdx.x; i < per_block_vec; i += blockDim.x * 8) {
            i32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const size_t j = i + (size_t)u * blockDim.x; const i32x4* q = p + (j < per_block_vec ? j : i); v[u] = MODE == 1 ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[0] = 1;
    __syncthreads();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) { tw[blockIdx.x * 2] = w0; tw[blockIdx.x * 2 + 1] = w1; }
}
__global__ void k_touch(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }
static double span_us(const std::vector<long long>& tw, int nb) {
    long long lo = tw[0], hi = tw[1];
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, tw[2 * b]); hi = std::max(hi, tw[2 * b + 1]); }
    return (hi - lo) / 100.0;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int nb = 256; const size_t big = (size_t)1 << 30;
    char *flush, *buf; int* sink; long long* tw;
    CK(hipMalloc(&flush, big)); CK(hipMalloc(&buf, 64 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&tw, nb * 16 * 4));
    CK(hipMemset(flush, 1, big)); CK(hipMemset(buf, 1, 64 << 20));
    std::vector<long long> h(nb * 2);
    const size_t mb = 16, per = (mb << 20) / nb / 16;
    for (int graph = 0; graph < 2; ++graph)
        for (int mode = 0; mode < 4; ++mode) {
            auto body = [&]() {
                hipLaunchKernelGGL(k_read<0>, dim3(1024), dim3(256), 0, s, (const i32x4*)flush, big / 1024 / 16, sink, tw + nb * 2);
                if (mode == 1) hipLaunchKernelGGL(k_read<0>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw + nb * 2);
                if (mode == 2) hipLaunchKernelGGL(k_read<2>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw + nb * 2);
                if (mode == 3) { hipLaunchKernelGGL(k_read<0>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw + nb * 2); hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, (float*)(buf + (48 << 20))); }
                hipLaunchKernelGGL(k_read<1>, dim3(nb), dim3(256), 0, s, (const i32x4*)buf, per, sink, tw);
            };
            if (graph) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); body(); CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); CK(hipGraphLaunch(ge, s));
            } else body();
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h.data(), tw, nb * 16, hipMemcpyDeviceToHost));
            const char* nm[] = {"cold", "after plain-load prefetch", "after LDS-DMA prefetch", "plain prefetch + a writing kernel between"};
            printf("%s 16 MB nt read, %-44s: %.2f us (%.0f GB/s)\n", graph ? "graph " : "stream", nm[mode], span_us(h, nb), (mb << 20) / span_us(h, nb) / 1e3);
        }
    return 0;
}
