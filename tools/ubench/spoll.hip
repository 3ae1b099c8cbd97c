// spoll.hip — can a workgroup with a weight tile in flight see a flag EARLY?  A CU answers its vector loads in request order, so a
// vector poll issued behind a 32 KB tile returns after the tile (the in-launch all-gathers of DESIGN.md 3.6b cost 2.3-4.2 us under the
// gate/up stream for that reason).  Scalar loads take another path (lgkmcnt, out of order with respect to vector memory): this
// measures, for 511 consumer workgroups that each have 32 KB of cold HBM reads outstanding, the time from the producer's store of a
// flag to the consumer seeing it - polled by vector loads (agent scope) and by scalar loads (s_load_dword ... glc), flag in
// coarse-grained and in fine-grained memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool LOADED>   // MODE 0: vector agent-scope poll, 1: scalar poll (glc)
__global__ __launch_bounds__(256) void k(unsigned* flag, const i32x4* cold, long long* t, int* sink, unsigned tag) {
    const int b = blockIdx.x;
    if (b == 0) {
        const long long w = wall_clock64();
        while (wall_clock64() - w < 250) {}                       // 2.5 us: every consumer has its tile in flight and is polling
        const long long t0 = wall_clock64();
        if (threadIdx.x == 0) { __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t[0] = t0; }
        return;
    }
    i32x4 v[8];
    if (LOADED) {
        const i32x4* p = cold + (size_t)b * 2048 + threadIdx.x;   // 32 KB per workgroup, never read before
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + u * 256);
    }
    long long t1 = 0;
    if (MODE == 0) {
        for (unsigned spins = 0; spins < (1u << 20); ++spins) {
            const unsigned f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f == tag) { t1 = wall_clock64(); break; }
        }
    } else {
        for (unsigned spins = 0; spins < (1u << 20); ++spins) {
            unsigned f;
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(flag) : "memory");
            if (f == tag) { t1 = wall_clock64(); break; }
        }
    }
    if (threadIdx.x == 0) t[b] = t1;
    if (LOADED) {
        i32x4 acc = v[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) acc ^= v[u];
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) *sink = 1;
    }
}

template <int MODE, bool LOADED> static void run(const char* name, unsigned* flag, const i32x4* cold, long long* t, int* sink, unsigned& tag) {
    const int grid = 512;
    std::vector<long long> h(grid);
    double best_med = 1e9, best_max = 1e9; int unseen = 0;
    for (int rep = 0; rep < 4; ++rep) {
        ++tag;
        CK(hipMemset(t, 0, grid * 8));
        hipLaunchKernelGGL((k<MODE, LOADED>), dim3(grid), dim3(256), 0, 0, flag, cold + (size_t)rep * grid * 2048, t, sink, tag);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost));
        std::vector<double> d;
        for (int b = 1; b < grid; ++b) { if (h[b]) d.push_back((h[b] - h[0]) / 100.0); else ++unseen; }
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        if (rep) { best_med = std::min(best_med, d[d.size() / 2]); best_max = std::min(best_max, d.back()); }
    }
    printf("%-72s median %6.2f us   last %6.2f us   (not seen: %d)\n", name, best_med, best_max, unseen);
}

int main() {
    unsigned *flag_c, *flag_f; i32x4* cold; long long* t; int* sink; unsigned tag = 100;
    CK(hipMalloc(&flag_c, 4096)); CK(hipExtMallocWithFlags((void**)&flag_f, 4096, hipDeviceMallocFinegrained));
    CK(hipMalloc(&cold, (size_t)4 * 512 * 2048 * 16 + (1 << 20))); CK(hipMalloc(&t, 8192)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(flag_c, 0, 4096)); CK(hipMemset(flag_f, 0, 4096)); CK(hipMemset(cold, 1, (size_t)4 * 512 * 2048 * 16));
    run<0, false>("vector poll, idle consumers, coarse-grained flag", flag_c, cold, t, sink, tag);
    run<0, true>("vector poll behind a 32 KB tile, coarse-grained flag", flag_c, cold, t, sink, tag);
    run<1, false>("scalar poll (glc), idle consumers, coarse-grained flag", flag_c, cold, t, sink, tag);
    run<1, true>("scalar poll (glc) beside a 32 KB tile, coarse-grained flag", flag_c, cold, t, sink, tag);
    run<0, true>("vector poll behind a 32 KB tile, fine-grained flag", flag_f, cold, t, sink, tag);
    run<1, false>("scalar poll (glc), idle consumers, fine-grained flag", flag_f, cold, t, sink, tag);
    run<1, true>("scalar poll (glc) beside a 32 KB tile, fine-grained flag", flag_f, cold, t, sink, tag);
    return 0;
}
