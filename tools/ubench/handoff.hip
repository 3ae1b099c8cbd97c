// handoff.hip — latency of an in-launch {value, tag} hand-off between two workgroups by cache scope and placement.
// Block 0 stores 64 granules (one per lane); a consumer block on the SAME XCD (block 8: blocks land on XCD b % 8) and one on ANOTHER
// XCD (block 1) poll them.  Stores: plain / workgroup scope (sc0) / agent scope (sc1) / system; loads: the same four.  Memory: hipMalloc
// (coarse-grained), as the decode path's granule buffers.  Time = consumer's wall clock when every tag is this round's minus the
// producer's wall clock just before its stores (100 MHz counter, shared by the chip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int S> __device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    if (S == 0) *(volatile unsigned long long*)p = v;
    else if (S == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (S == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int L> __device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
    if (L == 0) return *(const volatile unsigned long long*)p;
    else if (L == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (L == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr int R = 12;
template <int S, int L>
__global__ __launch_bounds__(64) void k(unsigned long long* gran, long long* t, unsigned* xcc, unsigned base) {
    const int b = blockIdx.x, lane = threadIdx.x;
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (lane == 0) xcc[b] = x & 15;
    if (b != 0 && b != 1 && b != 8) return;
    for (int r = 0; r < R; ++r) {
        const unsigned tag = base + r + 1;
        if (b == 0) {
            const long long w = wall_clock64();
            while (wall_clock64() - w < 300) {}                      // 3 us: both consumers are polling by now
            const long long t0 = wall_clock64();
            st<S>(gran + lane, ((unsigned long long)tag << 32) | lane);             // for block 1
            st<S>(gran + 64 + lane, ((unsigned long long)tag << 32) | lane);        // for block 8
            if (lane == 0) t[r * 3] = t0;
            // wait until both consumers acknowledge (agent-scope flags) before the next round
            while (__hip_atomic_load(gran + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag || __hip_atomic_load(gran + 129, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {}
        } else {
            const unsigned long long* p = gran + (b == 1 ? 0 : 64) + lane;
            long long t1 = 0;
            for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                const unsigned long long v = ld<L>(p);
                if (__all((unsigned)(v >> 32) == tag)) { t1 = wall_clock64(); break; }
            }
            if (lane == 0) { t[r * 3 + (b == 1 ? 1 : 2)] = t1; __hip_atomic_store(gran + (b == 1 ? 128 : 129), (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
}

template <int S, int L> static void run(unsigned long long* gran, long long* t, unsigned* xcc, unsigned& base) {
    static const char* nm[] = {"plain", "workgroup (sc0)", "agent (sc1)", "system (sc0 sc1)"};
    long long h[R * 3]; unsigned hx[16];
    CK(hipMemset(t, 0, sizeof h));
    hipLaunchKernelGGL((k<S, L>), dim3(16), dim3(64), 0, 0, gran, t, xcc, base);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("store %-16s load %-16s: %s\n", nm[S], nm[L], hipGetErrorString(e)); exit(1); }
    CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost));
    base += R + 4;
    double other = 1e9, same = 1e9; int seen_o = 0, seen_s = 0;
    for (int r = 2; r < R; ++r) {
        if (h[r * 3 + 1]) { seen_o++; double d = (h[r * 3 + 1] - h[r * 3]) / 100.0; if (d < other) other = d; }
        if (h[r * 3 + 2]) { seen_s++; double d = (h[r * 3 + 2] - h[r * 3]) / 100.0; if (d < same) same = d; }
    }
    printf("store %-16s load %-16s: other XCD (block 1 on xcc %u) %s %6.2f us   same XCD (block 8 on xcc %u, producer on %u) %s %6.2f us\n", nm[S], nm[L], hx[1],
           seen_o == R - 2 ? "seen, min" : "NOT SEEN  ", seen_o ? other : 0.0, hx[8], hx[0], seen_s == R - 2 ? "seen, min" : "NOT SEEN  ", seen_s ? same : 0.0);
}

int main() {
    unsigned long long* gran; long long* t; unsigned* xcc; unsigned base = 1;
    CK(hipMalloc(&gran, 4096)); CK(hipMalloc(&t, 4096)); CK(hipMalloc(&xcc, 256));
    CK(hipMemset(gran, 0, 4096));
    run<0, 1>(gran, t, xcc, base); run<0, 2>(gran, t, xcc, base); run<0, 3>(gran, t, xcc, base);
    run<1, 1>(gran, t, xcc, base); run<1, 2>(gran, t, xcc, base);
    run<2, 1>(gran, t, xcc, base); run<2, 2>(gran, t, xcc, base); run<2, 3>(gran, t, xcc, base);
    run<3, 2>(gran, t, xcc, base); run<3, 3>(gran, t, xcc, base);
    run<0, 0>(gran, t, xcc, base);
    return 0;
}
