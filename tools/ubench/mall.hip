// mall.hip — does the 256 MiB Infinity Cache serve a weight stream faster than HBM?  (If it did, a throttled prefetcher one layer ahead
// of the decode kernels could shorten the stream-bound launches: w1/w3 at 34.6 MB, the classifier at 271 MB.)
// A streaming read of B bytes (16 B per lane, 8 loads in flight per lane, 512 x 256 threads - the GEMV kernels' shape), repeated
// back to back on the same buffer: for B well under 256 MiB every pass after the first is served on-die; for B = 2 GiB none is.
// Both load policies: default and non-temporal (what the GEMV kernels use for weights).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void stream_read(const i32x4* __restrict__ p, size_t n16, int* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 8) {
        i32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t j = i + u * stride;
            const i32x4* q = p + (j < n16 ? j : i);
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678) *sink = acc;
}

int main() {
    HIPC(hipSetDevice(0));
    const size_t maxb = (size_t)2 << 30;
    char* buf; int* sink;
    HIPC(hipMalloc(&buf, maxb)); HIPC(hipMalloc(&sink, 4));
    HIPC(hipMemset(buf, 1, maxb));
    hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)16 << 20, (size_t)35 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)192 << 20, (size_t)271 << 20, (size_t)512 << 20, (size_t)2 << 30};
    for (int nt = 0; nt <= 1; ++nt)
        for (size_t b : sizes) {
            const size_t n16 = b / 16;
            const int reps = (int)(((size_t)8 << 30) / b) < 4 ? 4 : (int)(((size_t)8 << 30) / b);
            // a pass over another region first, so that the first timed pass starts cold
            hipLaunchKernelGGL(stream_read<false>, dim3(512), dim3(256), 0, 0, (const i32x4*)(buf + maxb / 2), (maxb / 2) / 16, sink);
            HIPC(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(nt ? stream_read<true> : stream_read<false>, dim3(512), dim3(256), 0, 0, (const i32x4*)buf, n16, sink);
            HIPC(hipEventRecord(e1, 0)); HIPC(hipEventSynchronize(e1));
            float cold = 0; HIPC(hipEventElapsedTime(&cold, e0, e1));
            HIPC(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(nt ? stream_read<true> : stream_read<false>, dim3(512), dim3(256), 0, 0, (const i32x4*)buf, n16, sink);
            HIPC(hipEventRecord(e1, 0)); HIPC(hipEventSynchronize(e1));
            float warm = 0; HIPC(hipEventElapsedTime(&warm, e0, e1));
            printf("%s loads, %5zu MiB: first pass %8.1f us (%6.0f GB/s)   repeated passes %8.1f us each (%6.0f GB/s)\n", nt ? "nt     " : "default", b >> 20,
                   cold * 1e3, b / (cold * 1e-3) / 1e9, warm * 1e3 / reps, b / (warm * 1e-3 / reps) / 1e9);
        }
    return 0;
}
