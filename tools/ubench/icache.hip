// icache.hip — is a kernel's instruction stream cold at every launch, and what does straight-line code cost the first time it runs?
//
// The decode kernels execute their prologue ONCE per launch, fully unrolled (a 256-add serial chain, ~250 VALU of quantiser): if the
// instruction cache is invalidated at each dispatch (acquire fence), every line of that code is a miss the first time.  Here the SAME
// straight-line body runs R times inside one launch (rep 0 = first touch, rep 1.. = warm), in wave 0 of workgroup 0 and of the last
// workgroup, for bodies of 1..16 KB, dependent chains and independent streams, the launch repeated (same kernel back to back, and
// with a different 16 KB kernel in between).  Variants: the other waves of the workgroup run the body ahead with EXEC = 0 (does a
// lane-less pass warm the cache, and does it cost the real wave anything?); the body as a loop of 32 adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int R = 4;

template <int NADD, int ILP>
__device__ __forceinline__ void body(float (&p)[4], float a) {
#pragma unroll
    for (int u = 0; u < NADD / ILP; ++u) {
#pragma unroll
        for (int c = 0; c < ILP; ++c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(a));
    }
}

// MODE 0: wave 0 runs the body, the others wait at the barrier.  1: the others run it too, with EXEC = 0 (from the kernel's start).
// 2: the others run it for real (all lanes).  3: as 0, body = loop of 32 adds.
template <int NADD, int ILP, int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, long long* t) {
    float p[4] = {in[threadIdx.x], in[threadIdx.x + 256], in[threadIdx.x + 512], in[threadIdx.x + 768]};
    const float a = in[1024 + (threadIdx.x & 63)];
    const int wave = threadIdx.x >> 6;
    long long* tw = t + ((size_t)blockIdx.x * 4 + wave) * (2 * R);
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        if (wave == 0 || MODE == 2) {
            const long long c0 = __builtin_readcyclecounter();
            if (MODE == 3) {
#pragma unroll 1
                for (int it = 0; it < NADD / 32; ++it) body<32, ILP>(p, a);
            } else body<NADD, ILP>(p, a);
            const long long c1 = __builtin_readcyclecounter();
            if ((threadIdx.x & 63) == 0) { tw[2 * r] = c1 - c0; tw[2 * r + 1] = c0; }
        } else if (MODE == 1) {
            unsigned long long saved;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0" : "=s"(saved));
            body<NADD, ILP>(p, a);
            asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
        }
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p[0] + p[1] + p[2] + p[3];
}

// a different kernel of ~16 KB of code, to sit between two launches of the kernel under test
__global__ __launch_bounds__(256) void other(float* out, const float* in) {
    float p[4] = {in[threadIdx.x], 0.f, 0.f, 0.f};
    const float a = in[1024 + (threadIdx.x & 63)];
    asm volatile("v_add_f32 %0, %0, %0" : "+v"(p[1]));
    body<4096, 1>(p, a);
    out[blockIdx.x * blockDim.x + threadIdx.x] = p[0] + p[1];
}

template <int NADD, int ILP, int MODE>
static void run(const char* name, int grid, float* out, float* in, long long* t, bool between) {
    const size_t words = (size_t)grid * 4 * 2 * R;
    long long* h = (long long*)malloc(words * 8);
    for (int launch = 0; launch < 3; ++launch) {
        CK(hipMemset(t, 0, words * 8));
        if (between) hipLaunchKernelGGL(other, dim3(grid), dim3(256), 0, 0, out, in);
        hipLaunchKernelGGL((k<NADD, ILP, MODE>), dim3(grid), dim3(256), 0, 0, out, in, t);
        CK(hipMemcpy(h, t, words * 8, hipMemcpyDeviceToHost));
        const long long* w0 = h;                                     // workgroup 0, wave 0
        const long long* wl = h + (size_t)(grid - 1) * 4 * 2 * R;    // last workgroup, wave 0
        // median of rep 0 over all workgroups' wave 0
        long long s0 = 0, s1 = 0;
        for (int b = 0; b < grid; ++b) { s0 += h[(size_t)b * 4 * 2 * R]; s1 += h[(size_t)b * 4 * 2 * R + 2]; }
        printf("%-58s %dB launch %d%s: wg0 reps %6lld %6lld %6lld %6lld | last wg %6lld %6lld %6lld %6lld (started %+lld cyc) | mean rep0 %6lld rep1 %6lld\n", name,
               NADD * 4, launch, between ? " (other kernel first)" : "", w0[0], w0[2], w0[4], w0[6], wl[0], wl[2], wl[4], wl[6], wl[1] - w0[1], s0 / grid, s1 / grid);
    }
    free(h);
}

int main() {
    float *in, *out; long long* t;
    CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&t, 1 << 20));
    CK(hipMemset(in, 0, 1 << 16));
    for (int grid : {256, 512}) {
        printf("grid %d x 256 threads (cycles of the shader clock counter per pass over the body)\n", grid);
        run<256, 1, 0>("dependent chain, wave 0 alone", grid, out, in, t, false);
        run<1024, 1, 0>("dependent chain, wave 0 alone", grid, out, in, t, false);
        run<4096, 1, 0>("dependent chain, wave 0 alone", grid, out, in, t, false);
        run<1024, 1, 0>("dependent chain, wave 0 alone", grid, out, in, t, true);
        run<1024, 4, 0>("4 independent chains, wave 0 alone", grid, out, in, t, false);
        run<4096, 4, 0>("4 independent chains, wave 0 alone", grid, out, in, t, false);
        run<1024, 4, 2>("4 independent chains, all four waves", grid, out, in, t, false);
        run<1024, 1, 1>("dependent chain, waves 1-3 run it with EXEC = 0", grid, out, in, t, false);
        run<1024, 4, 1>("4 independent chains, waves 1-3 run it with EXEC = 0", grid, out, in, t, false);
        run<1024, 1, 3>("dependent chain as a loop of 32 adds", grid, out, in, t, false);
        run<1024, 4, 3>("4 independent chains as a loop of 32 adds", grid, out, in, t, false);
    }
    return 0;
}
