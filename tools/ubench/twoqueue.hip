// twoqueue.hip — can the kernel boundary of a dependent chain be hidden?  A chain of "link" kernels, each consuming a 2048-value
// vector that the previous link produced (every workgroup needs ALL of it: the all-to-all edge of a decode layer).
//   mode 0: one stream, hipGraph, dependency = the stream's kernel boundary (what the decode step does today)
//   mode 1: the links alternate between TWO streams (two hardware queues) inside one graph, with NO dependency between the
//           queues; the data itself carries the dependency: every value is stored as a 64-bit (value, tag) pair, consumers poll
//           L2 until the tag of the link they wait for shows up.  Link n+1 is dispatched (and could prefetch) while link n runs.
//   mode 2: as mode 1 but on one stream (polling cost with the boundary still there)
// build: hipcc --offload-arch=gfx950 -O3 -o twoqueue twoqueue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int N = 2048, WG = 256, NT = 256;

template <bool POLL, int WORK>
__global__ __launch_bounds__(NT) void link(const unsigned long long* in, unsigned long long* out, unsigned tag_in, unsigned tag_out, int* err) {
    __shared__ float red[NT];
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < N / NT; ++u) {
        const int i = u * NT + threadIdx.x;
        unsigned long long v = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (POLL) {
            int spins = 0;
            while ((unsigned)(v >> 32) != tag_in) {
                if (++spins > 20000 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { *err = 1; break; }   // sticky: a missed hand-off ends the whole run quickly
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        acc += __uint_as_float((unsigned)v);
    }
    // some dependent work (stands for the prologue + dot of a GEMV)
    for (int k = 0; k < WORK; ++k) acc = acc * 1.0001f + 0.5f;
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < N / WG) {
        const float r = red[threadIdx.x] + red[threadIdx.x + 64] * 0.5f;
        const unsigned long long o = ((unsigned long long)tag_out << 32) | __float_as_uint(r * 1e-3f);
        __hip_atomic_store(out + blockIdx.x * (N / WG) + threadIdx.x, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void gate(volatile int* flag, int* err) {
    int spins = 0;
    while (!*flag) { if (++spins > 4000000) { *err = 2; break; } __builtin_amdgcn_s_sleep(8); }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipSetDevice(0));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned long long* buf[2]; int* err;
    CK(hipMalloc(&buf[0], N * 8)); CK(hipMalloc(&buf[1], N * 8)); CK(hipMalloc(&err, 4));
    const int LINKS = 100;
    for (int work : {0, 400}) {
        {   // mode 3: two streams, eager launches behind a host-released gate (no graph: the two queues really run side by side)
            int* hflag; CK(hipHostMalloc(&hflag, 4, hipHostMallocMapped)); *hflag = 0;
            CK(hipMemset(buf[0], 0, N * 8)); CK(hipMemset(buf[1], 0, N * 8)); CK(hipMemset(err, 0, 4));
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eb));
            hipLaunchKernelGGL(gate, dim3(1), dim3(1), 0, sa, hflag, err);
            hipLaunchKernelGGL(gate, dim3(1), dim3(1), 0, sb, hflag, err);
            CK(hipEventRecord(e0, sa));
            for (int l = 0; l < LINKS; ++l) {
                hipStream_t s = (l & 1) ? sb : sa;
                if (work) hipLaunchKernelGGL((link<true, 400>), dim3(WG), dim3(NT), 0, s, buf[l & 1], buf[(l + 1) & 1], (unsigned)l, (unsigned)(l + 1), err);
                else hipLaunchKernelGGL((link<true, 0>), dim3(WG), dim3(NT), 0, s, buf[l & 1], buf[(l + 1) & 1], (unsigned)l, (unsigned)(l + 1), err);
            }
            CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0));
            CK(hipEventRecord(e1, sa));
            *hflag = 1;
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            unsigned long long last; CK(hipMemcpy(&last, buf[LINKS & 1], 8, hipMemcpyDeviceToHost));
            printf("work %3d  %-52s: %.2f us per link   (timeout flag %d, last tag %u)\n", work, "two streams EAGER behind a gate, tagged data", ms * 1000.0 / LINKS, herr, (unsigned)(last >> 32));
            CK(hipHostFree(hflag));
        }
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(buf[0], 0, N * 8)); CK(hipMemset(buf[1], 0, N * 8)); CK(hipMemset(err, 0, 4));
            hipGraph_t g; hipGraphExec_t ge;
            hipEvent_t fork, join; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
            // tags: run r, link l -> r * 1000 + l + 1; the first link of a run polls for the LAST tag of the previous run (written by the last link)
            // (a graph is replayed with the same tags, so every replay is preceded by a reset of the first input to "tag 1000 * 0 + 0": done by memset node)
            CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
            CK(hipMemsetAsync(buf[0], 0, N * 8, sa)); CK(hipMemsetAsync(buf[1], 0, N * 8, sa));
            if (mode == 1) { CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0)); }
            for (int l = 0; l < LINKS; ++l) {
                hipStream_t s = (mode == 1 && (l & 1)) ? sb : sa;
                const unsigned tin = l, tout = l + 1;           // link 0 polls for tag 0 = the memset
                if (mode == 0) { if (work) hipLaunchKernelGGL((link<false, 400>), dim3(WG), dim3(NT), 0, s, buf[l & 1], buf[(l + 1) & 1], tin, tout, err); else hipLaunchKernelGGL((link<false, 0>), dim3(WG), dim3(NT), 0, s, buf[l & 1], buf[(l + 1) & 1], tin, tout, err); }
                else { if (work) hipLaunchKernelGGL((link<true, 400>), dim3(WG), dim3(NT), 0, s, buf[l & 1], buf[(l + 1) & 1], tin, tout, err); else hipLaunchKernelGGL((link<true, 0>), dim3(WG), dim3(NT), 0, s, buf[l & 1], buf[(l + 1) & 1], tin, tout, err); }
            }
            if (mode == 1) { CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0)); }
            CK(hipStreamEndCapture(sa, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipGraphLaunch(ge, sa)); CK(hipStreamSynchronize(sa));
            CK(hipEventRecord(e0, sa));
            const int REP = 5;
            for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, sa));
            CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            unsigned long long last; CK(hipMemcpy(&last, buf[LINKS & 1], 8, hipMemcpyDeviceToHost));
            const char* nm[] = {"one stream, kernel boundary", "two queues, tagged data (no boundary between links)", "one stream, tagged data + boundary"};
            printf("work %3d  %-52s: %.2f us per link   (timeout flag %d, last tag %u)\n", work, nm[mode], ms * 1000.0 / (REP * LINKS), herr, (unsigned)(last >> 32));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
