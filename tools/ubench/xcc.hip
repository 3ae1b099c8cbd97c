// xcc.hip — which XCD does block b of a launch land on (grid sizes / LDS footprints the decode kernels use)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k(int* out, int spin) {
    extern __shared__ char smem[];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 15);
}
int main() {
    int* d; CK(hipMalloc(&d, 1 << 20));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Cfg { int grid, nt, lds, spin; } cfgs[] = {{256, 256, 0, 100}, {256, 256, 112 * 1024, 100}, {384, 256, 12 * 1024, 100}, {512, 256, 12 * 1024, 100}, {256, 512, 9 * 1024, 100}, {512, 256, 12 * 1024, 0}, {1024, 256, 0, 50}};
    for (auto c : cfgs) {
        std::vector<int> h(c.grid);
        int bad = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.nt), c.lds, 0, d, c.spin);
            CK(hipMemcpy(h.data(), d, c.grid * 4, hipMemcpyDeviceToHost));
            for (int b = 0; b < c.grid; ++b) bad += h[b] != b % 8;
        }
        printf("grid %4d x %3d threads, lds %6d, spin %d: blocks with xcc != b %% 8 over 20 launches: %d   first 16:", c.grid, c.nt, c.lds, c.spin, bad);
        for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
        printf("\n");
    }
    return 0;
}
