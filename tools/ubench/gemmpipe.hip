// gemmpipe.hip — PROTOTYPE (not product code; next step of DESIGN.md section 10 item 1).  gemmstage.hip shows the five stages of gemm_q8_lds_kernel's
// group iteration adding up (loads 16.0 + stores / barrier 0.4 + fragment reads 5.9 + MFMAs 2.4 + combine 7.2 = 31.9 us for the w1/w3 projection at 512 tokens):
// a register-staged prefetch ONE group ahead does not hide the L2 round trip, and two workgroups per CU leave no register room for more.  Here: the ring
// kernel of lm.rs_amd/csrc/lmrs_prefill.inc (gemm_q8_dma_kernel: LDS-DMA into swizzled 128-byte rows, one barrier per group, counted vmcnt) generalised to
// a WGM x WGN grid of waves, so that ONE workgroup of 8 waves per CU owns a 256 x 128 tile (every wave a 64 x 64 sub-tile, as in the register-staged
// kernel) with THREE ring slots of 50 KB: the prefetch runs two groups ahead without a register, and the fragment reads / MFMAs / combine of 8 waves fill
// each other's latencies.  Same arithmetic: a group's integer sums by two v_mfma_i32_16x16x64_i8 per 16 x 16 tile, the float combine
// ((isum as f32) * ws) * xs added in ascending group order per element - checked here against a host loop, bit for bit, on a ragged shape.
//   usage: gemmpipe            (self-check, then the w1/w3 shape at 512 / 2048 tokens: 2 x 2 waves with 128 x 128 tiles and 4 slots = the product's
//                               LMRS_GEMM_DMA=3 form, against 4 x 2 waves with 256 x 128 tiles and 3 slots)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int i32x4m __attribute__((ext_vector_type(4)));
typedef float f32x4m __attribute__((ext_vector_type(4)));
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

struct Args { const int8_t* wq; const int8_t* xq; const float* ws; const float* xs; float* out; int n, o, n_tok, store; };

// WM x WN MFMA tiles (16 x 16) per wave, WGM x WGN waves per workgroup
template <int WM, int WN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_pipe(const Args a, const int n_rt, const int n_tt) {
    constexpr int NW = WGM * WGN, TM = 16 * WM * WGM, TN = 16 * WN * WGN, ROWS = TM + TN;
    static_assert((ROWS / 8) % NW == 0, "whole wave-loads per wave");
    constexpr int NL = ROWS / 8 / NW;                            // 16-byte-per-lane DMA loads per wave and group (8 rows per wave-load)
    constexpr int SLOT = ROWS * 128 + ROWS * 4 + 256;            // rows, their group scales, a dump line for the waves without scales to fetch
    constexpr int S = (150 * 1024) / SLOT >= 8 ? 8 : (150 * 1024) / SLOT;
    static_assert(S >= 3 && (S - 2) * (NL + 1) <= 63 && ROWS <= 64 * NW, "ring depth vs the 6-bit vmcnt; one scale per lane");
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, kb = lane >> 4;
    const int K = a.n, G = K / 128;
    int rt, tt;
    {   // block -> (row tile, token tile): the 8 XCDs take the row tiles round robin, each XCD runs all token tiles of its row tiles
        const int b = blockIdx.x, x = b & 7, j = b >> 3, per = (n_rt + 7) / 8;
        rt = x + 8 * (j / n_tt); tt = j % n_tt;
        if (j / n_tt >= per || rt >= n_rt) return;
    }
    const int r_base = rt * TM, t_base = tt * TN;
    const int8_t* src[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        const int row = 8 * (wave * NL + q) + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
        if (row < TM) { int r = r_base + row; r = r < a.o ? r : a.o - 1; src[q] = a.wq + (size_t)r * K + c * 16; }
        else { int t = t_base + row - TM; t = t < a.n_tok ? t : a.n_tok - 1; src[q] = a.xq + (size_t)t * K + c * 16; }
    }
    const float* ssrc;
    {
        int i = wave * 64 + lane; i = i < ROWS ? i : ROWS - 1;
        if (i < TM) { int r = r_base + i; r = r < a.o ? r : a.o - 1; ssrc = a.ws + (size_t)r * G; }
        else { int t = t_base + i - TM; t = t < a.n_tok ? t : a.n_tok - 1; ssrc = a.xs + (size_t)t * G; }
    }
    auto issue = [&](int g, int sl) __attribute__((always_inline)) {
        char* base = ring + sl * SLOT;
#pragma unroll
        for (int q = 0; q < NL; ++q)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src[q] + (size_t)g * 128), (LDS_AS void*)(base + (wave * NL + q) * 1024), 16, 0, 0);
        const int sw = wave < ROWS / 64 ? wave : ROWS / 64;       // (surplus waves: the dump line - every wave issues the same number of loads)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ssrc + g), (LDS_AS void*)(base + ROWS * 128 + sw * 256), 4, 0, 0);
    };
    f32x4m acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[m][j] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < S - 1; ++g) issue(g < G ? g : G - 1, g);
    for (int g = 0; g < G; ++g) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * (NL + 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        { const int gn = g + S - 1; issue(gn < G ? gn : G - 1, gn % S); }
        const char* base = ring + (g % S) * SLOT;
        const char* As = base; const char* Bs = base + TM * 128;
        const float* sc = reinterpret_cast<const float*>(base + ROWS * 128);
        i32x4m a0[WM], a1[WM], b0[WN], b1[WN]; f32x4m wsv[WM]; float xsv[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int row = wm * 16 * WM + m * 16 + lr, sw = (row >> 1) & 7;
            a0[m] = *reinterpret_cast<const i32x4m*>(As + row * 128 + ((kb ^ sw) << 4));
            a1[m] = *reinterpret_cast<const i32x4m*>(As + row * 128 + (((kb + 4) ^ sw) << 4));
            wsv[m] = *reinterpret_cast<const f32x4m*>(sc + wm * 16 * WM + m * 16 + kb * 4);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int row = wn * 16 * WN + j * 16 + lr, sw = (row >> 1) & 7;
            b0[j] = *reinterpret_cast<const i32x4m*>(Bs + row * 128 + ((kb ^ sw) << 4));
            b1[j] = *reinterpret_cast<const i32x4m*>(Bs + row * 128 + (((kb + 4) ^ sw) << 4));
            xsv[j] = sc[TM + wn * 16 * WN + j * 16 + lr];
        }
        i32x4m cprev;
        {
            i32x4m c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[0], b0[0], c, 0, 0, 0);
            cprev = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[0], b1[0], c, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WM * WN; ++i) {
            const int m = i / WN, j = i % WN;
            i32x4m cnext = {0, 0, 0, 0};
            if (i + 1 < WM * WN) {
                const int m2 = (i + 1) / WN, j2 = (i + 1) % WN;
                cnext = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[m2], b0[j2], cnext, 0, 0, 0);
                cnext = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[m2], b1[j2], cnext, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p = (float)cprev[e] * wsv[m][e];
                p = p * xsv[j];
                acc[m][j][e] = acc[m][j][e] + p;
            }
            cprev = cnext;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float fs = 0.f;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int rq = r_base + wm * 16 * WM + m * 16 + kb * 4;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int t = t_base + wn * 16 * WN + j * 16 + lr;
            if (a.store) {
                if (rq < a.o && t < a.n_tok) *reinterpret_cast<f32x4m*>(a.out + (size_t)t * a.o + rq) = acc[m][j];     // (o is a multiple of 16)
            } else fs += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
        }
    }
    if (!a.store && fs == 12345.678f) a.out[0] = fs;              // (timing without the output stream: as gemmstage's stage 4)
}

template <int WM, int WN, int WGM, int WGN>
static void launch(const Args& a) {
    constexpr int TM = 16 * WM * WGM, TN = 16 * WN * WGN, ROWS = TM + TN, SLOT = ROWS * 128 + ROWS * 4 + 256, S = (150 * 1024) / SLOT >= 8 ? 8 : (150 * 1024) / SLOT;
    const int n_rt = (a.o + TM - 1) / TM, n_tt = (a.n_tok + TN - 1) / TN, per = (n_rt + 7) / 8;
    static bool once = false;
    if (!once) { HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pipe<WM, WN, WGM, WGN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
    hipLaunchKernelGGL((gemm_pipe<WM, WN, WGM, WGN>), dim3(8 * per * n_tt), dim3(64 * WGM * WGN), (size_t)S * SLOT, 0, a, n_rt, n_tt);
}

template <int WM, int WN, int WGM, int WGN>
static float time_us(const Args& a, int reps) {
    hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) launch<WM, WN, WGM, WGN>(a);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch<WM, WN, WGM, WGN>(a);
    HIPC(hipEventRecord(e1, 0)); HIPC(hipEventSynchronize(e1));
    float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
    HIPC(hipEventDestroy(e0)); HIPC(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

template <int WM, int WN, int WGM, int WGN>
static int self_check(const char* name) {
    // ragged on purpose: rows and tokens that are not multiples of the tile, a K of 5 groups (fewer / more than the ring slots)
    const int K = 640, o = 16 * 37, n_tok = 200, G = K / 128;
    std::vector<int8_t> wq((size_t)o * K), xq((size_t)n_tok * K); std::vector<float> ws((size_t)o * G), xs((size_t)n_tok * G), ref((size_t)n_tok * o), got((size_t)n_tok * o);
    for (auto& v : wq) v = (int8_t)((int)(rnd() % 255) - 127);
    for (auto& v : xq) v = (int8_t)((int)(rnd() % 255) - 127);
    for (auto& v : ws) v = (float)(rnd() % 1000 + 1) * 1.7e-4f;
    for (auto& v : xs) v = (float)(rnd() % 1000 + 1) * 3.1e-3f;
    for (int t = 0; t < n_tok; ++t)
        for (int r = 0; r < o; ++r) {
            float acc = 0.f;
            for (int g = 0; g < G; ++g) {
                int isum = 0;
                for (int k = 0; k < 128; ++k) isum += (int)wq[(size_t)r * K + g * 128 + k] * (int)xq[(size_t)t * K + g * 128 + k];
                float p = (float)isum * ws[(size_t)r * G + g];
                p = p * xs[(size_t)t * G + g];
                acc = acc + p;
            }
            ref[(size_t)t * o + r] = acc;
        }
    Args a{}; float* out;
    int8_t *dw, *dx; float *dws, *dxs;
    HIPC(hipMalloc(&dw, wq.size())); HIPC(hipMalloc(&dx, xq.size())); HIPC(hipMalloc(&dws, ws.size() * 4)); HIPC(hipMalloc(&dxs, xs.size() * 4)); HIPC(hipMalloc(&out, got.size() * 4));
    HIPC(hipMemcpy(dw, wq.data(), wq.size(), hipMemcpyHostToDevice)); HIPC(hipMemcpy(dx, xq.data(), xq.size(), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(dws, ws.data(), ws.size() * 4, hipMemcpyHostToDevice)); HIPC(hipMemcpy(dxs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
    HIPC(hipMemset(out, 0xff, got.size() * 4));
    a.wq = dw; a.xq = dx; a.ws = dws; a.xs = dxs; a.out = out; a.n = K; a.o = o; a.n_tok = n_tok; a.store = 1;
    launch<WM, WN, WGM, WGN>(a);
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
    printf("self-check %-34s %d x %d, %d tokens: %zu of %zu outputs differ from the host loop%s\n", name, K, o, n_tok, bad, got.size(), bad ? "  <-- MISMATCH" : " (bit-equal)");
    HIPC(hipFree(dw)); HIPC(hipFree(dx)); HIPC(hipFree(dws)); HIPC(hipFree(dxs)); HIPC(hipFree(out));
    return bad != 0;
}

int main() {
    HIPC(hipSetDevice(0));
    int fail = 0;
    fail |= self_check<4, 4, 2, 2>("2 x 2 waves, 128 x 128 tile, 4 slots");
    fail |= self_check<4, 4, 4, 2>("4 x 2 waves, 256 x 128 tile, 3 slots");
    fail |= self_check<4, 2, 2, 4>("2 x 4 waves, 128 x 128 tile, 4 slots");
    fail |= self_check<2, 2, 2, 2>("2 x 2 waves, 64 x 64 tile, 8 slots");
    const int reps = 20;
    struct Shape { int K, o, n_tok; const char* what; };
    const Shape shapes[] = {{2048, 16384, 512, "w1/w3 of Llama-3.2-1B, 512 tokens"}, {2048, 16384, 2048, "the same, 2048 tokens"}, {8192, 2048, 512, "w2 of Llama-3.2-1B, 512 tokens"},
                            {2048, 3072, 512, "qkv of Llama-3.2-1B, 512 tokens"}};
    for (const Shape& sh : shapes) {
        const int G = sh.K / 128;
        Args a{}; int8_t *wq, *xq; float *ws, *xs, *out;
        HIPC(hipMalloc(&wq, (size_t)sh.o * sh.K)); HIPC(hipMalloc(&xq, (size_t)sh.n_tok * sh.K));
        HIPC(hipMalloc(&ws, (size_t)sh.o * G * 4)); HIPC(hipMalloc(&xs, (size_t)sh.n_tok * G * 4)); HIPC(hipMalloc(&out, (size_t)sh.o * sh.n_tok * 4));
        HIPC(hipMemset(wq, 3, (size_t)sh.o * sh.K)); HIPC(hipMemset(xq, 5, (size_t)sh.n_tok * sh.K));
        HIPC(hipMemset(ws, 0x3c, (size_t)sh.o * G * 4)); HIPC(hipMemset(xs, 0x3c, (size_t)sh.n_tok * G * 4));
        a.wq = wq; a.xq = xq; a.ws = ws; a.xs = xs; a.out = out; a.n = sh.K; a.o = sh.o; a.n_tok = sh.n_tok;
        const double ops = 2.0 * sh.o * sh.n_tok * sh.K;
        printf("%s (K = %d, o = %d):\n", sh.what, sh.K, sh.o);
        for (int store = 0; store <= 1; ++store) {
            a.store = store;
            const float u0 = time_us<2, 2, 2, 2>(a, reps), u1 = time_us<4, 4, 2, 2>(a, reps), u2 = time_us<4, 4, 4, 2>(a, reps), u3 = time_us<4, 2, 2, 4>(a, reps);
            printf("  %s   64 x 64 / 4 waves / 8 slots %7.1f us (%4.0f TOP/s)   128 x 128 / 4 waves / 4 slots %7.1f us (%4.0f)   256 x 128 / 8 waves / 3 slots %7.1f us (%4.0f)"
                   "   128 x 128 / 8 waves / 4 slots %7.1f us (%4.0f)\n", store ? "with the output stores" : "without output stores ", u0, ops / u0 / 1e6, u1, ops / u1 / 1e6, u2, ops / u2 / 1e6, u3, ops / u3 / 1e6);
        }
        HIPC(hipFree(wq)); HIPC(hipFree(xq)); HIPC(hipFree(ws)); HIPC(hipFree(xs)); HIPC(hipFree(out));
    }
    return fail;
}
