// gemmpipe.hip — micro-benchmark behind DESIGN.md section 4 (the batched matmul_q8, functional.rs:173-214 with sl > 1): the group iteration of the
// int8 matrix-core GEMM as a PIPELINE.  gemmstage.hip showed the stages of gemm_q8_lds_kernel adding up (loads 16.0 + fragment reads 5.9 + MFMAs 2.4 +
// combine 7.2 us for the w1/w3 projection at 512 tokens).  Two kernels here, same arithmetic (a group's integer sums by two v_mfma_i32_16x16x64_i8 per
// 16 x 16 tile, the float combine ((isum as f32) * ws) * xs added in ascending group order per element - checked against a host loop, bit for bit):
//   gemm_ring  the product's LDS-DMA ring kernel (lmrs_prefill.inc, round 3) generalised to a WGM x WGN grid of waves: per group  wait -> barrier -> issue
//              the DMA of group g + S - 1 -> read the fragments of group g -> MFMAs -> combine.
//   gemm_pipe  the same ring, but a wave reads the fragments of group g + 1 (token side: a second register set; weight side: one 16-row fragment ahead,
//              two small sets) while it multiplies group g: the LDS round trip leaves the dependency chain, and the DMA of group g + S - 1 has a whole
//              iteration to land.  Slot (g - 1) % S is the one refilled at the top of iteration g.
//   usage: gemmpipe [quick]
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

constexpr int kLds = 160 * 1024;
template <int WM, int WN, int WGM, int WGN>
struct Geo {
    static constexpr int NW = WGM * WGN, TM = 16 * WM * WGM, TN = 16 * WN * WGN, ROWS = TM + TN;
    static constexpr int NL = ROWS / 8 / NW;                     // 16-byte-per-lane DMA loads per wave and group (8 rows per wave-load)
    static constexpr int NSW = (ROWS + 63) / 64;                 // waves that fetch 64 group scales each
    static constexpr int SLOT = ROWS * 128 + NSW * 256 + 256;    // rows, their group scales, a dump line for the waves without scales to fetch
    static constexpr int S = kLds / SLOT >= 8 ? 8 : kLds / SLOT;
    static_assert((ROWS / 8) % NW == 0, "whole wave-loads per wave");
    static_assert(ROWS <= 64 * NW, "one scale per lane");
};

template <int WM, int WN, int WGM, int WGN>
struct Tile {
    using G_ = Geo<WM, WN, WGM, WGN>;
    int lane, wave, wm, wn, lr, kb, G, r_base, t_base;
    const int8_t* src[G_::NL];
    const float* ssrc;
    bool live;
    __device__ __forceinline__ Tile(const Args& a, int n_rt, int n_tt) {
        const int tid = threadIdx.x;
        lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6); wm = wave / WGN; wn = wave % WGN; lr = lane & 15; kb = lane >> 4;
        const int K = a.n; G = K / 128;
        // block -> (row tile, token tile): the 8 XCDs take the row tiles round robin, each XCD runs all token tiles of its row tiles
        const int b = blockIdx.x, x = b & 7, j = b >> 3, per = (n_rt + 7) / 8;
        const int rt = x + 8 * (j / n_tt), tt = j % n_tt;
        live = !(j / n_tt >= per || rt >= n_rt);
        r_base = rt * G_::TM; t_base = tt * G_::TN;
#pragma unroll
        for (int q = 0; q < G_::NL; ++q) {
            const int row = 8 * (wave * G_::NL + q) + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
            if (row < G_::TM) { int r = r_base + row; r = r < a.o ? r : a.o - 1; src[q] = a.wq + (size_t)r * K + c * 16; }
            else { int t = t_base + row - G_::TM; t = t < a.n_tok ? t : a.n_tok - 1; src[q] = a.xq + (size_t)t * K + c * 16; }
        }
        int i = wave * 64 + lane; i = i < G_::ROWS ? i : G_::ROWS - 1;
        if (i < G_::TM) { int r = r_base + i; r = r < a.o ? r : a.o - 1; ssrc = a.ws + (size_t)r * G; }
        else { int t = t_base + i - G_::TM; t = t < a.n_tok ? t : a.n_tok - 1; ssrc = a.xs + (size_t)t * G; }
    }
    __device__ __forceinline__ void issue(char* ring, int g, int sl) const {
        char* base = ring + sl * G_::SLOT;
#pragma unroll
        for (int q = 0; q < G_::NL; ++q)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src[q] + (size_t)g * 128), (LDS_AS void*)(base + (wave * G_::NL + q) * 1024), 16, 0, 0);
        const int sw = wave < G_::NSW ? wave : G_::NSW;           // (surplus waves: the dump line - every wave issues the same number of loads)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ssrc + g), (LDS_AS void*)(base + G_::ROWS * 128 + sw * 256), 4, 0, 0);
    }
    template <class ACC>
    __device__ __forceinline__ void finish(const Args& a, ACC& acc) const {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the surplus re-loads of the tail: nothing may still be writing LDS at exit)
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
        if (!a.store && fs == 12345.678f) a.out[0] = fs;          // (timing without the output stream)
    }
};

#define COMBINE(ACC, C, WS, XS)                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                 \
        float p = (float)(C)[e] * (WS)[e]; /* (ival as f32) * w.s[..] */                            \
        p = p * (XS);                      /*   * x.s[..]            */                             \
        (ACC)[e] = (ACC)[e] + p;           /* groups ascending       */                             \
    }

// ---- the round-3 ring kernel (baseline)
template <int WM, int WN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_ring(const Args a, const int n_rt, const int n_tt) {
    using G_ = Geo<WM, WN, WGM, WGN>;
    constexpr int S = G_::S, NL = G_::NL, TM = G_::TM, ROWS = G_::ROWS;
    static_assert(S >= 3 && (S - 2) * (NL + 1) <= 63, "ring depth vs the 6-bit vmcnt");
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const Tile<WM, WN, WGM, WGN> T(a, n_rt, n_tt);
    if (!T.live) return;
    const int G = T.G, lr = T.lr, kb = T.kb, wm = T.wm, wn = T.wn;
    f32x4m acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[m][j] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < S - 1; ++g) T.issue(ring, g < G ? g : G - 1, g);
    for (int g = 0; g < G; ++g) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * (NL + 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        { const int gn = g + S - 1; T.issue(ring, gn < G ? gn : G - 1, gn % S); }
        const char* base = ring + (g % S) * G_::SLOT;
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
            COMBINE(acc[m][j], cprev, wsv[m], xsv[j])
            cprev = cnext;
        }
    }
    T.finish(a, acc);
}

// ---- the pipelined kernel.  Fragment n = g * WM + m (16 weight rows of group g) lives in register set A[n & 1]; the token-side fragments of group g in
// B[g & 1].  Reads: B(g + 1) at the top of iteration g; A(n + 2) as soon as the last MFMA on A[n & 1] has been issued.  So when a wave arrives at the
// top of iteration g + 1 it already holds A(g + 1, 0), A(g + 1, 1) and B(g + 1): nothing of the LDS round trip is left between the barrier and the MFMAs.
// Slot of group g is read during iterations g - 1 and g; at the top of iteration g every wave is past iteration g - 1, slot (g - 1) % S is refilled.
// MODE: low nibble MI = MFMA pairs issued ahead of the combine they feed (1 = the next tile's, 2 = the next two tiles'); bit 4 = scheduling barriers that
// pin the order of fragment reads, MFMAs and combines to the one written here (without them the max-ilp scheduler rotates the loop its own way).
#define PIN() do { if constexpr (SB) __builtin_amdgcn_sched_barrier(0); } while (0)
template <int WM, int WN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_pipe(const Args a, const int n_rt, const int n_tt) {
    constexpr int MI = MODE & 15; constexpr bool SB = (MODE & 16) != 0;
    using G_ = Geo<WM, WN, WGM, WGN>;
    constexpr int S = G_::S, NL = G_::NL, TM = G_::TM, ROWS = G_::ROWS, NT = WM * WN;
    static_assert(S >= 3 && (S - 2) * (NL + 1) <= 63 && WM >= 2, "ring depth vs the 6-bit vmcnt; two weight fragments ahead stay inside group g + 1");
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const Tile<WM, WN, WGM, WGN> T(a, n_rt, n_tt);
    if (!T.live) return;
    const int G = T.G, lr = T.lr, kb = T.kb, wm = T.wm, wn = T.wn;
    struct ASet { i32x4m a0, a1; f32x4m ws; };
    struct BSet { i32x4m b0[WN], b1[WN]; float xs[WN]; };
    ASet A[2]; BSet B[2];
    // per-lane LDS offsets inside a slot (the swizzle depends on the row only)
    int aoff0[WM], aoff1[WM], boff0[WN], boff1[WN];
#pragma unroll
    for (int m = 0; m < WM; ++m) { const int row = wm * 16 * WM + m * 16 + lr, sw = (row >> 1) & 7; aoff0[m] = row * 128 + ((kb ^ sw) << 4); aoff1[m] = row * 128 + (((kb + 4) ^ sw) << 4); }
#pragma unroll
    for (int j = 0; j < WN; ++j) { const int row = wn * 16 * WN + j * 16 + lr, sw = (row >> 1) & 7; boff0[j] = TM * 128 + row * 128 + ((kb ^ sw) << 4); boff1[j] = TM * 128 + row * 128 + (((kb + 4) ^ sw) << 4); }
    auto readA = [&](ASet& s, int slot, int m) __attribute__((always_inline)) {
        const char* base = ring + slot * G_::SLOT;
        s.a0 = *reinterpret_cast<const i32x4m*>(base + aoff0[m]); s.a1 = *reinterpret_cast<const i32x4m*>(base + aoff1[m]);
        s.ws = *reinterpret_cast<const f32x4m*>(base + ROWS * 128 + (wm * 16 * WM + m * 16 + kb * 4) * 4);
    };
    auto readB = [&](BSet& s, int slot) __attribute__((always_inline)) {
        const char* base = ring + slot * G_::SLOT;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            s.b0[j] = *reinterpret_cast<const i32x4m*>(base + boff0[j]); s.b1[j] = *reinterpret_cast<const i32x4m*>(base + boff1[j]);
            s.xs[j] = *reinterpret_cast<const float*>(base + ROWS * 128 + (TM + wn * 16 * WN + j * 16 + lr) * 4);
        }
    };
    f32x4m acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[m][j] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < S - 1; ++g) T.issue(ring, g < G ? g : G - 1, g);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (NL + 1)) : "memory");      // group 0 has landed
    __builtin_amdgcn_s_barrier();
    readB(B[0], 0); readA(A[0], 0, 0); readA(A[1], 0, 1);
    int sl_cur = 0;                                                               // slot of group g
    for (int g0 = 0; g0 < G; g0 += 2) {                                           // (G is even: K is a multiple of 256)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const int g = g0 + gp;
            int sl_nxt = sl_cur + 1; sl_nxt = sl_nxt == S ? 0 : sl_nxt;
            int sl_prev = sl_cur - 1; sl_prev = sl_prev < 0 ? S - 1 : sl_prev;
            // group g + 1 has landed (this wave's share) when at most the S - 3 younger groups' loads are outstanding; lgkmcnt(0): this wave's reads
            // of slot g - 1 (and the fragments it is about to use) are complete
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 3) * (NL + 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            { const int gn = g + S - 1; T.issue(ring, gn < G ? gn : G - 1, sl_prev); }
            readB(B[(gp + 1) & 1], sl_nxt);
            PIN();
            const BSet& Bc = B[gp & 1];
            i32x4m cq[MI + 1];                                                    // MFMA results in flight (tile i, i + 1 .. i + MI)
#pragma unroll
            for (int u = 0; u < MI; ++u) {
                const int m2 = u / WN, j2 = u % WN, n2 = gp * WM + m2;
                i32x4m c = {0, 0, 0, 0};
                c = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[n2 & 1].a0, Bc.b0[j2], c, 0, 0, 0);
                cq[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[n2 & 1].a1, Bc.b1[j2], c, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int m = i / WN, j = i % WN, n = gp * WM + m;
                if (i + MI < NT) {
                    const int m2 = (i + MI) / WN, j2 = (i + MI) % WN, n2 = gp * WM + m2;
                    i32x4m c = {0, 0, 0, 0};
                    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[n2 & 1].a0, Bc.b0[j2], c, 0, 0, 0);
                    cq[MI] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[n2 & 1].a1, Bc.b1[j2], c, 0, 0, 0);
                }
                const f32x4m wsv = A[n & 1].ws;
                COMBINE(acc[m][j], cq[0], wsv, Bc.xs[j])
#pragma unroll
                for (int u = 0; u < MI; ++u) cq[u] = cq[u + 1];
                if (j == WN - 1) {                                                // row fragment n is done (its MFMAs were issued MI tiles ago, its scales used just now)
                    const int nn = n + 2, mg = nn / WM - gp, mm = nn % WM;        // fragment n + 2: group g + mg (0 or 1), row fragment mm
                    PIN();
                    readA(A[n & 1], mg ? sl_nxt : sl_cur, mm);
                    PIN();
                }
            }
            sl_cur = sl_nxt;
        }
    }
    T.finish(a, acc);
}


// ---- register-staged kernel (round 5).  Every LDS-DMA variant above lands on the same ~30-34 GB/s per CU (time = ingested bytes / 7.5 TB/s whatever the
// tile, the ring depth or the wave count: one 1-KiB global_load_lds_dwordx4 costs the CU ~70 cycles) while plain global_load_dwordx4 takes in 83 GB/s per CU
// (ingest.hip).  So: the next group's tile comes in through REGISTERS (issued at the top of the iteration, waited for and written to the other LDS slot at
// its end - the compute in between is the latency cover, and scheduling barriers keep the compiler from hoisting the wait), one barrier per group.  Group
// scales: SG groups of a row are one 64-byte piece of its scale array - fetched once per SG iterations into an LDS table [SG][rows] (double-buffered),
// instead of one 4-byte load per row and group, each on a cache line of its own.
// MODE: bit 4 = scheduling barriers between the three phases; bit 5 = two workgroups per CU (SG = 8, launch bound).
template <int WM, int WN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(64 * WGM * WGN, (MODE & 32) ? 2 * WGM * WGN / 4 : 1) void gemm_reg(const Args a, const int n_rt, const int n_tt) {
    constexpr int NW = WGM * WGN, NT = 64 * NW, TM = 16 * WM * WGM, TN = 16 * WN * WGN, ROWS = TM + TN, NLD = ROWS * 8 / NT;
    constexpr bool SB = (MODE & 16) != 0;
    constexpr int SG = (MODE & 32) ? 8 : 16;
    static_assert((ROWS * 8) % NT == 0 && ROWS <= NT, "whole 16-byte pieces per thread; one scale row per thread");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const slot0 = lds; float* const sct = reinterpret_cast<float*>(lds + 2 * ROWS * 128);     // [2][SG][ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WGN, wn = wave % WGN, lr = lane & 15, kb = lane >> 4;
    const int K = a.n, G = K / 128;
    int rt, tt;
    {
        const int b = blockIdx.x, x = b & 7, j = b >> 3, per = (n_rt + 7) / 8;
        rt = x + 8 * (j / n_tt); tt = j % n_tt;
        if (j / n_tt >= per || rt >= n_rt) return;
    }
    const int r_base = rt * TM, t_base = tt * TN;
    const int8_t* src[NLD]; int dst[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int f = tid + q * NT, row = f >> 3, c = f & 7;
        if (row < TM) { int r = r_base + row; r = r < a.o ? r : a.o - 1; src[q] = a.wq + (size_t)r * K + c * 16; }
        else { int t = t_base + row - TM; t = t < a.n_tok ? t : a.n_tok - 1; src[q] = a.xq + (size_t)t * K + c * 16; }
        dst[q] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }
    const float* ssrc = nullptr;
    if (tid < ROWS) {
        if (tid < TM) { int r = r_base + tid; r = r < a.o ? r : a.o - 1; ssrc = a.ws + (size_t)r * G; }
        else { int t = t_base + tid - TM; t = t < a.n_tok ? t : a.n_tok - 1; ssrc = a.xs + (size_t)t * G; }
    }
    i32x4m R[NLD]; float2 RS[SG / 2];
    auto gload = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) R[q] = *reinterpret_cast<const i32x4m*>(src[q] + (size_t)g * 128);
    };
    auto sload = [&](int g0) __attribute__((always_inline)) {      // groups g0 .. g0 + SG - 1 of this thread's row (G is even; pairs past the end re-read the last one)
        if (tid < ROWS) {
#pragma unroll
            for (int k = 0; k < SG / 2; ++k) { const int g = g0 + 2 * k < G ? g0 + 2 * k : G - 2; RS[k] = *reinterpret_cast<const float2*>(ssrc + g); }
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        char* base = slot0 + buf * (ROWS * 128);
#pragma unroll
        for (int q = 0; q < NLD; ++q) *reinterpret_cast<i32x4m*>(base + dst[q]) = R[q];
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        if (tid < ROWS) {
            float* t = sct + buf * (SG * ROWS) + tid;
#pragma unroll
            for (int k = 0; k < SG / 2; ++k) { t[(2 * k) * ROWS] = RS[k].x; t[(2 * k + 1) * ROWS] = RS[k].y; }
        }
    };
    int aoff[WM], boff[WN];
#pragma unroll
    for (int m = 0; m < WM; ++m) { const int row = wm * 16 * WM + m * 16 + lr, sw = (row >> 1) & 7; aoff[m] = row * 128 + ((kb ^ sw) << 4); }
#pragma unroll
    for (int j = 0; j < WN; ++j) { const int row = wn * 16 * WN + j * 16 + lr, sw = (row >> 1) & 7; boff[j] = TM * 128 + row * 128 + ((kb ^ sw) << 4); }
    f32x4m acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[m][j] = f32x4m{0.f, 0.f, 0.f, 0.f};
    gload(0); sload(0);
    lstore(0); sstore(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int g = 0; g < G; ++g) {
        const int buf = g & 1, gs = g % SG, sbuf = (g / SG) & 1;
        const bool snext = gs == SG - 1 && g + 1 < G;               // the next group opens a new block of scales (workgroup-uniform)
        gload(g + 1 < G ? g + 1 : G - 1);
        if (snext) sload(g + 1);
        PIN();
        const char* base = slot0 + buf * (ROWS * 128);
        const float* sc = sct + sbuf * (SG * ROWS) + gs * ROWS;
        i32x4m a0[WM], a1[WM], b0[WN], b1[WN]; f32x4m wsv[WM]; float xsv[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            a0[m] = *reinterpret_cast<const i32x4m*>(base + aoff[m]);
            a1[m] = *reinterpret_cast<const i32x4m*>(base + (aoff[m] ^ 64));
            wsv[m] = *reinterpret_cast<const f32x4m*>(sc + wm * 16 * WM + m * 16 + kb * 4);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            b0[j] = *reinterpret_cast<const i32x4m*>(base + boff[j]);
            b1[j] = *reinterpret_cast<const i32x4m*>(base + (boff[j] ^ 64));
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
            COMBINE(acc[m][j], cprev, wsv[m], xsv[j])
            cprev = cnext;
        }
        PIN();
        lstore(buf ^ 1);
        if (snext) sstore(sbuf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float fs = 0.f;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int rq = r_base + wm * 16 * WM + m * 16 + kb * 4;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int t = t_base + wn * 16 * WN + j * 16 + lr;
            if (a.store) {
                if (rq < a.o && t < a.n_tok) *reinterpret_cast<f32x4m*>(a.out + (size_t)t * a.o + rq) = acc[m][j];
            } else fs += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
        }
    }
    if (!a.store && fs == 12345.678f) a.out[0] = fs;
}

template <class K>
static void set_lds(K k) { HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds)); }

template <int WM, int WN, int WGM, int WGN, int MI>
static void launch(const Args& a) {
    using G_ = Geo<WM, WN, WGM, WGN>;
    const int n_rt = (a.o + G_::TM - 1) / G_::TM, n_tt = (a.n_tok + G_::TN - 1) / G_::TN, per = (n_rt + 7) / 8;
    static bool once = false;
    if constexpr (MI >= 64) {
        constexpr int MODE = MI - 64, ROWS = G_::TM + G_::TN, SG = (MODE & 32) ? 8 : 16;
        constexpr size_t smem = (size_t)2 * ROWS * 128 + (size_t)2 * SG * ROWS * 4;
        if (!once) { set_lds(gemm_reg<WM, WN, WGM, WGN, MODE>); once = true; }
        hipLaunchKernelGGL((gemm_reg<WM, WN, WGM, WGN, MODE>), dim3(8 * per * n_tt), dim3(64 * WGM * WGN), smem, 0, a, n_rt, n_tt);
    } else if constexpr (MI == 0) {
        if (!once) { set_lds(gemm_ring<WM, WN, WGM, WGN>); once = true; }
        hipLaunchKernelGGL((gemm_ring<WM, WN, WGM, WGN>), dim3(8 * per * n_tt), dim3(64 * WGM * WGN), (size_t)G_::S * G_::SLOT, 0, a, n_rt, n_tt);
    } else {
        if (!once) { set_lds(gemm_pipe<WM, WN, WGM, WGN, MI>); once = true; }
        hipLaunchKernelGGL((gemm_pipe<WM, WN, WGM, WGN, MI>), dim3(8 * per * n_tt), dim3(64 * WGM * WGN), (size_t)G_::S * G_::SLOT, 0, a, n_rt, n_tt);
    }
}

template <int WM, int WN, int WGM, int WGN, int MI>
static float time_us(const Args& a, int reps) {
    hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) launch<WM, WN, WGM, WGN, MI>(a);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch<WM, WN, WGM, WGN, MI>(a);
    HIPC(hipEventRecord(e1, 0)); HIPC(hipEventSynchronize(e1));
    float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
    HIPC(hipEventDestroy(e0)); HIPC(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

struct Problem {
    int K, o, n_tok, G;
    std::vector<int8_t> wq, xq; std::vector<float> ws, xs, ref;
    int8_t *dw, *dx; float *dws, *dxs, *out;
    Problem(int K_, int o_, int n_tok_, bool with_ref) : K(K_), o(o_), n_tok(n_tok_), G(K_ / 128) {
        wq.resize((size_t)o * K); xq.resize((size_t)n_tok * K); ws.resize((size_t)o * G); xs.resize((size_t)n_tok * G);
        for (auto& v : wq) v = (int8_t)((int)(rnd() % 255) - 127);
        for (auto& v : xq) v = (int8_t)((int)(rnd() % 255) - 127);
        for (auto& v : ws) v = (float)(rnd() % 1000 + 1) * 1.7e-4f;
        for (auto& v : xs) v = (float)(rnd() % 1000 + 1) * 3.1e-3f;
        if (with_ref) {
            ref.resize((size_t)n_tok * o);
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
        }
        HIPC(hipMalloc(&dw, wq.size())); HIPC(hipMalloc(&dx, xq.size())); HIPC(hipMalloc(&dws, ws.size() * 4)); HIPC(hipMalloc(&dxs, xs.size() * 4)); HIPC(hipMalloc(&out, (size_t)n_tok * o * 4));
        HIPC(hipMemcpy(dw, wq.data(), wq.size(), hipMemcpyHostToDevice)); HIPC(hipMemcpy(dx, xq.data(), xq.size(), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(dws, ws.data(), ws.size() * 4, hipMemcpyHostToDevice)); HIPC(hipMemcpy(dxs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
    }
    ~Problem() { hipFree(dw); hipFree(dx); hipFree(dws); hipFree(dxs); hipFree(out); }
    Args args(int store) const { Args a{}; a.wq = dw; a.xq = dx; a.ws = dws; a.xs = dxs; a.out = out; a.n = K; a.o = o; a.n_tok = n_tok; a.store = store; return a; }
};

template <int WM, int WN, int WGM, int WGN, int MI>
static int self_check(Problem& p, const char* name) {
    std::vector<float> got((size_t)p.n_tok * p.o);
    HIPC(hipMemset(p.out, 0xff, got.size() * 4));
    launch<WM, WN, WGM, WGN, MI>(p.args(1));
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(got.data(), p.out, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += memcmp(&got[i], &p.ref[i], 4) != 0;
    printf("self-check %-44s K %d, %d rows, %d tokens: %zu of %zu outputs differ from the host loop%s\n", name, p.K, p.o, p.n_tok, bad, got.size(), bad ? "  <-- MISMATCH" : " (bit-equal)");
    return bad != 0;
}

#define VARIANTS(X)                                                           \
    X(2, 2, 2, 2, 0, " 64 x  64, 4 waves, DMA ring (product)")                 \
    X(2, 2, 2, 2, 64, " 64 x  64, 4 waves, registers         ")                \
    X(2, 2, 2, 2, 80, " 64 x  64, 4 waves, registers, pinned ")                \
    X(2, 1, 2, 4, 80, " 64 x  64, 8 waves, registers, pinned ")                \
    X(2, 2, 2, 2, 112, " 64 x  64, 4 waves, regs, pinned, 2/CU")               \
    X(3, 2, 2, 2, 80, " 96 x  64, 4 waves, registers, pinned ")                \
    X(2, 2, 4, 2, 80, "128 x  64, 8 waves, registers, pinned ")                \
    X(4, 2, 2, 2, 80, "128 x  64, 4 waves, registers, pinned ")                \
    X(4, 2, 2, 2, 112, "128 x  64, 4 waves, regs, pinned, 2/CU")               \
    X(4, 4, 2, 2, 0, "128 x 128, 4 waves, DMA ring          ")                 \
    X(4, 4, 2, 2, 80, "128 x 128, 4 waves, registers, pinned ")                \
    X(4, 4, 2, 2, 112, "128 x 128, 4 waves, regs, pinned, 2/CU")               \
    X(2, 4, 4, 2, 64, "128 x 128, 8 waves, registers         ")                \
    X(2, 4, 4, 2, 80, "128 x 128, 8 waves, registers, pinned ")                \
    X(4, 4, 4, 2, 0, "256 x 128, 8 waves, DMA ring          ")                 \
    X(4, 4, 4, 2, 64, "256 x 128, 8 waves, registers         ")                \
    X(4, 4, 4, 2, 80, "256 x 128, 8 waves, registers, pinned ")                \
    X(4, 4, 2, 4, 80, "128 x 256, 8 waves, registers, pinned ")

int main(int argc, char** argv) {
    HIPC(hipSetDevice(0));
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    int fail = 0;
    {   // ragged on purpose: rows and tokens that are not multiples of the tile, K of 6 groups (fewer / more than the ring slots)
        Problem p(768, 16 * 37, 200, true);
#define X(WM, WN, WGM, WGN, MI, NAME) fail |= self_check<WM, WN, WGM, WGN, MI>(p, NAME);
        VARIANTS(X)
#undef X
        Problem p3(128 * 38, 96, 80, true);                        // 38 groups: the scale table is refilled (16 / 8 groups per block)
#define X(WM, WN, WGM, WGN, MI, NAME) fail |= self_check<WM, WN, WGM, WGN, MI>(p3, NAME);
        VARIANTS(X)
#undef X
        Problem p2(256, 256, 64, true);                            // the shortest K: two groups
#define X(WM, WN, WGM, WGN, MI, NAME) fail |= self_check<WM, WN, WGM, WGN, MI>(p2, NAME);
        VARIANTS(X)
#undef X
    }
    const int reps = quick ? 5 : 20;
    struct Shape { int K, o, n_tok; const char* what; };
    const Shape shapes[] = {{2048, 16384, 512, "w1/w3 of Llama-3.2-1B, 512 tokens"}, {2048, 16384, 256, "w1/w3, 256 tokens"}, {2048, 16384, 2048, "w1/w3, 2048 tokens"},
                            {8192, 2048, 512, "w2 of Llama-3.2-1B, 512 tokens"}, {8192, 2048, 256, "w2, 256 tokens"},
                            {2048, 3072, 512, "qkv of Llama-3.2-1B, 512 tokens"}, {2048, 3072, 256, "qkv, 256 tokens"},
                            {2048, 2048, 512, "wo of Llama-3.2-1B, 512 tokens"}, {2048, 2048, 256, "wo, 256 tokens"},
                            {1024, 4096, 1154, "CLIP fc1 (2 crops x 577 rows)"}, {4096, 1024, 1154, "CLIP fc2"}};
    for (const Shape& sh : shapes) {
        Problem p(sh.K, sh.o, sh.n_tok, false);
        const double ops = 2.0 * sh.o * sh.n_tok * sh.K;
        printf("%s (K = %d, %d rows):\n", sh.what, sh.K, sh.o);
        for (int store = 0; store <= 1; ++store) {
            const Args a = p.args(store);
#define X(WM, WN, WGM, WGN, MI, NAME) { using G_ = Geo<WM, WN, WGM, WGN>; const int wgs = ((sh.o + G_::TM - 1) / G_::TM) * ((sh.n_tok + G_::TN - 1) / G_::TN); \
            const float u = time_us<WM, WN, WGM, WGN, MI>(a, reps); printf("  %s %s %5d workgroups: %7.1f us  %5.0f TOP/s  %4.1f %%\n", store ? "store   " : "no store", NAME, wgs, u, ops / u / 1e6, ops / u / 1e6 / 39.44); }
            VARIANTS(X)
#undef X
        }
    }
    return fail;
}
