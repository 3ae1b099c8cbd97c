// engine.hip — the WHOLE decode layer of Llama-3.2-1B Q8_0 as ONE persistent launch (go / no-go micro-benchmark, round 6).
//
// What the chip guide measured as 0.87-0.89 x of five launches for this very layer in bf16 (MI355X_MICROARCH.md, row engine-vs-launches),
// built here with the reference's arithmetic (transformer.rs:388-657, functional.rs:48-78,122-140,173-214, quantization.rs:44-67) on the
// real shapes: 256 workgroups (one per CU) x (1 loader wave + 3 consumer waves), 16 layers in the loop, position ~24.
//   * loader wave: streams this CU's share of every weight matrix - a fixed row slice of qkv / wo / w1|w3 / w2, repacked on the host in the
//     order the consumers read it - through a ring of 14 x (8 KiB int8 + 256 B scales) LDS slots by LDS-DMA (global_load_lds_dwordx4,
//     non-temporal), running ahead of the dependency stalls; thinned to ONE outstanding slot while its CU gathers.
//   * consumer waves: a slot is one JOB of 64 quantisation groups (4 rows x 2048 or 1 row x 8192): 8 steps of ds_read_b128 + 4 v_dot4 +
//     DPP butterfly + ordered float combine (the product's arithmetic), the activation held in registers for the whole stage.
//   * all-to-all edges as 8-byte {value, tag} granules (one write-through store each), swept by the consumer waves:
//       x (2048 f32, 16 KB)  -> every CU: RMSNorm chain + quantise by ONE wave per CU (bit-exact: 8 strided chains, reduce_add8)
//       q / k / v (3072)     -> the 16 attention CUs (two heads = one 128-value quantisation group per CU, one wave per head)
//       att_out              -> quantised AT THE PRODUCER (the group is local to the pair): 528 granules = 4 KB instead of 16 KB
//       h (8192)             -> two hops: per-CU max -> the 4 CUs of a group -> quantised {4 x int8} granules, 2112 granules = 16.5 KB
//   * self-check: x after 16 layers against a host loop of the same arithmetic, bit for bit, for CHECK positions.
// Output: us per step / per layer and a stamp table (gather / prologue / rows / attention per stage) of an attention CU and a plain one.
//
// Build: make -C tools/ubench engine.   Run: tools/ubench/engine [steps] [thin=1|0] [depth 2..6] [debug=0|1] [consumer waves 3|7]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../lm.rs_amd/csrc/lmrs_stage.h"
using namespace lmrs;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int D = 2048, HS = 64, NH = 32, NKV = 8, KVD = NKV * HS, QKV = D + 2 * KVD, HID = 8192, NL = 16, NCU = 256;
constexpr int JQ = QKV / 4 / NCU, JO = D / 4 / NCU, J13 = 2 * HID / 4 / NCU, J2 = D / NCU;       // jobs per CU and layer: 3, 2, 16, 8
constexpr int SLOTS = JQ + JO + J13 + J2, SLOTB = 8192 + 256, RING = 14;                          // 29 slots of 8448 bytes per CU and layer
constexpr int TW = 32;                                                                           // longest context of this benchmark
constexpr int NATT = NH / 2;                                                                     // attention CUs (two heads each)
constexpr float EPS = 1e-5f;
constexpr unsigned SPIN_MAX = 1u << 21;
constexpr int NST = 16, NSTX = 96;                                                               // stamps per (layer, wave): 16 stage stamps + 9 jobs x 8

struct EArgs {
    const char* stream; const float* norms; const float2* rope; float* kc; float* vc; const float* x_in; float* x_out;
    u64* xg; u64* qkvg; u64* attg; u64* hmaxg; u64* hqg; int* err; long long* stamps;
    int pos, S; unsigned base; int thin, depth, nl; float* dbg;
};
__host__ __device__ constexpr unsigned tag_of(unsigned base, int layer, int edge) { return base + (unsigned)layer * 8u + (unsigned)edge + 1u; }

// ---------------------------------------------------------------------------------------------------------------- LDS map
struct Ctl { unsigned full[16], freeq[16], bar, gath, pairbar, issued; float pairmax[2]; float ss, pad; float hraw[64]; };
constexpr int JP = 328;                                            // row pitch of the RMS squares (8 rows; skewed: see prologue)
constexpr size_t OFF_RING = 0, OFF_XQ = OFF_RING + (size_t)RING * SLOTB, OFF_XS = OFF_XQ + HID, OFF_XF = OFF_XS + 256, OFF_SQ = OFF_XF + D * 4,
                 OFF_ATT = OFF_SQ + 8 * JP * 4 + 64, OFF_CTL = OFF_ATT + 2 * 1024, SMEM = OFF_CTL + sizeof(Ctl);
static_assert(SMEM <= 160 * 1024, "LDS");

__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool give_up(unsigned& spins, int* err, int code) {
    if (++spins < SPIN_MAX && (spins & 4095u) != 4095u) return false;
    const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e != 0) return true;
    if (spins >= SPIN_MAX) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    return false;
}
// wait until an LDS word reaches a value (one lane's read is enough: the branch is wave-uniform)
__device__ __forceinline__ void lds_wait_ge(const unsigned* p, unsigned v, int* err, int code) {
    unsigned spins = 0;
    while (lds_ld(p) < v) { __builtin_amdgcn_s_sleep(1); if (give_up(spins, err, code)) break; }
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------- loader wave
__device__ __forceinline__ void loader(const EArgs& a, char* smem, Ctl* ctl, int cu, int lane) {
    const char* src = a.stream + (size_t)cu * NL * SLOTS * SLOTB;
    const int total = a.nl * SLOTS;
    int published = 0;                                              // slots [0, published) are marked full
    for (int s = 0; s < total; ++s) {
        const int ring = s % RING; const unsigned round = (unsigned)(s / RING);
        if (lds_ld(&ctl->freeq[ring]) < round) {                     // ring full: everything issued has time to land - mark it before waiting
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) for (int p = published; p < s; ++p) lds_st(&ctl->full[p % RING], (unsigned)(p / RING) + 1u);
            if (s > published) published = s;
            lds_wait_ge(&ctl->freeq[ring], round, a.err, 100);
        }
        char* dst = smem + OFF_RING + (size_t)ring * SLOTB;
        const char* g = src + (size_t)s * SLOTB;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_global_load_lds((const LMRS_GLOBAL void*)(g + u * 1024 + lane * 16), (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const LMRS_GLOBAL void*)(g + 8192 + lane * 4), (__attribute__((address_space(3))) void*)(dst + 8192), 4, 0, 2);
        // slots in flight behind this one: depth - 1 normally, ONE while this CU gathers (its sweeps queue behind the loader's requests)
        if (lane == 0) lds_st(&ctl->issued, (unsigned)(s + 1));
        const bool thin = a.thin && lds_ld(&ctl->gath) != 0;
        int keep;
        if (thin || a.depth <= 2) { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); keep = 1; }
        else if (a.depth == 3) { asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); keep = 2; }
        else if (a.depth == 4) { asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); keep = 3; }
        else { asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); keep = 5; }
        const int landed = s + 1 - keep;                            // slots [0, landed) have landed
        if (lane == 0) for (int p = published; p < landed; ++p) lds_st(&ctl->full[p % RING], (unsigned)(p / RING) + 1u);
        if (landed > published) published = landed;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) for (int p = published; p < total; ++p) lds_st(&ctl->full[p % RING], (unsigned)(p / RING) + 1u);
}

// ---------------------------------------------------------------------------------------------------------------- consumer pieces
struct Act { i32x4 xa[8]; float xs[8]; };                          // this lane's share of the quantised activation, for every job of a stage
template <int L> __device__ __forceinline__ void load_act(Act& A, const char* smem, int lane) {
    const int cl = (lane % L) / 8, rc = lane & 7;
    const int8_t* xq = reinterpret_cast<const int8_t*>(smem + OFF_XQ); const float* xs = reinterpret_cast<const float*>(smem + OFF_XS);
#pragma unroll
    for (int u = 0; u < 8; ++u) { A.xa[u] = *reinterpret_cast<const i32x4*>(xq + (cl * 8 + u) * 128 + rc * 16); A.xs[u] = xs[cl * 8 + u]; }
}
// one job = one ring slot: -> the rows' sums (functional.rs:173-214: groups ascending), valid in the last cluster of every row of L lanes
template <int L> __device__ __forceinline__ float run_job(const EArgs& a, char* smem, Ctl* ctl, int s, const Act& A, int lane, long long* js = nullptr) {
    const int ring = s % RING; const unsigned round = (unsigned)(s / RING);
    if (js) { js[0] = wall_clock64(); js[4] = (long long)lds_ld(&ctl->issued) - s; }
    lds_wait_ge(&ctl->full[ring], round + 1u, a.err, 200);
    if (js) js[1] = wall_clock64();
    const char* slot = smem + OFF_RING + (size_t)ring * SLOTB;
    i32x4 w[8]; float sc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const i32x4*>(slot + u * 1024 + lane * 16);
    {
        const float4 s0 = *reinterpret_cast<const float4*>(slot + 8192 + (lane >> 3) * 32), s1 = *reinterpret_cast<const float4*>(slot + 8192 + (lane >> 3) * 32 + 16);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    }
    float pb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int d = __builtin_amdgcn_sdot4(w[u].x, A.xa[u].x, 0, false);
        d = __builtin_amdgcn_sdot4(w[u].y, A.xa[u].y, d, false);
        d = __builtin_amdgcn_sdot4(w[u].z, A.xa[u].z, d, false);
        d = __builtin_amdgcn_sdot4(w[u].w, A.xa[u].w, d, false);
        d = cluster8_sum(d);
        const float p = (float)d * sc[u];
        pb[u] = p * A.xs[u];
        if (u == 7) { lds_drain(); if (lane == 0) lds_st(&ctl->freeq[ring], round + 1u); }          // the slot is in registers: hand it back
    }
    if (js) js[2] = wall_clock64();
    constexpr int NC = L / 8;
    const int cl = (lane % L) / 8;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        float carry = 0.0f;
        if (j > 0) carry = cluster_carry<L>(acc, j);
        if (cl == j) {
            acc = carry;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + pb[u];
        }
    }
    if (js) js[3] = wall_clock64();
    return acc;
}
template <int NCW> __device__ __forceinline__ void cbar(Ctl* ctl, unsigned& phase, int* err, int lane) {  // the NCW consumer waves
    lds_drain();
    phase += NCW;
    if (lane == 0) __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_wait_ge(&ctl->bar, phase, err, 300);
}
__device__ __forceinline__ void gath_mark(Ctl* ctl, int lane, int d) {
    if (lane == 0) __hip_atomic_fetch_add(&ctl->gath, (unsigned)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// sweep NG granules per lane (granule k of this lane: g[first + 64 k + lane], those past `count` are not waited for) until every tag is `tag`
template <int NG> __device__ __forceinline__ void sweep(const u64* g, int first, int count, unsigned tag, unsigned (&val)[NG], int* err, int code, int lane) {
    const LMRS_GLOBAL u64* p[NG]; bool live[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) { const int i = 64 * k + lane; live[k] = i < count; p[k] = (const LMRS_GLOBAL u64*)g + first + (live[k] ? i : 0); }
    unsigned spins = 0;
    for (;;) {
        u64 x[NG]; unsigned bad = 0u;                               // (branch-free: a short-circuit && here compiles to a ladder of exec-masked blocks)
#pragma unroll
        for (int k = 0; k < NG; ++k) x[k] = __hip_atomic_load(p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < NG; ++k) { val[k] = (unsigned)x[k]; bad |= live[k] ? ((unsigned)(x[k] >> 32) ^ tag) : 0u; }
        if (__all(bad == 0u)) break;
        if (give_up(spins, err, code)) break;
    }
}
__device__ __forceinline__ void put_gran(u64* g, int i, unsigned value, unsigned tag) {
    __hip_atomic_store((LMRS_GLOBAL u64*)g + i, ((u64)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int quant_one(float v, float m, float& sc_out) {                          // quantization.rs:44-67 for one value of a group with maximum m
    const bool sane = quant_group_sane(m);
    float mm = m; asm volatile("" : "+v"(mm));
    const float sc = sane ? div127_sane(m) : mm / 127.0f;
    float d; int q = quant_q8_cand(v, __builtin_amdgcn_rcpf(sc), d);
    if (!sane || fabsf(d) > kQuantDevMax) q = quant_q8(v, sc);
    sc_out = sc; return q;
}
__device__ __forceinline__ unsigned pack_quad(int q, int lane) {                                      // the four int8 of lanes 4i .. 4i+3, in every lane of the quad
    int v = (q & 0xff) << (8 * (lane & 3));
    v += dpp_i<0xB1>(v); v += dpp_i<0x4E>(v);
    return (unsigned)v;
}

// RMSNorm of the 2048 floats in LDS (functional.rs:48-78).  The gathering waves leave the squares in `sq` beside the values: chain k = e & 7
// takes x[8 j + k] in ascending j; row k of sq, position j + 4 (j >> 4) (the skew spreads consecutive 16-value blocks over the banks).
__device__ __forceinline__ int sq_at(int e) { const int j = e >> 3; return (e & 7) * JP + j + 4 * (j >> 4); }
__device__ __forceinline__ float rms_chain(char* smem, int lane) {                                   // one wave -> 1 / sqrt(mean square + eps), in every lane
    const float* sq = reinterpret_cast<const float*>(smem + OFF_SQ);
    float p = 0.0f;
    const int cl = lane & 15;
    const float* row = sq + (cl & 7) * JP + 4 * (cl >> 3);
    float4 A[4], B[4];
    auto rd = [&](float4 (&X)[4], int m0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int m = m0 + u; X[u] = *reinterpret_cast<const float4*>(row + 8 * m + 4 * (m >> 1)); }
    };
    rd(A, 0);
#pragma unroll
    for (int m0 = 0; m0 < D / 64; m0 += 8) {                          // a chain is D / 8 values = D / 64 steps of (4 own + 4 of the lane 8 above)
        rd(B, m0 + 4);
        asm volatile("" ::: "memory");
        rms_chain32(p, A);
        if (m0 + 8 < D / 64) rd(A, m0 + 8);
        asm volatile("" ::: "memory");
        rms_chain32(p, B);
    }
    const int pi = __float_as_int(p);
    const float p0 = __int_as_float(__builtin_amdgcn_readlane(pi, 0)), p1 = __int_as_float(__builtin_amdgcn_readlane(pi, 1));
    const float p2 = __int_as_float(__builtin_amdgcn_readlane(pi, 2)), p3 = __int_as_float(__builtin_amdgcn_readlane(pi, 3));
    const float p4 = __int_as_float(__builtin_amdgcn_readlane(pi, 4)), p5 = __int_as_float(__builtin_amdgcn_readlane(pi, 5));
    const float p6 = __int_as_float(__builtin_amdgcn_readlane(pi, 6)), p7 = __int_as_float(__builtin_amdgcn_readlane(pi, 7));
    float ss = reduce_add8(p0, p1, p2, p3, p4, p5, p6, p7);
    ss = ss * (1.0f / (float)D); ss = ss + EPS; ss = 1.0f / sqrtf(ss);
    return ss;
}
// normalise + quantise (quantization.rs:44-67) by NQ waves: wave wq owns 16 / NQ groups, 4 NQ lanes per group, 8 / NQ float4 per lane
template <int NQ> struct QGeom { static constexpr int LG = 4 * NQ, NF = 8 / NQ;
    __device__ static __forceinline__ int elem(int wq, int lane, int i) { return (wq * (16 / NQ) + lane / LG) * 128 + (lane % LG) * 4 + LG * 4 * i; } };
template <int NQ> __device__ __forceinline__ void norm_quant(char* smem, int wq, const float4 (&nw)[(QGeom<NQ>::NF)], float ss, int lane) {
    using Q = QGeom<NQ>;
    const float* xf = reinterpret_cast<const float*>(smem + OFF_XF);
    int8_t* xq = reinterpret_cast<int8_t*>(smem + OFF_XQ); float* xs = reinterpret_cast<float*>(smem + OFF_XS);
    float4 v[Q::NF];
    float mg = 0.0f;
#pragma unroll
    for (int i = 0; i < Q::NF; ++i) {
        v[i] = *reinterpret_cast<const float4*>(xf + Q::elem(wq, lane, i));
        v[i].x = nw[i].x * (ss * v[i].x); v[i].y = nw[i].y * (ss * v[i].y); v[i].z = nw[i].z * (ss * v[i].z); v[i].w = nw[i].w * (ss * v[i].w);
        mg = absmax4(v[i], mg);
    }
    mg = cluster_max<Q::LG>(mg);
    const bool sane = quant_group_sane(mg);
    float mm = mg; asm volatile("" : "+v"(mm));
    const float sc = sane ? div127_sane(mg) : mm / 127.0f, inv = __builtin_amdgcn_rcpf(sc);
#pragma unroll
    for (int i = 0; i < Q::NF; ++i) {
        float4 d; int q[4];
        q[0] = quant_q8_cand(v[i].x, inv, d.x); q[1] = quant_q8_cand(v[i].y, inv, d.y); q[2] = quant_q8_cand(v[i].z, inv, d.z); q[3] = quant_q8_cand(v[i].w, inv, d.w);
        if (__any(!sane || absmax4(d, 0.0f) > kQuantDevMax)) {
            if (!sane || fabsf(d.x) > kQuantDevMax) q[0] = quant_q8(v[i].x, sc);
            if (!sane || fabsf(d.y) > kQuantDevMax) q[1] = quant_q8(v[i].y, sc);
            if (!sane || fabsf(d.z) > kQuantDevMax) q[2] = quant_q8(v[i].z, sc);
            if (!sane || fabsf(d.w) > kQuantDevMax) q[3] = quant_q8(v[i].w, sc);
        }
        *reinterpret_cast<unsigned*>(xq + Q::elem(wq, lane, i)) = (unsigned)(q[0] & 0xff) | ((unsigned)(q[1] & 0xff) << 8) | ((unsigned)(q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24);
    }
    if ((lane % Q::LG) == 0) xs[wq * (16 / NQ) + lane / Q::LG] = sc;
}

// one head of one attention CU (transformer.rs:443-544): RoPE, K / V row of this position, scores, softmax, weighted values, then the pair's
// 128 outputs quantised HERE (the two waves of the CU hold one whole quantisation group) and published as {4 x int8, tag} granules.
__device__ __forceinline__ void attention(const EArgs& a, char* smem, Ctl* ctl, int layer, int cu, int cw, int lane, unsigned& pairphase, long long* st) {
    const int h = 2 * cu + cw, kvh = h / (NH / NKV), pos = a.pos, T = pos + 1, S = a.S;
    float* qs = reinterpret_cast<float*>(smem + OFF_ATT + cw * 1024); float* ks = qs + 64;
    const uint64_t etab = exp2f_tab_lane();
    const float2 cs = a.rope[(size_t)pos * (HS / 2) + (lane & 31)];
    float* kT = a.kc + ((size_t)layer * NKV + kvh) * HS * S;          // [HS / 4][S][4]
    float* vb = a.vc + (size_t)layer * S * KVD + kvh * HS;            // [S][KVD]
    // history: this lane's key (lane = position) and the value rows (lane = dim), before the poll
    float4 kk[16];
    const int tk = lane < pos ? lane : 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) kk[g] = *reinterpret_cast<const float4*>(kT + ((size_t)g * S + tk) * 4);
    float v[TW];
#pragma unroll
    for (int t0 = 0; t0 < TW; t0 += 8) {
        if (t0 < pos) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[t0 + u] = vb[(size_t)(t0 + u < pos ? t0 + u : 0) * KVD + lane];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[t0 + u] = 0.0f;
        }
    }
    unsigned g3[3]; u64 xq3[3];
    {
        const LMRS_GLOBAL u64* p0 = (const LMRS_GLOBAL u64*)a.qkvg + h * HS + lane, * p1 = (const LMRS_GLOBAL u64*)a.qkvg + D + kvh * HS + lane, * p2 = (const LMRS_GLOBAL u64*)a.qkvg + D + KVD + kvh * HS + lane;
        const unsigned tag = tag_of(a.base, layer, 1);
        unsigned spins = 0;
        for (;;) {
            xq3[0] = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); xq3[1] = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xq3[2] = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned bad = ((unsigned)(xq3[0] >> 32) ^ tag) | ((unsigned)(xq3[1] >> 32) ^ tag) | ((unsigned)(xq3[2] >> 32) ^ tag);
            if (__all(bad == 0u)) break;
            if (give_up(spins, a.err, 400 + layer)) break;
        }
        g3[0] = (unsigned)xq3[0]; g3[1] = (unsigned)xq3[1]; g3[2] = (unsigned)xq3[2];
    }
    if (st) st[4] = wall_clock64();
    const float qv = __uint_as_float(g3[0]), kv = __uint_as_float(g3[1]), vnew = __uint_as_float(g3[2]);
    // RoPE: pair (j, j + 32); lanes below 32 produce the first half, the others the second
    float qr, kr;
    {
        const float q0 = __shfl(qv, lane & 31), q1 = __shfl(qv, (lane & 31) + 32), k0 = __shfl(kv, lane & 31), k1 = __shfl(kv, (lane & 31) + 32);
        const float a0 = q0 * cs.x, a1 = q1 * cs.y, b0 = q0 * cs.y, b1 = q1 * cs.x;
        qr = lane < 32 ? a0 - a1 : b0 + b1;
        const float c0 = k0 * cs.x, c1 = k1 * cs.y, d0 = k0 * cs.y, d1 = k1 * cs.x;
        kr = lane < 32 ? c0 - c1 : d0 + d1;
    }
    qs[lane] = qr; ks[lane] = kr;
    if ((h & 3) == 0) { kT[((size_t)(lane >> 2) * S + pos) * 4 + (lane & 3)] = kr; vb[(size_t)pos * KVD + lane] = vnew; }
    lds_drain();
    if (lane == pos) {
#pragma unroll
        for (int g = 0; g < 16; ++g) kk[g] = reinterpret_cast<const float4*>(ks)[g];
    }
    float score = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float4 q4 = reinterpret_cast<const float4*>(qs)[g];
        float pr;
        pr = q4.x * kk[g].x; score = score + pr; pr = q4.y * kk[g].y; score = score + pr;
        pr = q4.z * kk[g].z; score = score + pr; pr = q4.w * kk[g].w; score = score + pr;
    }
    score = score / sqrtf((float)HS);
    const float ninf = __uint_as_float(0xff800000u);
    const float mx = wave64_max(lane < T ? score : ninf);
    float e = expf_glibc_t(lane < T ? score - mx : 0.0f, etab);
    e = lane < T ? e : 0.0f;
    const float sum = wave_serial_sum(0.0f, e, (T + 15) >> 4);
    const float w = e / sum;
    if (st) st[5] = wall_clock64();
    float o = 0.0f;
#pragma unroll
    for (int t0 = 0; t0 < TW; t0 += 8) {
        if (t0 < pos) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < pos) {                                                                   // wave-uniform
                    const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), t0 + u));
                    const float pr = wt * v[t0 + u]; o = o + pr;
                }
            }
        }
    }
    {
        const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), pos));
        const float pr = wt * vnew; o = o + pr;
    }
    // the pair's quantisation group: maximum across the two waves through LDS
    const float m1 = wave64_max(fabsf(o));
    if (lane == 0) ctl->pairmax[cw] = m1;
    lds_drain();
    pairphase += 2;
    if (lane == 0) __hip_atomic_fetch_add(&ctl->pairbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_wait_ge(&ctl->pairbar, pairphase, a.err, 500);
    const float m = fmaxf(*(volatile float*)&ctl->pairmax[0], *(volatile float*)&ctl->pairmax[1]);
    float sc; const int q = quant_one(o, m, sc);
    const unsigned pk = pack_quad(q, lane);
    const unsigned tag = tag_of(a.base, layer, 2);
    if ((lane & 3) == 0) put_gran(a.attg, cu * 32 + cw * 16 + (lane >> 2), pk, tag);
    if (cw == 0 && lane == 0) put_gran(a.attg, 512 + cu, __float_as_uint(sc), tag);
    // (second use of pairmax in the next layer is ordered behind this one by the layer's all-to-all edges)
}

template <int NCW> __device__ __forceinline__ void consumer(const EArgs& a, char* smem, Ctl* ctl, int cu, int cw, int lane) {
    constexpr int NGW = NCW >= 4 ? 4 : 2, NGX = 2048 / NGW / 64;           // waves that sweep x, granules per lane
    constexpr int NQ = NCW >= 4 ? 4 : 1;                                   // waves that normalise + quantise
    constexpr int ATT0 = NCW >= 5 ? 3 : 0;                                 // the attention CUs' two head waves (7 consumers: waves without a qkv job)
    constexpr int AGW = NCW >= 6 ? 5 : 0;                                  // the wave that gathers att_out
    constexpr int NHW = NCW >= 6 ? 6 : 3, HCNT = 2112 / NHW, NGH = (HCNT + 63) / 64;   // waves that sweep h, granules per wave / per lane
    float* xf = reinterpret_cast<float*>(smem + OFF_XF); float* sq = reinterpret_cast<float*>(smem + OFF_SQ);
    unsigned* xqw = reinterpret_cast<unsigned*>(smem + OFF_XQ); float* xs = reinterpret_cast<float*>(smem + OFF_XS);
    unsigned phase = 0, pairphase = 0;
    const bool stamped = a.stamps && (cu == 0 || cu == 200) && lane == 0 && cw == 0;
    const uint64_t etab = exp2f_tab_lane();
    if (cw == 0 && lane < 8) put_gran(a.xg, cu * 8 + lane, __float_as_uint(a.x_in[cu * 8 + lane]), tag_of(a.base, 0, 0));
    Act A;
    for (int layer = 0; layer < a.nl; ++layer) {
        long long* st = stamped ? a.stamps + ((size_t)(cu == 0 ? 0 : 1) * NL + layer) * NSTX : nullptr;
        const int s0 = layer * SLOTS;
        auto gather_x = [&](int edge, const float* nwp) __attribute__((always_inline)) {
            float4 nw[QGeom<NQ>::NF];
            if (cw < NQ) {
#pragma unroll
                for (int i = 0; i < QGeom<NQ>::NF; ++i) nw[i] = *reinterpret_cast<const float4*>(nwp + QGeom<NQ>::elem(cw, lane, i));
            }
            if (cw < NGW) {
                unsigned val[NGX];
                gath_mark(ctl, lane, 1);
                sweep<NGX>(a.xg, cw * (2048 / NGW), 2048 / NGW, tag_of(a.base, layer, edge), val, a.err, 600 + edge, lane);
                gath_mark(ctl, lane, -1);
#pragma unroll
                for (int k = 0; k < NGX; ++k) { const int e = cw * (2048 / NGW) + 64 * k + lane; const float v = __uint_as_float(val[k]); xf[e] = v; sq[sq_at(e)] = v * v; }
            }
            if (st) st[edge == 0 ? 0 : 8] = wall_clock64();
            cbar<NCW>(ctl, phase, a.err, lane);
            if (cw == 0) { const float ss = rms_chain(smem, lane); if (lane == 0) ctl->ss = ss; }
            if (st) st[edge == 0 ? 14 : 15] = wall_clock64();
            cbar<NCW>(ctl, phase, a.err, lane);
            if (cw < NQ) norm_quant<NQ>(smem, cw, nw, *(volatile float*)&ctl->ss, lane);
            if (a.dbg && cu == 0 && cw == 0 && layer == 0 && edge == 0) { lds_drain(); for (int i = lane; i < D; i += 64) a.dbg[i] = xf[i]; a.dbg[D + 528] = ctl->ss; }
            cbar<NCW>(ctl, phase, a.err, lane);
            if (a.dbg && cu == 0 && cw == 0 && layer == 0 && edge == 0) { for (int i = lane; i < 512; i += 64) a.dbg[D + i] = __uint_as_float(xqw[i]); if (lane < 16) a.dbg[D + 512 + lane] = xs[lane]; }
            if (st) st[edge == 0 ? 1 : 9] = wall_clock64();
        };
        // ---- x -> RMSNorm -> quantise -> q, k, v rows
        gather_x(0, a.norms + (size_t)layer * 2 * D);
        if (cw < JQ) {
            load_act<16>(A, smem, lane);
            const float r = run_job<16>(a, smem, ctl, s0 + cw, A, lane);
            if ((lane & 15) == 8) put_gran(a.qkvg, cu * 12 + cw * 4 + (lane >> 4), __float_as_uint(r), tag_of(a.base, layer, 1));
        }
        if (st) st[2] = wall_clock64();
        if (cu < NATT && cw >= ATT0 && cw < ATT0 + 2) attention(a, smem, ctl, layer, cu, cw - ATT0, lane, pairphase, (a.stamps && cu == 0 && lane == 0 && cw == ATT0) ? a.stamps + ((size_t)layer) * NSTX : nullptr);
        if (st) st[3] = wall_clock64();
        // ---- att_out (quantised by its producers) -> wo rows, x +=
        if (cw == AGW) {
            unsigned val[9];
            gath_mark(ctl, lane, 1);
            sweep<9>(a.attg, 0, 512 + 16, tag_of(a.base, layer, 2), val, a.err, 700, lane);
            gath_mark(ctl, lane, -1);
#pragma unroll
            for (int k = 0; k < 8; ++k) xqw[64 * k + lane] = val[k];
            if (lane < 16) xs[lane] = __uint_as_float(val[8]);
        }
        cbar<NCW>(ctl, phase, a.err, lane);
        if (st) st[6] = wall_clock64();
        if (cw < JO) {
            load_act<16>(A, smem, lane);
            const float r = run_job<16>(a, smem, ctl, s0 + JQ + cw, A, lane);
            const int row = cu * 8 + cw * 4 + (lane >> 4);
            const float xn = xf[row] + r;
            if ((lane & 15) == 8) put_gran(a.xg, row, __float_as_uint(xn), tag_of(a.base, layer, 3));
        }
        if (st) st[7] = wall_clock64();
        // ---- x -> RMSNorm -> quantise -> w1 | w3 rows (gate / up pairs, raw, into LDS)
        gather_x(3, a.norms + (size_t)layer * 2 * D + D);
        load_act<16>(A, smem, lane);
        for (int jb = cw; jb < J13; jb += NCW) {
            const float r = run_job<16>(a, smem, ctl, s0 + JQ + JO + jb, A, lane, (st && NCW == 3) ? st + NST + 8 * (jb / 3) : nullptr);
            if ((lane & 15) == 8) ctl->hraw[jb * 4 + (lane >> 4)] = r;
        }
        if (st) st[10] = wall_clock64();
        cbar<NCW>(ctl, phase, a.err, lane);
        // ---- h = SiLU(gate) * up (transformer.rs:617-620) for the CU's 32 values at once; per-CU maximum -> the group's four CUs -> quantised granules
        if (cw == 0) {
            const int hl = lane & 31;
            float hv = swiglu_t(ctl->hraw[2 * hl], ctl->hraw[2 * hl + 1], etab);
            hv = lane < 32 ? hv : 0.0f;
            const float m1 = wave64_max(fabsf(hv));
            const unsigned t4 = tag_of(a.base, layer, 4);
            if (lane == 0) put_gran(a.hmaxg, cu, __float_as_uint(m1), t4);
            unsigned mv[1];
            sweep<1>(a.hmaxg, cu & ~3, 4, t4, mv, a.err, 800, lane);
            float m = lane < 4 ? __uint_as_float(mv[0]) : 0.0f;
            m = wave64_max(m);
            float sc; const int q = quant_one(hv, m, sc);
            const unsigned pk = pack_quad(q, lane);
            const unsigned t5 = tag_of(a.base, layer, 5);
            if (lane < 32 && (lane & 3) == 0) put_gran(a.hqg, cu * 8 + (lane >> 2), pk, t5);
            if ((cu & 3) == 0 && lane == 0) put_gran(a.hqg, 2048 + (cu >> 2), __float_as_uint(sc), t5);
        }
        if (st) st[11] = wall_clock64();
        if (cw < NHW) {
            unsigned val[NGH];
            const int first = cw * HCNT;
            gath_mark(ctl, lane, 1);
            sweep<NGH>(a.hqg, first, HCNT, tag_of(a.base, layer, 5), val, a.err, 900, lane);
            gath_mark(ctl, lane, -1);
#pragma unroll
            for (int k = 0; k < NGH; ++k) {
                const int i = first + 64 * k + lane;
                if (64 * k + lane < HCNT) { if (i < 2048) xqw[i] = val[k]; else xs[i - 2048] = __uint_as_float(val[k]); }
            }
        }
        cbar<NCW>(ctl, phase, a.err, lane);
        if (st) st[12] = wall_clock64();
        load_act<64>(A, smem, lane);
        // ---- w2 rows, x +=
        for (int jb = cw; jb < J2; jb += NCW) {
            const float r = run_job<64>(a, smem, ctl, s0 + JQ + JO + J13 + jb, A, lane, (st && NCW == 3) ? st + NST + 48 + 8 * (jb / 3) : nullptr);
            const int row = cu * 8 + jb;
            const float xn = xf[row] + r;
            if (lane == 56) { put_gran(a.xg, row, __float_as_uint(xn), tag_of(a.base, layer + 1, 0)); if (layer == a.nl - 1) a.x_out[row] = xn; }
        }
        if (st) st[13] = wall_clock64();
    }
}

template <int NCW> __global__ __launch_bounds__(64 * (1 + NCW), 1) void engine_kernel(const EArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctl* ctl = reinterpret_cast<Ctl*>(smem + OFF_CTL);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    if (threadIdx.x < sizeof(Ctl) / 4) reinterpret_cast<unsigned*>(ctl)[threadIdx.x] = 0u;
    __syncthreads();
    if (wave == 0) loader(a, smem, ctl, cu, lane);
    else consumer<NCW>(a, smem, ctl, cu, wave - 1, lane);
}

// ================================================================================================================ host
static unsigned rng_state = 12345u;
static inline unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static inline float urand() { return (float)(rnd() & 0xffffff) / 16777216.0f; }

struct Mat { int o, n; std::vector<int8_t> q; std::vector<float> s; };
static void fill_mat(Mat& m, int o, int n) {
    m.o = o; m.n = n; m.q.resize((size_t)o * n); m.s.resize((size_t)o * n / 128);
    for (auto& b : m.q) b = (int8_t)((int)(rnd() % 255) - 127);
    const float base = 1.0f / (127.0f * sqrtf((float)n));
    for (auto& f : m.s) f = (0.5f + urand()) * base;
}
// a job's slot: 8 pieces of 1 KiB (lane l of piece u: row l / L, cluster (l % L) / 8, group cluster * 8 + u, bytes 16 (l % 8) ..) + 64 scales [cluster][u]
static void pack_job(char* dst, const Mat& m, int row0, int L) {
    for (int u = 0; u < 8; ++u)
        for (int l = 0; l < 64; ++l) {
            const int r = row0 + l / L, g = ((l % L) / 8) * 8 + u;
            memcpy(dst + u * 1024 + l * 16, m.q.data() + (size_t)r * m.n + g * 128 + (l % 8) * 16, 16);
        }
    float* sc = reinterpret_cast<float*>(dst + 8192);
    for (int c = 0; c < 8; ++c)
        for (int u = 0; u < 8; ++u) { const int l = c * 8, r = row0 + l / L, g = ((l % L) / 8) * 8 + u; sc[c * 8 + u] = m.s[(size_t)r * (m.n / 128) + g]; }
}
static void h_rmsnorm(float* o, const float* x, const float* w) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < D / 8; ++j) for (int k = 0; k < 8; ++k) p[k] = p[k] + x[8 * j + k] * x[8 * j + k];
    const float q0 = p[0] + p[4], q1 = p[1] + p[5], q2 = p[2] + p[6], q3 = p[3] + p[7];
    float ss = (q0 + q2) + (q1 + q3);
    ss = ss / (float)D; ss = ss + EPS; ss = 1.0f / sqrtf(ss);
    for (int i = 0; i < D; ++i) o[i] = w[i] * (ss * x[i]);
}
static void h_quant(int8_t* q, float* s, const float* x, int n) {
    for (int g = 0; g < n / 128; ++g) {
        float m = 0.0f;
        for (int i = 0; i < 128; ++i) { const float v = fabsf(x[g * 128 + i]); if (v > m) m = v; }
        const float sc = m / 127.0f; s[g] = sc;
        for (int i = 0; i < 128; ++i) {
            float r = roundf(x[g * 128 + i] / sc);
            int v = (r != r) ? 0 : (int)fminf(fmaxf(r, -128.0f), 127.0f);
            q[g * 128 + i] = (int8_t)v;
        }
    }
}
static void h_matmul(float* out, const Mat& m, const int8_t* xq, const float* xs) {
    const int G = m.n / 128;
    for (int r = 0; r < m.o; ++r) {
        float acc = 0.0f;
        const int8_t* w = m.q.data() + (size_t)r * m.n;
        for (int g = 0; g < G; ++g) {
            int isum = 0;
            for (int i = 0; i < 128; ++i) isum += (int)w[g * 128 + i] * (int)xq[g * 128 + i];
            const float p = (float)isum * m.s[(size_t)r * G + g];
            acc = acc + p * xs[g];
        }
        out[r] = acc;
    }
}

struct Layer { Mat qkv, wo, w13, w2; std::vector<float> n_att, n_ffn; };

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 16, thin = argc > 2 ? atoi(argv[2]) : 1, depth = argc > 3 ? atoi(argv[3]) : 4, ncw = argc > 5 ? atoi(argv[5]) : 7;
    const int P0 = 16, S = 64, CHECK = 3;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("engine: %s, %d CUs; Llama-3.2-1B Q8_0 shapes, %d layers per launch, positions %d..%d, loader thinning %d, depth %d, %d consumer waves per CU\n", prop.name, prop.multiProcessorCount, NL, P0, P0 + steps - 1, thin, depth, ncw);
    if (prop.multiProcessorCount < NCU) { printf("needs %d CUs\n", NCU); return 1; }
    std::vector<Layer> Ls(NL);
    for (auto& l : Ls) {
        fill_mat(l.qkv, QKV, D); fill_mat(l.wo, D, D); fill_mat(l.w13, 2 * HID, D); fill_mat(l.w2, D, HID);
        l.n_att.resize(D); l.n_ffn.resize(D);
        for (auto& f : l.n_att) f = 0.5f + urand();
        for (auto& f : l.n_ffn) f = 0.5f + urand();
    }
    // the per-CU streams
    const size_t stream_bytes = (size_t)NCU * NL * SLOTS * SLOTB;
    std::vector<char> stream(stream_bytes);
    for (int cu = 0; cu < NCU; ++cu)
        for (int l = 0; l < NL; ++l) {
            char* base = stream.data() + ((size_t)cu * NL + l) * SLOTS * SLOTB;
            int s = 0;
            for (int j = 0; j < JQ; ++j) pack_job(base + (size_t)(s++) * SLOTB, Ls[l].qkv, cu * 12 + j * 4, 16);
            for (int j = 0; j < JO; ++j) pack_job(base + (size_t)(s++) * SLOTB, Ls[l].wo, cu * 8 + j * 4, 16);
            for (int j = 0; j < J13; ++j) pack_job(base + (size_t)(s++) * SLOTB, Ls[l].w13, cu * 64 + j * 4, 16);
            for (int j = 0; j < J2; ++j) pack_job(base + (size_t)(s++) * SLOTB, Ls[l].w2, cu * 8 + j, 64);
        }
    std::vector<float> norms((size_t)NL * 2 * D);
    for (int l = 0; l < NL; ++l) { memcpy(&norms[(size_t)l * 2 * D], Ls[l].n_att.data(), D * 4); memcpy(&norms[(size_t)l * 2 * D + D], Ls[l].n_ffn.data(), D * 4); }
    std::vector<float> rope((size_t)S * 32 * 2);
    for (int p = 0; p < S; ++p) for (int j = 0; j < 32; ++j) { const float f = (float)p * powf(500000.0f, -(float)(2 * j) / 64.0f); rope[((size_t)p * 32 + j) * 2] = cosf(f); rope[((size_t)p * 32 + j) * 2 + 1] = sinf(f); }
    // history of the positions below P0: random keys / values (host layout [layer][pos][KVD]); inputs of every step
    std::vector<float> hk((size_t)NL * S * KVD, 0.0f), hv((size_t)NL * S * KVD, 0.0f);
    for (int l = 0; l < NL; ++l) for (int p = 0; p < P0; ++p) for (int d = 0; d < KVD; ++d) { hk[((size_t)l * S + p) * KVD + d] = urand() - 0.5f; hv[((size_t)l * S + p) * KVD + d] = urand() - 0.5f; }
    std::vector<float> xin((size_t)steps * D);
    for (auto& f : xin) f = 2.0f * urand() - 1.0f;
    std::vector<float> kc_dev((size_t)NL * NKV * HS * S), vc_dev = hv;
    for (int l = 0; l < NL; ++l) for (int kvh = 0; kvh < NKV; ++kvh) for (int d = 0; d < HS; ++d) for (int p = 0; p < S; ++p)
        kc_dev[(((size_t)l * NKV + kvh) * (HS / 4) + d / 4) * S * 4 + (size_t)p * 4 + (d & 3)] = hk[((size_t)l * S + p) * KVD + kvh * HS + d];

    // ---- device
    char* d_stream; float *d_norms, *d_rope, *d_kc, *d_vc, *d_xin, *d_xout; u64* d_gran; int* d_err; long long* d_st;
    CK(hipMalloc(&d_stream, stream_bytes)); CK(hipMemcpy(d_stream, stream.data(), stream_bytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_norms, norms.size() * 4)); CK(hipMemcpy(d_norms, norms.data(), norms.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_rope, rope.size() * 4)); CK(hipMemcpy(d_rope, rope.data(), rope.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_kc, kc_dev.size() * 4)); CK(hipMalloc(&d_vc, vc_dev.size() * 4));
    CK(hipMalloc(&d_xin, xin.size() * 4)); CK(hipMemcpy(d_xin, xin.data(), xin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_xout, (size_t)steps * D * 4)); CK(hipMemset(d_xout, 0xff, (size_t)steps * D * 4));
    const size_t n_gran = 2048 + 3072 + 1024 + 256 + 2112 + 64;
    CK(hipMalloc(&d_gran, n_gran * 8)); CK(hipMemset(d_gran, 0, n_gran * 8));
    CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
    const size_t n_st = (size_t)2 * NL * NSTX;
    CK(hipMalloc(&d_st, n_st * 8)); CK(hipMemset(d_st, 0, n_st * 8));
    CK(hipFuncSetAttribute((const void*)engine_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
    CK(hipFuncSetAttribute((const void*)engine_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
    auto launch = [&](const EArgs& ea) { if (ncw == 3) hipLaunchKernelGGL(engine_kernel<3>, dim3(NCU), dim3(256), SMEM, 0, ea); else hipLaunchKernelGGL(engine_kernel<7>, dim3(NCU), dim3(512), SMEM, 0, ea); };
    EArgs a{};
    a.stream = d_stream; a.norms = d_norms; a.rope = (const float2*)d_rope; a.kc = d_kc; a.vc = d_vc;
    a.xg = d_gran; a.qkvg = d_gran + 2048; a.attg = a.qkvg + 3072; a.hmaxg = a.attg + 1024; a.hqg = a.hmaxg + 256; a.err = d_err; a.S = S; a.thin = thin; a.depth = depth;
    unsigned base = 0;
    a.nl = NL;
    const bool debug = argc > 4 && atoi(argv[4]) != 0;
    auto run_steps = [&](bool stamps) {
        for (int s = 0; s < steps; ++s) {
            a.pos = P0 + s; a.x_in = d_xin + (size_t)s * D; a.x_out = d_xout + (size_t)s * D; a.base = base; base += (NL + 1) * 8;
            a.stamps = (stamps && s == steps / 2) ? d_st : nullptr;
            launch(a);
        }
    };
    auto reset_cache = [&]() { CK(hipMemcpy(d_kc, kc_dev.data(), kc_dev.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vc, vc_dev.data(), vc_dev.size() * 4, hipMemcpyHostToDevice)); };
    reset_cache();
    run_steps(false);
    CK(hipDeviceSynchronize());
    int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    printf("first pass: err %d\n", err);
    std::vector<float> xout((size_t)steps * D);
    CK(hipMemcpy(xout.data(), d_xout, xout.size() * 4, hipMemcpyDeviceToHost));

    // ---- debug: ONE layer of the first position, every edge against the host (the granule buffers keep the layer's values)
    if (debug) {
        CK(hipMemset(d_gran, 0, n_gran * 8)); reset_cache();
        float* d_dbg; CK(hipMalloc(&d_dbg, 4096 * 4)); CK(hipMemset(d_dbg, 0, 4096 * 4)); a.dbg = d_dbg;
        a.nl = 1; a.pos = P0; a.x_in = d_xin; a.x_out = d_xout; a.base = 1u << 20; a.stamps = nullptr;
        launch(a);
        CK(hipDeviceSynchronize());
        std::vector<u64> gr(n_gran); CK(hipMemcpy(gr.data(), d_gran, n_gran * 8, hipMemcpyDeviceToHost));
        const u64 *gx = gr.data(), *gqkv = gx + 2048, *gatt = gqkv + 3072, *ghmax = gatt + 1024, *ghq = ghmax + 256;
        const Layer& Lr = Ls[0]; const int pos = P0, l = 0;
        std::vector<float> K = hk, V = hv, x(xin.begin(), xin.begin() + D), xb(D), qkv(QKV), att(D), t2(D), h(HID), gu(2 * HID), xs(HID / 128);
        std::vector<int8_t> xq(HID);
        auto cmpf = [&](const char* what, const u64* g, const float* ref, int n, unsigned tag) {
            int bad = 0, badtag = 0;
            for (int i = 0; i < n; ++i) { unsigned u; memcpy(&u, &ref[i], 4); if ((unsigned)g[i] != u) { if (bad < 3) { float f; unsigned gv = (unsigned)g[i]; memcpy(&f, &gv, 4); printf("   %s[%d]: host %.9g device %.9g\n", what, i, ref[i], f); } ++bad; } if ((unsigned)(g[i] >> 32) != tag) ++badtag; }
            printf("debug %-28s %d of %d differ, %d wrong tags\n", what, bad, n, badtag);
        };
        h_rmsnorm(xb.data(), x.data(), Lr.n_att.data()); h_quant(xq.data(), xs.data(), xb.data(), D); h_matmul(qkv.data(), Lr.qkv, xq.data(), xs.data());
        {
            std::vector<float> dg(4096); CK(hipMemcpy(dg.data(), d_dbg, 4096 * 4, hipMemcpyDeviceToHost));
            int bx = 0, bq = 0, bs = 0;
            for (int i = 0; i < D; ++i) if (memcmp(&dg[i], &x[i], 4)) { if (bx < 3) printf("   xf[%d] host %.9g device %.9g\n", i, x[i], dg[i]); ++bx; }
            for (int i = 0; i < 512; ++i) if (memcmp(&dg[D + i], &xq[4 * i], 4)) { if (bq < 3) { unsigned a_, b_; memcpy(&a_, &dg[D + i], 4); memcpy(&b_, &xq[4 * i], 4); printf("   xq word %d host %08x device %08x\n", i, b_, a_); } ++bq; }
            for (int i = 0; i < 16; ++i) if (memcmp(&dg[D + 512 + i], &xs[i], 4)) { if (bs < 3) printf("   xs[%d] host %.9g device %.9g\n", i, xs[i], dg[D + 512 + i]); ++bs; }
            printf("debug CU 0 first prologue: xf %d of 2048 differ, xq %d of 512 words, xs %d of 16; 1 / rms device %.9g\n", bx, bq, bs, dg[D + 528]);
        }
        cmpf("q / k / v rows", gqkv, qkv.data(), QKV, tag_of(a.base, 0, 1));
        for (int hh = 0; hh < NH + NKV; ++hh) {
            float* vec = hh < NH ? &qkv[hh * HS] : &qkv[D + (hh - NH) * HS];
            for (int j = 0; j < 32; ++j) {
                const float fcr = rope[((size_t)pos * 32 + j) * 2], fci = rope[((size_t)pos * 32 + j) * 2 + 1];
                const float v0 = vec[j], v1 = vec[j + 32];
                const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
                vec[j] = a0 - a1; vec[j + 32] = b0 + b1;
            }
        }
        memcpy(&K[((size_t)l * S + pos) * KVD], &qkv[D], KVD * 4); memcpy(&V[((size_t)l * S + pos) * KVD], &qkv[D + KVD], KVD * 4);
        for (int hh = 0; hh < NH; ++hh) {
            const int kvh = hh / 4; float sc[TW + 1]; float mx = -INFINITY;
            for (int t = 0; t <= pos; ++t) {
                const float* kt = &K[((size_t)l * S + t) * KVD + kvh * HS];
                float score = 0.0f;
                for (int d = 0; d < HS; ++d) { const float pr = qkv[hh * HS + d] * kt[d]; score = score + pr; }
                score = score / sqrtf((float)HS); sc[t] = score; if (score > mx) mx = score;
            }
            float sum = 0.0f;
            for (int t = 0; t <= pos; ++t) { sc[t] = expf(sc[t] - mx); sum = sum + sc[t]; }
            for (int t = 0; t <= pos; ++t) sc[t] = sc[t] / sum;
            for (int d = 0; d < HS; ++d) { float o = 0.0f; for (int t = 0; t <= pos; ++t) { const float pr = sc[t] * V[((size_t)l * S + t) * KVD + kvh * HS + d]; o = o + pr; } att[hh * HS + d] = o; }
        }
        h_quant(xq.data(), xs.data(), att.data(), D);
        { int bad = 0; for (int i = 0; i < 512; ++i) { unsigned w; memcpy(&w, &xq[4 * i], 4); if ((unsigned)gatt[i] != w) { if (bad < 3) printf("   att word %d: host %08x device %08x (att %.6g %.6g %.6g %.6g scale %.6g)\n", i, w, (unsigned)gatt[i], att[4*i], att[4*i+1], att[4*i+2], att[4*i+3], xs[i / 32]); ++bad; } }
          printf("debug %-28s %d of 512 words differ\n", "att_out quantised", bad); cmpf("att_out scales", gatt + 512, xs.data(), 16, tag_of(a.base, 0, 2)); }
        h_matmul(t2.data(), Lr.wo, xq.data(), xs.data());
        for (int i = 0; i < D; ++i) x[i] = x[i] + t2[i];
        h_rmsnorm(xb.data(), x.data(), Lr.n_ffn.data()); h_quant(xq.data(), xs.data(), xb.data(), D); h_matmul(gu.data(), Lr.w13, xq.data(), xs.data());
        for (int i = 0; i < HID; ++i) { const float g = gu[2 * i], u = gu[2 * i + 1]; const float e = expf(-g); const float gg = 1.0f / (1.0f + e); float val = g * gg; val = val * u; h[i] = val; }
        h_quant(xq.data(), xs.data(), h.data(), HID);
        { std::vector<float> hm(256); for (int c = 0; c < 256; ++c) { float m = 0; for (int i = 0; i < 32; ++i) m = fmaxf(m, fabsf(h[c * 32 + i])); hm[c] = m; } cmpf("h per-CU maxima", ghmax, hm.data(), 256, tag_of(a.base, 0, 4)); }
        { int bad = 0; for (int i = 0; i < 2048; ++i) { unsigned w; memcpy(&w, &xq[4 * i], 4); if ((unsigned)ghq[i] != w) { if (bad < 3) printf("   h word %d: host %08x device %08x\n", i, w, (unsigned)ghq[i]); ++bad; } }
          printf("debug %-28s %d of 2048 words differ\n", "h quantised", bad); cmpf("h scales", ghq + 2048, xs.data(), 64, tag_of(a.base, 0, 5)); }
        h_matmul(t2.data(), Lr.w2, xq.data(), xs.data());
        for (int i = 0; i < D; ++i) x[i] = x[i] + t2[i];
        cmpf("x after the layer", gx, x.data(), D, tag_of(a.base, 1, 0));
        return 0;
    }
    // ---- host loop for the first CHECK positions
    {
        std::vector<float> K = hk, V = hv, x(D), xb(D), qkv(QKV), att(D), t2(D), h(HID), gu(2 * HID), xs(HID / 128);
        std::vector<int8_t> xq(HID);
        int bad_total = 0;
        for (int s = 0; s < CHECK && s < steps; ++s) {
            const int pos = P0 + s;
            memcpy(x.data(), &xin[(size_t)s * D], D * 4);
            for (int l = 0; l < NL; ++l) {
                const Layer& Lr = Ls[l];
                h_rmsnorm(xb.data(), x.data(), Lr.n_att.data()); h_quant(xq.data(), xs.data(), xb.data(), D); h_matmul(qkv.data(), Lr.qkv, xq.data(), xs.data());
                float* kr = &K[((size_t)l * S + pos) * KVD]; float* vr = &V[((size_t)l * S + pos) * KVD];
                for (int hh = 0; hh < NH + NKV; ++hh) {
                    float* vec = hh < NH ? &qkv[hh * HS] : &qkv[D + (hh - NH) * HS];
                    for (int j = 0; j < 32; ++j) {
                        const float fcr = rope[((size_t)pos * 32 + j) * 2], fci = rope[((size_t)pos * 32 + j) * 2 + 1];
                        const float v0 = vec[j], v1 = vec[j + 32];
                        const float a0 = v0 * fcr, a1 = v1 * fci, b0 = v0 * fci, b1 = v1 * fcr;
                        vec[j] = a0 - a1; vec[j + 32] = b0 + b1;
                    }
                }
                memcpy(kr, &qkv[D], KVD * 4); memcpy(vr, &qkv[D + KVD], KVD * 4);
                for (int hh = 0; hh < NH; ++hh) {
                    const int kvh = hh / 4; float sc[TW + 1]; float mx = -INFINITY;
                    for (int t = 0; t <= pos; ++t) {
                        const float* kt = &K[((size_t)l * S + t) * KVD + kvh * HS];
                        float score = 0.0f;
                        for (int d = 0; d < HS; ++d) { const float pr = qkv[hh * HS + d] * kt[d]; score = score + pr; }
                        score = score / sqrtf((float)HS); sc[t] = score; if (score > mx) mx = score;
                    }
                    float sum = 0.0f;
                    for (int t = 0; t <= pos; ++t) { sc[t] = expf(sc[t] - mx); sum = sum + sc[t]; }
                    for (int t = 0; t <= pos; ++t) sc[t] = sc[t] / sum;
                    for (int d = 0; d < HS; ++d) {
                        float o = 0.0f;
                        for (int t = 0; t <= pos; ++t) { const float pr = sc[t] * V[((size_t)l * S + t) * KVD + kvh * HS + d]; o = o + pr; }
                        att[hh * HS + d] = o;
                    }
                }
                h_quant(xq.data(), xs.data(), att.data(), D); h_matmul(t2.data(), Lr.wo, xq.data(), xs.data());
                for (int i = 0; i < D; ++i) x[i] = x[i] + t2[i];
                h_rmsnorm(xb.data(), x.data(), Lr.n_ffn.data()); h_quant(xq.data(), xs.data(), xb.data(), D); h_matmul(gu.data(), Lr.w13, xq.data(), xs.data());
                for (int i = 0; i < HID; ++i) { const float g = gu[2 * i], u = gu[2 * i + 1]; const float e = expf(-g); const float gg = 1.0f / (1.0f + e); float val = g * gg; val = val * u; h[i] = val; }
                h_quant(xq.data(), xs.data(), h.data(), HID); h_matmul(t2.data(), Lr.w2, xq.data(), xs.data());
                for (int i = 0; i < D; ++i) x[i] = x[i] + t2[i];
            }
            int bad = 0; double amax = 0;
            for (int i = 0; i < D; ++i) { unsigned ua, ub; memcpy(&ua, &x[i], 4); memcpy(&ub, &xout[(size_t)s * D + i], 4); if (ua != ub) { if (bad < 3) printf("  step %d x[%d]: host %.9g device %.9g\n", s, i, x[i], xout[(size_t)s * D + i]); ++bad; } if (fabs(x[i]) > amax) amax = fabs(x[i]); }
            printf("self-check position %d: %d of %d values differ from the host loop (max |x| %.3f)\n", pos, bad, D, amax);
            bad_total += bad;
        }
        printf("self-check: %s\n", bad_total == 0 ? "BIT-EXACT" : "MISMATCH");
    }

    // ---- timing: the same positions again (the cache rows they write are rewritten with the same values)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        run_steps(rep == 4);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %.1f us per step = %.2f us per layer (incl. one launch per step)\n", rep, ms * 1000.0f / steps, ms * 1000.0f / steps / NL);
        if (ms < best) best = ms;
    }
    CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    printf("RESULT engine consumers=%d thin=%d depth=%d: %.2f us per layer (best of 5; %.1f us per 16-layer step at positions %d..%d), err %d, stream %.1f MB per layer = %.2f TB/s\n",
           ncw, thin, depth, best * 1000.0f / steps / NL, best * 1000.0f / steps, P0, P0 + steps - 1, err, (double)NCU * SLOTS * SLOTB / 1e6, (double)NCU * SLOTS * SLOTB * NL * steps / (best * 1e-3) / 1e12);
    // ---- stamp table (100 MHz wall clock): mean over layers 1..15 of the stamped step
    std::vector<long long> stv(n_st); CK(hipMemcpy(stv.data(), d_st, n_st * 8, hipMemcpyDeviceToHost));
    const char* nm[NST] = {"x gathered (16 KB of granules)", "  normalise + quantise done", "qkv rows published", "attention done (CU 0, 3 consumers only)", "  q/k/v granules seen", "  softmax done",
                           "att_out gathered (4 KB)", "wo rows published", "x gathered (16 KB of granules)", "  normalise + quantise done", "w1|w3 rows done", "h: SiLU, max hop, quantise, published", "h gathered (16.5 KB)", "w2 rows published", "  RMS chain done", "  RMS chain done"};
    const int order[] = {0, 14, 1, 2, 4, 5, 3, 6, 7, 8, 15, 9, 10, 11, 12, 13};
    for (int c = 0; c < 2; ++c) {
        printf("stamps, CU %d (%s), wave 0, us since the previous row (mean over layers 1..%d):\n", c == 0 ? 0 : 200, c == 0 ? "attention CU" : "plain CU", NL - 1);
        for (int oi = 0; oi < 16; ++oi) {
            const int k = order[oi];
            double acc = 0; int n = 0;
            for (int l = 1; l < NL; ++l) {
                const long long* st = &stv[((size_t)c * NL + l) * NSTX];
                long long prev;
                if (oi == 0) prev = stv[((size_t)c * NL + l - 1) * NSTX + 13]; else prev = st[order[oi - 1]];
                if ((k == 4 || k == 5) && c == 1) continue;
                if (k == 3 && c == 0) prev = st[5];
                if (k == 3 && c == 1) prev = st[2];
                if (st[k] && prev) { acc += (double)(st[k] - prev) / 100.0; ++n; }
            }
            if (n) printf("  %-40s %6.2f\n", nm[k], acc / n);
        }
        double tot = 0; int n = 0;
        for (int l = 1; l < NL; ++l) { const long long a0 = stv[((size_t)c * NL + l - 1) * NSTX + 13], a1 = stv[((size_t)c * NL + l) * NSTX + 13]; if (a0 && a1) { tot += (double)(a1 - a0) / 100.0; ++n; } }
        if (n) printf("  %-40s %6.2f\n", "layer (w2 published -> w2 published)", tot / n);
        for (int j = 0; j < 9; ++j) {
            double w = 0, d = 0, ch = 0, ahead = 0; int m = 0;
            for (int l = 1; l < NL; ++l) { const long long* js = &stv[((size_t)c * NL + l) * NSTX + NST + 8 * j]; if (js[0] && js[3]) { w += (js[1] - js[0]) / 100.0; d += (js[2] - js[1]) / 100.0; ch += (js[3] - js[2]) / 100.0; ahead += (double)js[4]; ++m; } }
            if (m) printf("    wave 0 %s job %d: wait for the slot %5.2f, reads + dots %5.2f, chain %5.2f us; loader %5.1f slots ahead at entry\n", j < 6 ? "w1|w3" : "w2", j < 6 ? j : j - 6, w / m, d / m, ch / m, ahead / m);
        }
    }
    return err != 0;
}
