// lmrs_stage.h — fully static (compile-time shape) building blocks of the decode path, shared by the
// per-stage kernels (lmrs_kernels.hip), the merged qkv + attention launch and the batched prefill (lmrs_prefill.inc).
//
// Why static: with the vector length N, the lanes-per-row L and the steps-per-lane U known at compile
// time there is not a single data-dependent branch around a global load, so hipcc keeps *counted*
// s_waitcnt vmcnt(k) waits: the activation loads (issued first) can be consumed while the whole weight
// tile issued right after them is still in flight.  (With runtime trip counts the compiler falls back to
// vmcnt(0) before the first LDS write and the prologue serialises behind the weight stream.)
//
// Arithmetic: identical, operation for operation, to the generic kernels (see lmrs_kernels.hip header).
#pragma once
#include "lmrs_device_math.h"
#include "lmrs_kernels.h"

namespace lmrs {

typedef int i32x4 __attribute__((ext_vector_type(4)));     // native vector: what the nontemporal builtin accepts

// Pointers that reach a kernel through a table in device memory (the engine's per-layer weight table) are generic
// to the compiler, which would then emit flat_load (also counted in lgkmcnt).  Every such access goes through an
// explicit global-address-space pointer.
#define LMRS_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const LMRS_GLOBAL T* as_global(const T* p) { return (const LMRS_GLOBAL T*)p; }
constexpr int kBlk = 256;
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// ------------------------------------------------------------------------------------------------
// Memory flavours.  COH = data exchanged between workgroups INSIDE one launch (persistent engine):
// agent-scope relaxed atomics (global_load/store ... sc1), which bypass the per-CU L1 and are written
// through, so no fences (a fence would also drain the in-flight weight tiles).  !COH = data produced by
// an earlier launch: plain accesses.
// ------------------------------------------------------------------------------------------------
template <bool COH> __device__ __forceinline__ float ld_f32(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH> __device__ __forceinline__ void st_f32(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH> __device__ __forceinline__ float4 ld_f32x4(const float* p) {
    if constexpr (COH) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                           __uint_as_float((unsigned)(b >> 32)));
    } else {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 t = *as_global(reinterpret_cast<const f32x4*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
}

// ------------------------------------------------------------------------------------------------
// Activation vector in registers: thread t owns elements i*4*NTH + 4t .. +3, i < NP = ceil(N/(4*NTH))
// (32 consecutive lanes own one 128-element quantisation group).
// ------------------------------------------------------------------------------------------------
// NTH = threads per workgroup (256 or 512): more threads shorten the per-lane share of the prologue.
// Grouped passes (round 4): the first NPG = 2, 4 or 8 whole passes are laid out so that a lane's NPG float4s come from ONE 128-element
// quantisation group - LG = 32 / NPG consecutive lanes own a group, lane l of the cluster the elements 4l .. 4l+3 of each of the group's
// NPG slices of 4 LG elements (a load instruction still reads whole 128-byte lines: 8 lanes x 16 B at NPG = 4, 16 x 16 B at NPG = 2).
// The quantiser then needs one group maximum, one scale and one exactness check per LANE instead of one per lane and pass: w2's prologue
// (8192 values, 512 threads, NPG = 4) went from 245 to 107 vector instructions per wave.  Passes beyond NPG (3072 = 2 + 1, the ragged
// tails of 2304 and 9216) keep the linear layout, 32 lanes per group.  Element order in LDS / memory is unchanged.
template <int N, int NTH = kBlk> struct VecGeom {
    static constexpr int PER = NTH * 4;                       // elements per pass over the workgroup
    static constexpr int NP = (N + PER - 1) / PER;
    static constexpr bool FULL = (N % PER) == 0;
    static constexpr int G = N / 128;
#ifndef LMRS_NO_GROUPED
    static constexpr int NPG = N / PER >= 8 ? 8 : N / PER >= 4 ? 4 : N / PER >= 2 ? 2 : 0;     // grouped passes
#else
    static constexpr int NPG = 0;
#endif
    static constexpr int LG = NPG ? 32 / NPG : 32;            // lanes that own one quantisation group of the grouped passes
    // first element of thread t's float4 number i (i: a constant after unrolling)
    __device__ static __forceinline__ int elem(int i, int t) {
        if (i < NPG) return (t / LG) * 128 + i * (LG * 4) + (t % LG) * 4;
        return i * PER + t * 4;
    }
    // is float4 number i of thread t inside the vector (only the last pass of a ragged vector can be outside)
    __device__ static __forceinline__ bool live(int i, int t) { return FULL || i < NP - 1 || i * PER + t * 4 < N; }
};

template <int N, bool COH, int NTH = kBlk>
__device__ __forceinline__ void vec_load(float4 (&v)[(VecGeom<N, NTH>::NP)], const float* __restrict__ x) {
    constexpr int NP = VecGeom<N, NTH>::NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = VecGeom<N, NTH>::elem(i, (int)threadIdx.x);
        if constexpr (VecGeom<N, NTH>::FULL) v[i] = ld_f32x4<COH>(x + e);
        else if (i < NP - 1) v[i] = ld_f32x4<COH>(x + e);
        else {                                                     // (the ragged pass is never a grouped one: NPG <= N / PER)
            // ragged last pass (N = 2304, 9216): the load stays UNCONDITIONAL on a clamped address and the lanes past the end are
            // zeroed by a select - a load under a lane predicate makes hipcc wait for ALL outstanding loads (vmcnt(0)) right there,
            // i.e. before the weight tile is even requested (measured: x landed at 2.9 us instead of 0.7 in the Gemma prologues)
            const bool live = e < N;
            const float4 t4 = ld_f32x4<COH>(x + (live ? e : N - 4));
            v[i] = live ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// RMSNorm (reference functional.rs:48-78), in place on v[]; nw[] = norm weights of the same elements.
// scratch: rms_scratch_floats(N) floats of LDS.
// `landed` runs right after the first barrier, i.e. once the activation has arrived (hook for deferred weight loads).
constexpr int rms_scratch_floats(int n) { return 8 * (n / 8 + 8) + 4; }
// (Round 4, measured and removed: the 8 chains from REGISTERS - chain k in row k & 3 of two waves, 16 squares per register, every add a
// v_add_f32_dpp row_shr as in wave_serial_sum, no LDS read in the loop.  Bit-equal; the chain itself 0.83 -> 0.70 us in the qkv launch,
// unchanged in w1/w3 (a DPP add issues every ~6 cycles, not 4), and the step 426 -> 441 us: two busy waves per workgroup instead of one
// push out the slowest workgroup of every launch that has more workgroups than CUs.)
// one 16-byte batch of the chain: this lane's four squares, then the four of the lane 8 above (row_shl:8: the DPP operand of the add
// itself fetches them - see vec_rmsnorm)
__device__ __forceinline__ void rms_chain8(float& p, const float4& a) {
    asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\t"
                 "v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_shl:8 row_mask:0xf bank_mask:0xf"
                 : "+v"(p) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
}
// four batches as ONE asm block: between separate asm statements hipcc inserts an s_nop (it cannot see what the block reads), and an
// s_nop costs the chain a whole 4-cycle issue slot like an add (tools/ubench/addlat.hip)
#define LMRS_RMS8(a, b, c, d) "v_add_f32 %0, %0, %" #a "\n\tv_add_f32 %0, %0, %" #b "\n\tv_add_f32 %0, %0, %" #c "\n\tv_add_f32 %0, %0, %" #d "\n\t" \
    "v_add_f32_dpp %0, %" #a ", %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %" #b ", %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %0, %" #c ", %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %" #d ", %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void rms_chain32(float& p, const float4 (&a)[4]) {
    asm volatile(LMRS_RMS8(1, 2, 3, 4) LMRS_RMS8(5, 6, 7, 8) LMRS_RMS8(9, 10, 11, 12) LMRS_RMS8(13, 14, 15, 16)
                 : "+v"(p)
                 : "v"(a[0].x), "v"(a[0].y), "v"(a[0].z), "v"(a[0].w), "v"(a[1].x), "v"(a[1].y), "v"(a[1].z), "v"(a[1].w),
                   "v"(a[2].x), "v"(a[2].y), "v"(a[2].z), "v"(a[2].w), "v"(a[3].x), "v"(a[3].y), "v"(a[3].z), "v"(a[3].w));
}
template <int N, int NTH = kBlk, class F = NoHook>
__device__ __forceinline__ void vec_rmsnorm(float4 (&v)[(VecGeom<N, NTH>::NP)], const float4 (&nw)[(VecGeom<N, NTH>::NP)], float eps, int add_unit, float* scratch,
                                            unsigned long long* dbg = nullptr, F landed = F(), const int chain_wave = 0) {
    constexpr int NP = VecGeom<N, NTH>::NP, JP = N / 8 + 8, NM = N / 64;     // JP % 64 == 8 (or 40): the 16 chain lanes' 16-byte reads hit 16 different bank groups
    static_assert(N % 256 == 0, "N must be a multiple of 256");
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = VecGeom<N, NTH>::elem(i, t);
        if (VecGeom<N, NTH>::FULL || i < NP - 1 || e < N) {
            const int j = e >> 3, k0 = e & 7;
            scratch[(k0 + 0) * JP + j] = v[i].x * v[i].x;
            scratch[(k0 + 1) * JP + j] = v[i].y * v[i].y;
            scratch[(k0 + 2) * JP + j] = v[i].z * v[i].z;
            scratch[(k0 + 3) * JP + j] = v[i].w * v[i].w;
        }
    }
    lds_barrier();
    landed();
    if (dbg && t == 0) dbg[4] = wall_clock64();          // activation landed, squares in LDS
    // chain_wave: which wave of the workgroup runs the serial chain.  Launches with two workgroups per CU (w1/w3, the classifier) put
    // both workgroups' wave 0 on the same SIMD, where the two chains share one instruction issue port: the second workgroup of a CU
    // (dispatch order b + 256) runs its chain on another wave, i.e. another SIMD.
    if ((t >> 6) == chain_wave) {
        // The 8 strided partial sums (ss_sim += x*x) are 8 serial chains of N/8 adds each, in lanes 0..7.  A chain is bound by
        // instruction issue (one dependent v_add_f32 per 4 cycles), so everything that is NOT an add of the chain is overhead: the
        // squares are read 16 bytes per lane by SIXTEEN lanes - lane k the values j .. j+3 of chain k, lane 8+k the values j+4 .. j+7
        // of the same chain - and lane k adds its own four, then the four of lane 8+k through the add's DPP operand (row_shl:8):
        // 8 adds per LDS read instead of 4, no move instructions.  (Lanes 16..63 repeat lanes 0..15: no divergence, no extra traffic.)
        // Batches of 4 reads ping-pong so that the next batch is in flight while the current one is added.
        float p = 0.0f;
        const int cl = t & 15;
        const float4* row = reinterpret_cast<const float4*>(scratch + (cl & 7) * JP) + (cl >> 3);     // float4 2m + (cl >> 3) = values 8m + 4 (cl >> 3) ..
        constexpr int BF = 4, NB = NM / BF;
        static_assert(NM % BF == 0, "row length must be a multiple of 32 values");
        float4 A[BF], B[BF];
#pragma unroll
        for (int u = 0; u < BF; ++u) A[u] = row[2 * u];
#pragma unroll
        for (int b0 = 0; b0 < NB; b0 += 2) {
            if (b0 + 1 < NB) {
#pragma unroll
                for (int u = 0; u < BF; ++u) B[u] = row[2 * ((b0 + 1) * BF + u)];
            }
            asm volatile("" ::: "memory");                       // the reads above are issued before the adds below
            rms_chain32(p, A);
            if (b0 + 1 < NB) {
                if (b0 + 2 < NB) {
#pragma unroll
                    for (int u = 0; u < BF; ++u) A[u] = row[2 * ((b0 + 2) * BF + u)];
                }
                asm volatile("" ::: "memory");
                rms_chain32(p, B);
            }
        }
#ifndef LMRS_SHFL_RMS_TAIL
        // lanes 0..7 hold the chains' sums: v_readlane_b32 (no LDS crossbar round trip as with __shfl); every lane then computes the same scalar
        const int pi = __float_as_int(p);                    // (the builtin is int -> int: a float argument would be CONVERTED)
        const float p0 = __int_as_float(__builtin_amdgcn_readlane(pi, 0)), p1 = __int_as_float(__builtin_amdgcn_readlane(pi, 1));
        const float p2 = __int_as_float(__builtin_amdgcn_readlane(pi, 2)), p3 = __int_as_float(__builtin_amdgcn_readlane(pi, 3));
        const float p4 = __int_as_float(__builtin_amdgcn_readlane(pi, 4)), p5 = __int_as_float(__builtin_amdgcn_readlane(pi, 5));
        const float p6 = __int_as_float(__builtin_amdgcn_readlane(pi, 6)), p7 = __int_as_float(__builtin_amdgcn_readlane(pi, 7));
#else
        const int wl = t & 48;                               // lanes 0..7 of this lane's own row hold the chains' sums
        const float p0 = __shfl(p, wl + 0), p1 = __shfl(p, wl + 1), p2 = __shfl(p, wl + 2), p3 = __shfl(p, wl + 3);
        const float p4 = __shfl(p, wl + 4), p5 = __shfl(p, wl + 5), p6 = __shfl(p, wl + 6), p7 = __shfl(p, wl + 7);
#endif
        if ((t & 63) == 0) {
            float ss = reduce_add8(p0, p1, p2, p3, p4, p5, p6, p7);
            // (N a power of two: the quotient and the product by 2^-k are the same real number, rounded once either way - no division sequence)
            if constexpr ((N & (N - 1)) == 0) ss = ss * (1.0f / (float)N);
            else ss = ss / (float)N;
            ss = ss + eps;
            ss = 1.0f / sqrtf(ss);
            scratch[8 * JP] = ss;
        }
    }
    lds_barrier();
    if (dbg && t == 0) dbg[5] = wall_clock64();          // serial chain done
    const float ss = scratch[8 * JP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float4 o;
        if (add_unit) {
            o.x = (1.0f + nw[i].x) * (ss * v[i].x); o.y = (1.0f + nw[i].y) * (ss * v[i].y);
            o.z = (1.0f + nw[i].z) * (ss * v[i].z); o.w = (1.0f + nw[i].w) * (ss * v[i].w);
        } else {
            o.x = nw[i].x * (ss * v[i].x); o.y = nw[i].y * (ss * v[i].y);
            o.z = nw[i].z * (ss * v[i].z); o.w = nw[i].w * (ss * v[i].w);
        }
        v[i] = o;
    }
}

// quantize (reference quantization.rs:44-67) of v[] into LDS: xq[N] int8, xs[N/128] f32.
// Every workgroup of a launch quantises the SAME vector, so whatever one wave does here is on the critical path of the launch - the
// exact redo of candidates near a rounding boundary included.  It is kept small: per pass, and inside a pass only the elements some
// lane flagged (usually one; round 3 redid the four elements of every lane of the pass, ~100 instructions, on 4 launches out of 5).
template <int N, int NTH = kBlk, class F = NoHook>
__device__ __forceinline__ void vec_quantize_q8(const float4 (&v)[(VecGeom<N, NTH>::NP)], int8_t* xq, float* xs, unsigned long long* dbg = nullptr, F landed = F()) {
    using V = VecGeom<N, NTH>;
    constexpr int NP = V::NP, NPG = V::NPG, NL = NP - NPG;         // grouped passes [0, NPG), linear passes [NPG, NP)
    const int t = threadIdx.x;
    // flat phases (all maxima, then all scales, then all candidates) so that the independent cross-lane reductions of the
    // grouped part and of every linear pass overlap instead of running one after the other
    float mg = 0.0f, ml[NL ? NL : 1];
#pragma unroll
    for (int i = 0; i < NPG; ++i) mg = absmax4(v[i], mg);          // the lane's NPG float4s belong to one group
#pragma unroll
    for (int i = NPG; i < NP; ++i) ml[i - NPG] = V::live(i, t) ? absmax4(v[i], 0.0f) : 0.0f;
    landed();                                                     // the activation has arrived in this wave
    if constexpr (NPG > 0) mg = cluster_max<V::LG>(mg);           // wmax of each 128-group (max is order-free)
#pragma unroll
    for (int i = 0; i < NL; ++i) ml[i] = group32_max(ml[i]);
    if (dbg && t == 0) dbg[6] = wall_clock64();
    // scale: stored, must be the IEEE quotient (div127_sane); inv: 1 ulp suffices (quant_q8_cand)
    float scg = div127_sane(mg), scl[NL ? NL : 1];
    const float invg = __builtin_amdgcn_rcpf(scg);
    float invl[NL ? NL : 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) { scl[i] = div127_sane(ml[i]); invl[i] = __builtin_amdgcn_rcpf(scl[i]); }
    // (Round 4, measured and removed: candidates by magic-number add - t = x * inv + 1.5 * 2^23 leaves rint(x * inv) in t's low byte, packed
    // v_pk_mul_f32 / v_pk_add_f32, no rint, no convert; 0 mismatches in 6e8 cases (quant_check), bit-equal on the GPU, and 419 -> 422 us
    // per step: a packed f32 instruction issues in twice the time of a scalar-per-lane one on this chip, the count halved buys nothing.)
    int q[NP][4];
    float4 d[NP];
    float dev[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float inv = i < NPG ? invg : invl[i < NPG ? 0 : i - NPG];
        q[i][0] = quant_q8_cand(v[i].x, inv, d[i].x); q[i][1] = quant_q8_cand(v[i].y, inv, d[i].y);
        q[i][2] = quant_q8_cand(v[i].z, inv, d[i].z); q[i][3] = quant_q8_cand(v[i].w, inv, d[i].w);
        dev[i] = absmax4(d[i], 0.0f);
    }
    // groups whose maximum is zero / denormal / huge / NaN: the division, then every element the reference's way.  (Wave-uniform branch,
    // operand behind an opaque asm: as a plain select hipcc hoists the division sequence onto the fast path.)
    const bool insane_g = NPG > 0 && !quant_group_sane(mg);
    bool insane_l[NL ? NL : 1];
    bool any_insane = insane_g;
#pragma unroll
    for (int i = 0; i < NL; ++i) { insane_l[i] = !quant_group_sane(ml[i]); any_insane = any_insane || insane_l[i]; }
    if (__any(any_insane)) {
        float mm = mg; asm volatile("" : "+v"(mm));
        if (insane_g) scg = mm / 127.0f;
#pragma unroll
        for (int i = 0; i < NL; ++i) { float ml2 = ml[i]; asm volatile("" : "+v"(ml2)); if (insane_l[i]) scl[i] = ml2 / 127.0f; }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const bool insane = i < NPG ? insane_g : insane_l[i < NPG ? 0 : i - NPG];
        const float sc = i < NPG ? scg : scl[i < NPG ? 0 : i - NPG];
        if (insane || dev[i] > kQuantDevMax) {
            if (__any(insane || fabsf(d[i].x) > kQuantDevMax)) q[i][0] = quant_q8(v[i].x, sc);
            if (__any(insane || fabsf(d[i].y) > kQuantDevMax)) q[i][1] = quant_q8(v[i].y, sc);
            if (__any(insane || fabsf(d[i].z) > kQuantDevMax)) q[i][2] = quant_q8(v[i].z, sc);
            if (__any(insane || fabsf(d[i].w) > kQuantDevMax)) q[i][3] = quant_q8(v[i].w, sc);
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (V::live(i, t)) {
            const int e = V::elem(i, t);
            const unsigned lo = __builtin_amdgcn_perm((unsigned)q[i][1], (unsigned)q[i][0], 0x0c0c0400u);
            const unsigned hi = __builtin_amdgcn_perm((unsigned)q[i][3], (unsigned)q[i][2], 0x0c0c0400u);
            *reinterpret_cast<unsigned*>(xq + e) = lo | (hi << 16);
            if (i >= NPG && (t & 31) == 0) xs[e >> 7] = scl[i < NPG ? 0 : i - NPG];
        }
    }
    if (NPG > 0 && (t % V::LG) == 0) xs[t / V::LG] = scg;
}

// quantize_q4 (reference quantization.rs:69-95) of v[] into LDS: xq4[N/2] packed nibbles, xs[N/128] f32.
// The LDS copy keeps the reference's packing (even element = low nibble) but stores every nibble XOR 8: (q - 8), the
// value the reference multiplies (functional.rs:236-240), is the 4-bit two's-complement number with exactly that bit
// pattern, so v_dot8_i32_i4 on (weights ^ 0x88888888, this) is the group's integer sum with no unpacking at all.
// Candidates without the per-element division as in vec_quantize_q8: n = rint(x * inv + 8.0); lanes whose value lies
// within 4e-5 of a rounding boundary, and abnormal groups, redo the reference arithmetic (quant_q4_cand, lmrs_device_math.h).
template <int N, int NTH = kBlk, class F = NoHook>
__device__ __forceinline__ void vec_quantize_q4(const float4 (&v)[(VecGeom<N, NTH>::NP)], int8_t* xq4, float* xs, unsigned long long* dbg = nullptr, F landed = F()) {
    using V = VecGeom<N, NTH>;
    constexpr int NP = V::NP, NPG = V::NPG, NL = NP - NPG;
    const int t = threadIdx.x;
    float mg = 0.0f, ml[NL ? NL : 1];
#pragma unroll
    for (int i = 0; i < NPG; ++i) mg = absmax4(v[i], mg);
#pragma unroll
    for (int i = NPG; i < NP; ++i) ml[i - NPG] = V::live(i, t) ? absmax4(v[i], 0.0f) : 0.0f;
    landed();
    if constexpr (NPG > 0) mg = cluster_max<V::LG>(mg);
#pragma unroll
    for (int i = 0; i < NL; ++i) ml[i] = group32_max(ml[i]);
    if (dbg && t == 0) dbg[6] = wall_clock64();
    const float scg = mg / -8.0f, invg = __builtin_amdgcn_rcpf(scg);      // (a power of two: hipcc multiplies)
    float scl[NL ? NL : 1], invl[NL ? NL : 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) { scl[i] = ml[i] / -8.0f; invl[i] = __builtin_amdgcn_rcpf(scl[i]); }
    unsigned q[NP][4];
    float4 d[NP];
    float dev[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float inv = i < NPG ? invg : invl[i < NPG ? 0 : i - NPG];
        q[i][0] = quant_q4_cand(v[i].x, inv, d[i].x); q[i][1] = quant_q4_cand(v[i].y, inv, d[i].y);
        q[i][2] = quant_q4_cand(v[i].z, inv, d[i].z); q[i][3] = quant_q4_cand(v[i].w, inv, d[i].w);
        dev[i] = absmax4(d[i], 0.0f);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float m = i < NPG ? mg : ml[i < NPG ? 0 : i - NPG], sc = i < NPG ? scg : scl[i < NPG ? 0 : i - NPG];
        const bool insane = !quant_group_sane(m);
        if (insane || dev[i] > kQuantDevMax) {
            if (__any(insane || fabsf(d[i].x) > kQuantDevMax)) q[i][0] = quant_q4(v[i].x, sc);
            if (__any(insane || fabsf(d[i].y) > kQuantDevMax)) q[i][1] = quant_q4(v[i].y, sc);
            if (__any(insane || fabsf(d[i].z) > kQuantDevMax)) q[i][2] = quant_q4(v[i].z, sc);
            if (__any(insane || fabsf(d[i].w) > kQuantDevMax)) q[i][3] = quant_q4(v[i].w, sc);
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (V::live(i, t)) {
            const int e = V::elem(i, t);
            const unsigned packed = (q[i][0] | (q[i][1] << 4) | (q[i][2] << 8) | (q[i][3] << 12)) ^ 0x8888u;
            *reinterpret_cast<unsigned short*>(xq4 + (e >> 1)) = (unsigned short)packed;
            if (i >= NPG && (t & 31) == 0) xs[e >> 7] = scl[i < NPG ? 0 : i - NPG];
        }
    }
    if (NPG > 0 && (t % V::LG) == 0) xs[t / V::LG] = scg;
}

// ------------------------------------------------------------------------------------------------
// Weight tile: the 16-byte steps one lane holds for one row-pass.
//   L lanes per row; CL lanes (a "cluster") cover one 128-element group: 8 x 16 B for Q8_0, 4 x 16 B for Q4_0;
//   NC = L/CL clusters, each owning a contiguous segment of U = G/NC groups.
// ------------------------------------------------------------------------------------------------
template <int N, int L, int NTH = kBlk, bool Q4 = false> struct RowGeom {
    static constexpr int G = N / 128, CL = Q4 ? 4 : 8, NC = L / CL, U = G / NC;
    static constexpr int ROWB = Q4 ? N / 2 : N;                    // bytes per weight row
    static constexpr int RW = 64 / L, RB = RW * (NTH / 64);        // rows per wave / per workgroup pass
    static_assert(L % CL == 0 && G % NC == 0, "row groups must divide over the clusters");
    static_assert(U >= 1 && U <= 24, "a cluster's groups must fit one tile");
    // Q4_0: a group is only 64 bytes, half a cache line.  Two clusters form a PAIR (8 aligned lanes) that owns a contiguous segment
    // of 2 * U groups and takes them alternately (even groups: first cluster, odd: second), so that one load instruction reads
    // 8 lanes x 16 B = one whole 128-byte line per pair.  (With a contiguous segment per cluster every instruction touched twice
    // as many lines, half of each: the Q4_0 streams ran at half the byte rate of the Q8_0 ones.)
    static_assert(!Q4 || (L % 8 == 0), "Q4_0: lanes per row in pairs of clusters");
    static constexpr int WR = Q4 ? L - 8 : L - CL;                // first lane of the row that holds the finished sum
};

// The running sum of a row's cluster j - 1 (8 lanes, every lane holds it) for cluster j, without an LDS round trip: inside a 16-lane
// row the lane 8 below (row_shr:8); across rows lane 15 of the row below (row_bcast:15), from row 1 to row 2 lane 31 (row_bcast:31).
// j = index of the RECEIVING cluster inside its row of L lanes (1 .. L/8 - 1).  ds_bpermute (__shfl) cost ~100 cycles per hop on the
// critical path of every row: 7 hops for the 64-lane rows of w2.
template <int L> __device__ __forceinline__ float cluster_carry(float acc, const int j) {      // j: a constant after unrolling - the branches fold
    const int first = j * 8;                                      // first lane (within the row of L) of the receiving cluster
    if (first % 16 == 8) return dpp_f<0x118>(acc);                // row_shr:8
    if (L == 64 && first == 32) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x143, 0xF, 0xF, false));   // row_bcast:31 (rows 0,1 -> 2,3)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x142, 0xF, 0xF, false));                            // row_bcast:15 (lane 15 of the row below)
}

template <int U> struct WTile { i32x4 w[U]; float sc[U]; };

// steps [U0, U1) of the tile (the whole tile by default)
template <int N, int L, bool Q4 = false, int U0 = 0, int U1 = RowGeom<N, L, kBlk, Q4>::U>
__device__ __forceinline__ void tile_issue(WTile<RowGeom<N, L, kBlk, Q4>::U>& t, const int8_t* __restrict__ wq, const float* __restrict__ ws, int row) {
    using R = RowGeom<N, L, kBlk, Q4>;
    const int lane = threadIdx.x & 63, r = lane % L;
    if constexpr (!Q4) {
        const int g0 = (r / R::CL) * R::U, rc = r % R::CL;
        const LMRS_GLOBAL i32x4* wrow = as_global(reinterpret_cast<const i32x4*>(wq + (size_t)row * R::ROWB)) + g0 * R::CL + rc;
        const LMRS_GLOBAL float* srow = as_global(ws) + (size_t)row * R::G + g0;
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            t.w[u] = __builtin_nontemporal_load(wrow + u * R::CL);
            t.sc[u] = srow[u];
        }
    } else {                                                       // pair-interleaved: step u = group seg0 + 2u + sub
        const int seg0 = (r / 8) * 2 * R::U, sub = (r >> 2) & 1, rc = r & 3;
        const LMRS_GLOBAL i32x4* wrow = as_global(reinterpret_cast<const i32x4*>(wq + (size_t)row * R::ROWB)) + (seg0 + sub) * 4 + rc;
        const LMRS_GLOBAL float* srow = as_global(ws) + (size_t)row * R::G + seg0 + sub;
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            t.w[u] = __builtin_nontemporal_load(wrow + u * 8);
            t.sc[u] = srow[2 * u];
        }
    }
}

// -> the row's result, valid in lane R::WR of the row (Q8_0: the whole last cluster; Q4_0: the first cluster of the last pair).
template <int N, int L, bool Q4 = false>
__device__ __forceinline__ float tile_consume(const WTile<RowGeom<N, L, kBlk, Q4>::U>& t, const int8_t* xq, const float* xs) {
    using R = RowGeom<N, L, kBlk, Q4>;
    const int lane = threadIdx.x & 63, r = lane % L, cl = r / R::CL, g0 = cl * R::U, rc = r % R::CL;
    float pb[R::U];
    if constexpr (Q4) {
        // pair-interleaved layout (RowGeom): this lane's step u is group seg0 + 2u + sub; the chain runs in the pair's first cluster,
        // which takes its own product, then its neighbour's (DPP row_shl:4, fetched while every lane is active), group by group
        const int rp = r / 8, seg0 = rp * 2 * R::U, sub = (r >> 2) & 1;
        float nb[R::U];
#pragma unroll
        for (int u = 0; u < R::U; ++u) {
            const int g = seg0 + 2 * u + sub;
            const i32x4 x = *reinterpret_cast<const i32x4*>(xq + (g * 4 + rc) * 16);
            int d = __builtin_amdgcn_sdot8(t.w[u].x ^ (int)0x88888888, x.x, 0, false);
            d = __builtin_amdgcn_sdot8(t.w[u].y ^ (int)0x88888888, x.y, d, false);
            d = __builtin_amdgcn_sdot8(t.w[u].z ^ (int)0x88888888, x.z, d, false);
            d = __builtin_amdgcn_sdot8(t.w[u].w ^ (int)0x88888888, x.w, d, false);
            d += dpp_i<0xB1>(d); d += dpp_i<0x4E>(d);             // the 4 lanes of the cluster
            const float p = (float)d * t.sc[u];                    // (ival as f32) * w.s[..]
            pb[u] = p * xs[g];                                     //   * x.s[..]
        }
#pragma unroll
        for (int u = 0; u < R::U; ++u) nb[u] = dpp_f<0x104>(pb[u]);   // row_shl:4: the odd group's product, into the chain lanes
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < L / 8; ++j) {
            const float carry = j == 0 ? 0.0f : __shfl(acc, (lane & ~(L - 1)) + (j - 1) * 8);
            if (rp == j && sub == 0) {
                acc = carry;
#pragma unroll
                for (int u = 0; u < R::U; ++u) { acc = acc + pb[u]; acc = acc + nb[u]; }      // xout += ..., groups ascending
            }
        }
        return acc;
    }
#pragma unroll
    for (int u = 0; u < R::U; ++u) {
        const i32x4 x = *reinterpret_cast<const i32x4*>(xq + ((g0 + u) * R::CL + rc) * 16);
        int d;
        if constexpr (!Q4) {
            d = __builtin_amdgcn_sdot4(t.w[u].x, x.x, 0, false);
            d = __builtin_amdgcn_sdot4(t.w[u].y, x.y, d, false);
            d = __builtin_amdgcn_sdot4(t.w[u].z, x.z, d, false);
            d = __builtin_amdgcn_sdot4(t.w[u].w, x.w, d, false);
            d = cluster8_sum(d);
        } else {                                      // (nibble - 8) of both operands = signed 4-bit value of nibble ^ 8
            d = __builtin_amdgcn_sdot8(t.w[u].x ^ (int)0x88888888, x.x, 0, false);
            d = __builtin_amdgcn_sdot8(t.w[u].y ^ (int)0x88888888, x.y, d, false);
            d = __builtin_amdgcn_sdot8(t.w[u].z ^ (int)0x88888888, x.z, d, false);
            d = __builtin_amdgcn_sdot8(t.w[u].w ^ (int)0x88888888, x.w, d, false);
            d += dpp_i<0xB1>(d); d += dpp_i<0x4E>(d);  // the 4 lanes of the cluster
        }
        float p = (float)d * t.sc[u];                 // (ival as f32) * w.s[..]
        pb[u] = p * xs[g0 + u];                       //   * x.s[..]
    }
    float acc = 0.0f;
    if constexpr (R::NC == 1) {
#pragma unroll
        for (int u = 0; u < R::U; ++u) acc = acc + pb[u];                    // xout += ..., groups ascending
    } else {
#pragma unroll
        for (int j = 0; j < R::NC; ++j) {
            float carry = 0.0f;
#ifdef LMRS_SHFL_CARRY                                               // (A/B build: the ds_bpermute hops of rounds 1-3)
            if (j > 0) carry = __shfl(acc, (lane & ~(L - 1)) + (j - 1) * R::CL);
#else
            if (j > 0) carry = cluster_carry<L>(acc, j);
#endif
            if (cl == j) {
                acc = carry;
#pragma unroll
                for (int u = 0; u < R::U; ++u) acc = acc + pb[u];
            }
        }
    }
    return acc;
}

// GELU(gate) * up, reference transformer.rs:607-616 (f32 cubic, f64 tanh)
__device__ __forceinline__ float geglu(float gate, float up) {
    float cube = 0.044715f * gate; cube = cube * gate; cube = cube * gate;
    const float inner = gate + cube;
    const double th = tanh(0.7978845608028654 * (double)inner);
    const float g = 0.5f * (1.0f + (float)th);
    float val = gate * g;
    val = val * up;
    return val;
}

// SiLU(gate) * up, reference transformer.rs:617-620
__device__ __forceinline__ float swiglu(float gate, float up) {
    const float e = expf_glibc(-gate);
    const float g = 1.0f / (1.0f + e);
    float val = gate * g;
    val = val * up;
    return val;
}

// same, expf table fetched by lane shuffle: every lane of the wave must call it
__device__ __forceinline__ float swiglu_t(float gate, float up, uint64_t lane_tab) {
    const float e = expf_glibc_t(-gate, lane_tab);
    const float g = 1.0f / (1.0f + e);
    float val = gate * g;
    val = val * up;
    return val;
}

}  // namespace lmrs
