// lmrs_device_math.h — bit-exact device arithmetic shared by every kernel of the decode path.
//
// Everything here is written so that the GPU reproduces, bit for bit, what the reference's CPU
// code computes (reference src/functional.rs, src/quantization.rs); the file is compiled with
// -ffp-contract=off, so a*b+c below is two rounded operations unless __builtin_fma is spelled out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lmrs {

// ---------------------------------------------------------------- cross-lane (wave64, DPP)
// quad_perm[1,0,3,2] / quad_perm[2,3,0,1] / row_half_mirror / row_mirror: no LDS traffic.
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }

// Sum over each aligned cluster of 8 lanes; exact (integers); every lane of the cluster gets the total.
__device__ __forceinline__ int cluster8_sum(int v) {
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    return v;
}
// v = max(v, v of another lane) as ONE instruction: the DPP operand of the max itself.  (Spelled fmaxf(v, dpp_f(v)) hipcc emits three -
// v_mov_b32_dpp, a canonicalising v_max v, v, v of the moved value, then the max: the quantiser's group maxima were 68 of the 249
// vector instructions of w2's prologue.)  IEEE v_max_f32: a NaN operand is dropped, as f32::max does (quantization.rs:52).
#ifndef LMRS_NO_ASM_MAX
#define LMRS_DPP_MAX(name, ctrl)                                                                                       \
    __device__ __forceinline__ float name(float v) {                                                                   \
        float r;                                                                                                       \
        /* s_nop 1: a DPP operand must not be read sooner than two wait states after a VALU wrote it, and hipcc's hazard    */ \
        /* recogniser does not know this asm reads through DPP (it put ONE state between two of these: wrong maxima)         */ \
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));               \
        return r;                                                                                                      \
    }
#else
#define LMRS_DPP_MAX(name, ctrl) __device__ __forceinline__ float name(float v);
#endif
LMRS_DPP_MAX(max_quad1, "quad_perm:[1,0,3,2]")
LMRS_DPP_MAX(max_quad2, "quad_perm:[2,3,0,1]")
LMRS_DPP_MAX(max_half_mirror, "row_half_mirror")
LMRS_DPP_MAX(max_mirror, "row_mirror")
#ifdef LMRS_NO_ASM_MAX
__device__ __forceinline__ float max_quad1(float v) { return fmaxf(v, dpp_f<0xB1>(v)); }
__device__ __forceinline__ float max_quad2(float v) { return fmaxf(v, dpp_f<0x4E>(v)); }
__device__ __forceinline__ float max_half_mirror(float v) { return fmaxf(v, dpp_f<0x141>(v)); }
__device__ __forceinline__ float max_mirror(float v) { return fmaxf(v, dpp_f<0x140>(v)); }
#endif
// max(|a.x|, |a.y|, |a.z|, |a.w|, m) in two instructions (the source modifiers are free; fmaxf(fabsf()) canonicalises every input first)
__device__ __forceinline__ float absmax4(const float4& a, float m) {
#ifndef LMRS_NO_ASM_MAX
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a.x), "v"(a.y), "v"(m));
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a.z), "v"(a.w), "v"(r));
    return r;
#else
    return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), m);
#endif
}
// Max over each aligned cluster of LG lanes (LG = 4 .. 32); every lane of the cluster gets it.
template <int LG> __device__ __forceinline__ float cluster_max(float v) {
    static_assert(LG == 4 || LG == 8 || LG == 16 || LG == 32, "cluster of 4, 8, 16 or 32 lanes");
    v = max_quad1(v);
    v = max_quad2(v);
    if constexpr (LG >= 8) v = max_half_mirror(v);
    if constexpr (LG >= 16) v = max_mirror(v);
    if constexpr (LG == 32) {
        // lanes l and l^16: gfx950's v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of
        // another (VALU, no LDS round trip): with both registers = v the results are [r0 r0 r2 r2] and [r1 r1 r3 r3]
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    return v;
}
// Max over each aligned group of 32 lanes (one 128-element quantisation group at 4 elements per lane).
__device__ __forceinline__ float group32_max(float v) { return cluster_max<32>(v); }

// Max over the whole wave: the two 32-lane halves meet through gfx950's v_permlane32_swap (VALU, no LDS round trip).
__device__ __forceinline__ float wave64_max(float v) {
    v = group32_max(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// o + p[0] + p[STRIDE] + p[2*STRIDE] + ... strictly left to right over n terms rounded up to a multiple of 16 (the caller
// pads with +-0.0, exact for an accumulator that starts at +0.0).  The adds are one dependent chain; the LDS reads are
// not: batches of 16 ping-pong so that the next batch is in flight while the current one is added.
template <int STRIDE>
__device__ __forceinline__ void lds_batch16(float (&d)[16], const float* p) {
    if constexpr (STRIDE == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 t = *reinterpret_cast<const float4*>(p + u * 4);
            d[u * 4] = t.x; d[u * 4 + 1] = t.y; d[u * 4 + 2] = t.z; d[u * 4 + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) d[u] = p[u * STRIDE];
    }
}
// The read-ahead is unconditional: up to 31 elements past the rounded-up end are READ (never added), so the caller's
// LDS region must extend that far.  It pays for contiguous data (4 x 16-byte reads per batch: softmax, 1.28 -> 1.21 us);
// for strided data (16 reads per batch: the V chain) issuing the next batch ahead of the adds costs what it hides.
template <int STRIDE, bool AHEAD = true>
__device__ __forceinline__ float serial_sum16(float o, const float* p, int n) {
    if constexpr (!AHEAD) {                                  // plain batches: 16 reads, then 16 adds
        for (int t = 0; t < n; t += 16) {
            float A[16];
            lds_batch16<STRIDE>(A, p + t * STRIDE);
#pragma unroll
            for (int u = 0; u < 16; ++u) o = o + A[u];
        }
        return o;
    }
    float A[16], B[16];
    lds_batch16<STRIDE>(A, p);
    for (int t = 0;; t += 32) {
        lds_batch16<STRIDE>(B, p + (t + 16) * STRIDE);
        asm volatile("" : "+v"(o) : : "memory");             // the adds below cannot move above the reads
#pragma unroll
        for (int u = 0; u < 16; ++u) o = o + A[u];
        if (t + 16 >= n) break;
        lds_batch16<STRIDE>(A, p + (t + 32) * STRIDE);
        asm volatile("" : "+v"(o) : : "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) o = o + B[u];
        if (t + 32 >= n) break;
    }
    return o;
}

// carry + e[lane 0] + e[lane 1] + ... + e[lane 16 * nrows - 1], strictly in lane order, with no LDS traffic: the running sum lives in
// lane 15 of the current 16-lane row and takes the row's values one by one through the DPP operand of the add itself
// (v_add_f32_dpp ... row_shr:k fetches lane 15 - k: 16 dependent adds per row and nothing else); row_bcast:15 hands the sum to
// lane 15 of the next row.  Every lane of the wave must be active; nrows (1..4) wave-uniform.  The result is returned in every lane.
#define LMRS_DPP_ADD(k) "v_add_f32_dpp %0, %1, %0 row_shr:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define LMRS_DPP_ROW                                                                                                                  \
    LMRS_DPP_ADD(15) LMRS_DPP_ADD(14) LMRS_DPP_ADD(13) LMRS_DPP_ADD(12) LMRS_DPP_ADD(11) LMRS_DPP_ADD(10) LMRS_DPP_ADD(9) LMRS_DPP_ADD(8) \
    LMRS_DPP_ADD(7) LMRS_DPP_ADD(6) LMRS_DPP_ADD(5) LMRS_DPP_ADD(4) LMRS_DPP_ADD(3) LMRS_DPP_ADD(2) LMRS_DPP_ADD(1) "v_add_f32 %0, %1, %0"
__device__ __forceinline__ float wave_serial_sum(float carry, float e, int nrows) {
    float acc = carry;
    // (s_nop 1: a VGPR written by a VALU instruction needs two wait states before a DPP read - inline asm gets no hazard handling)
    asm volatile("s_nop 1\n\t" LMRS_DPP_ROW : "+v"(acc) : "v"(e));
    if (nrows > 1) {
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0x2 bank_mask:0xf\n\t" LMRS_DPP_ROW : "+v"(acc) : "v"(e));
        if (nrows > 2) {
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0x4 bank_mask:0xf\n\t" LMRS_DPP_ROW : "+v"(acc) : "v"(e));
            if (nrows > 3) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0x8 bank_mask:0xf\n\t" LMRS_DPP_ROW : "+v"(acc) : "v"(e));
        }
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 16 * nrows - 1));
}

// Workgroup barrier that orders LDS traffic only.  hipcc's __syncthreads() also waits for every
// outstanding global load (s_waitcnt vmcnt(0)), which would drain the weight stream that is deliberately
// left in flight across the activation prologue; the kernels below only ever hand LDS data across a
// barrier, so lgkmcnt(0) + s_barrier is sufficient (global results are consumed by the lane that loaded
// them, under the compiler's own counted vmcnt waits).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------- f32::exp == glibc expf
// Restatement of glibc >= 2.27 expf (sysdeps/ieee754/flt-32/e_expf.c; N = 32 table + cubic in double)
// as compiled for x86-64 CPUs with FMA (the ifunc'd __expf_fma build, where GCC fuses every a*b+c).
// oracle/expf_check.c compares exactly this operation sequence with the host libm over all 2^32
// inputs (0 mismatches in the build container); tests/test_gpu_ops.py repeats the comparison
// device-vs-host on the GPU box.  Used by softmax (functional.rs:133) and SiLU (transformer.rs:617).
__device__ const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// The 32-entry table costs a dependent global load per call; hot kernels instead keep entry (lane & 31) in a
// register pair per lane (exp2f_tab_lane(), loaded once at kernel start under the other loads) and fetch
// T[ki % 32] with two ds_bpermute lane reads (expf_glibc_t).  Same arithmetic, same bits.
__device__ __forceinline__ uint64_t exp2f_tab_lane() { return EXP2F_TAB[threadIdx.x & 31]; }

template <bool LANE_TAB>
__device__ __forceinline__ float expf_glibc_impl(float x, uint64_t lane_tab) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const uint32_t ux = __float_as_uint(x);
    const uint32_t abstop = (ux >> 20) & 0x7ff;
    // special cases (|x| >= 88 or NaN) are resolved by selection AFTER the main path so that every lane stays
    // active through the lane-table shuffles
    bool special = false; float sval = 0.0f;
    if (abstop >= 0x42b) {
        if (ux == 0xff800000u) { special = true; sval = 0.0f; }
        else if (abstop >= 0x7f8) { special = true; sval = x + x; }
        else if (x > 0x1.62e42ep6f) { special = true; sval = __uint_as_float(0x7f800000u); }
        else if (x < -0x1.9fe368p6f) { special = true; sval = 0.0f; }
    }
    const double xd = special ? 0.0 : (double)x;
    double kd = __builtin_fma(InvLn2N, xd, SHIFT);
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - SHIFT;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    uint64_t t;
    if constexpr (LANE_TAB) {
        const int src = (int)((threadIdx.x & 32) | (unsigned)(ki & 31));          // same half-wave, lane ki % 32
        const unsigned lo = (unsigned)__shfl((int)(unsigned)lane_tab, src), hi = (unsigned)__shfl((int)(unsigned)(lane_tab >> 32), src);
        t = ((uint64_t)hi << 32) | lo;
    } else t = EXP2F_TAB[ki % 32];
    t += ki << (52 - 5);
    const double s = __longlong_as_double((long long)t);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return special ? sval : (float)y;
}
__device__ __forceinline__ float expf_glibc(float x) { return expf_glibc_impl<false>(x, 0); }
// All lanes of the wave must call this together (it shuffles).
__device__ __forceinline__ float expf_glibc_t(float x, uint64_t lane_tab) { return expf_glibc_impl<true>(x, lane_tab); }

// ---------------------------------------------------------------- quantize primitives (quantization.rs:44-95)
// Rust `f32 as i8` after f32::round: saturating, NaN -> 0.
__device__ __forceinline__ int quant_q8(float x, float scale) {
    float q = roundf(x / scale);                              // IEEE divide; round half away from zero
    if (!(q == q)) return 0;
    q = fminf(fmaxf(q, -128.0f), 127.0f);
    return (int)q;
}
// Same result without the per-element IEEE division (12 dependent instructions each; the quantiser sits on the
// critical path of every GEMV prologue).  Branch-free candidate: n = rint(x * inv) with inv = 1/scale (one division
// per group, or v_rcp_f32: 1 ulp).  The product is at most 3.1e-5 away from the correctly rounded quotient when the group maximum is a
// normal number in [1e-30, 1e30] (then |x * inv| <= 127.00002, no clamp and no NaN can occur), so the candidate equals
// round(x / scale) unless the product lies within 1e-4 of a rounding boundary k + 0.5; those lanes - and every lane of
// a group whose maximum is zero / denormal / huge / NaN - are `quant_slow` and the caller redoes them with quant_q8.
// oracle/quant_check.c restates this and compares with the reference arithmetic: 0 mismatches in 4e8 random +
// adversarial cases.
__device__ __forceinline__ bool quant_group_sane(float wmax) { return wmax > 1.0e-30f && wmax < 1.0e30f; }
__device__ __forceinline__ int quant_q8_try(float x, float inv, float& dev) {   // dev: running max of |x*inv - rint|
    const float r = x * inv;
    const float n = rintf(r);
    dev = __builtin_fmaxf(dev, fabsf(r - n));
    return (int)n;
}
// m / 127 correctly rounded, without the 12-instruction IEEE division sequence: y = RN(1/127), q0 = m y, r = m - 127 q0 (exact, one
// FMA), q = q0 + r y (one FMA).  oracle/quant_check.c: equal to m / 127.0f for EVERY float in [1e-30, 1e30] (1.67e9 cases) - the range
// quant_group_sane admits; every other group goes through the slow path, which divides.
__device__ __forceinline__ float div127_sane(float m) {
#ifndef LMRS_NO_FAST_DIV127
    const float y = 0x1.020408p-7f;
    const float q0 = m * y;
    const float r = __builtin_fmaf(-127.0f, q0, m);
    return __builtin_fmaf(r, y, q0);
#else
    return m / 127.0f;
#endif
}

// (the window: the product is provably within 3.1e-5 of the quotient - 1 ulp of v_rcp_f32, the product's and the division's roundings at
// |q| <= 127 - so 4e-5 suffices; round 3 used 1e-4, and every hit is on the critical path of a launch: all workgroups quantise the same vector)
constexpr float kQuantDevMax = 0.49996f;
__device__ __forceinline__ bool quant_slow(float wmax, float dev) { return !quant_group_sane(wmax) || dev > kQuantDevMax; }
// candidate + its distance from the integer it was rounded to (the caller flags |d| > kQuantDevMax)
__device__ __forceinline__ int quant_q8_cand(float x, float inv, float& d) {
    const float r = x * inv;
    const float n = rintf(r);
    d = r - n;
    return (int)n;
}
// ((x/scale + 8.0).round() as u8).clamp(0, 15)
__device__ __forceinline__ unsigned quant_q4(float x, float scale) {
    float q = roundf(x / scale + 8.0f);
    if (!(q == q)) return 0u;
    q = fminf(fmaxf(q, 0.0f), 15.0f);
    return (unsigned)q;
}

// Q4_0 candidate, same scheme: n = rint(x * inv + 8.0); for a sane group |x * inv| <= 8.000002, the sum differs from the
// reference's fl(fl(x / scale) + 8.0) by < 3e-6, so n is the reference's result unless the sum lies within 1e-4 of k + 0.5.
__device__ __forceinline__ unsigned quant_q4_try(float x, float inv, float& dev) {
    const float s = x * inv + 8.0f;                              // -ffp-contract=off: product rounded, then the sum
    const float n = rintf(s);
    dev = __builtin_fmaxf(dev, fabsf(s - n));
    const unsigned q = (unsigned)(int)n;                         // n >= -0.0
    return q < 15u ? q : 15u;
}

__device__ __forceinline__ unsigned quant_q4_cand(float x, float inv, float& d) {
    const float s = x * inv + 8.0f;
    const float n = rintf(s);
    d = s - n;
    const unsigned q = (unsigned)(int)n;
    return q < 15u ? q : 15u;
}

// wide 0.7.x f32x8::reduce_add (AVX path): ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)).
__device__ __forceinline__ float reduce_add8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    const float q0 = v0 + v4, q1 = v1 + v5, q2 = v2 + v6, q3 = v3 + v7;
    const float d0 = q0 + q2, d1 = q1 + q3;
    return d0 + d1;
}

}  // namespace lmrs
