// lmrs_vision_att.h - what the CLIP tower's attention kernels share (lmrs_vision.inc: the score phase; lmrs_vision_att.hip: the others).
#pragma once

constexpr int kVisHS = 64, kVisQB = 64;
#ifndef LMRS_VIS_KC
#define LMRS_VIS_KC 24
#endif
constexpr int kVisKC = LMRS_VIS_KC;            // keys per score wave
__host__ __device__ static int vis_key_chunks(int T) { return (T + kVisKC - 1) / kVisKC; }
// (+ the per-chunk maxima, one row of 64 per chunk of kVisKC keys, and the 64 row sums)
__host__ __device__ static size_t vis_slab_floats(int T) { return (size_t)T * kVisQB + 8 * kVisHS * kVisQB + (size_t)(vis_key_chunks(T) + 1) * kVisQB; }

typedef float f32x16v __attribute__((ext_vector_type(16)));
// d0 / d1 <- 32 floats at p; k0 / k1 (the half-row about to be multiplied) ride through as operands so that its multiplies stay below the request
__device__ __forceinline__ void srow_request(f32x16v& d0, f32x16v& d1, const float* p, f32x16v& k0, f32x16v& k1) {
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40" : "=&s"(d0), "=&s"(d1), "+s"(k0), "+s"(k1) : "s"(p));
}
__device__ __forceinline__ void srow_first(f32x16v& d0, f32x16v& d1, const float* p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(p));
}
__device__ __forceinline__ void srow_wait(f32x16v& a, f32x16v& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

// ------------------------------------------------------------------------------------------------
// The attention as FOUR launches, one per phase (the phases of vis_attention_kernel already talk through the global slab), so that
// every phase gets a wave-granular grid: 320 eight-wave workgroups on 256 CUs leave 64 CUs with twice the work, the same waves as
// independent one-wave workgroups do not (score + output loops 127.7 -> 82 us, tools/ubench/bcast.hip).
//   scores  : one wave per (query block, head, crop, chunk of kVisKC = 24 keys) - scalar rows; S[key][query], chunk maximum M[chunk][query]
//             (round 6: with 64-key chunks the launch was 2880 waves, all resident from the start, three on most SIMDs and two on the rest - it took
//             three waves' time; 24-key chunks are 6912 shorter waves, more than fit at once, and the dispatcher refills SIMDs as they drain: 51.4 -> 43.7 us)
//   softmax : eight waves per (query block, head, crop) - maximum over the chunks, exp in place, the row's sequential sum -> SUM[query]
//             (the divide moves into the consumers: every weight is divided exactly once there, by the same two operands)
//   output  : one wave per (query block, head, crop, residue r, half of the dims) - lane sum r of matmul_rest over its 72 keys, weights S / SUM
//             (round 6, the same reason: 2304 whole-row waves -> 4608 half-row ones, 51.6 -> 45.7 us)
//   stray   : the last T % 64 queries (one, for T = 577) as workgroups of their own inside the softmax launch instead of a tenth block of dead lanes
//   tree    : the 8 lane sums' tree + the scalar tail (key 576), 16 dims per wave
// Same operations in the same order per value as vis_attention_kernel: bit-identical.
// ------------------------------------------------------------------------------------------------
struct VisSlab { float* S; float* P2; float* M; float* SUM; };
__device__ __forceinline__ VisSlab vis_slab(float* scratch, int crop, int n_heads, int head, int nqb, int qb, int T, int lane) {
    float* slab = scratch + (((size_t)crop * n_heads + head) * nqb + qb) * vis_slab_floats(T);
    VisSlab v;
    v.S = slab + lane; v.P2 = slab + (size_t)T * kVisQB + lane; v.M = slab + (size_t)T * kVisQB + 8 * kVisHS * kVisQB + lane;
    v.SUM = v.M + (size_t)vis_key_chunks(T) * kVisQB;
    return v;
}
