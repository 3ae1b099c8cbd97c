/*
 * expf_check.c — TEST INFRASTRUCTURE.  Exhaustively compares a restatement of glibc's expf
 * (the ARM "optimized routines" algorithm glibc ships since 2.27: N=32 table + cubic in double,
 * sysdeps/ieee754/flt-32/e_expf.c) with the libm expf of THIS host over every finite f32 input.
 *
 * Why: Rust's f32::exp (reference src/functional.rs:133, src/transformer.rs:617) lowers to the
 * system libm expf.  The HIP kernels restate that algorithm with the same double operations
 * (lm.rs_amd/csrc/lmrs_device_math.h: expf_glibc) so that softmax and SiLU are bit-identical to
 * the CPU path.  glibc selects an FMA build of expf on x86-64 CPUs with FMA (ifunc); the two
 * variants differ only in whether the three polynomial steps are fused.  This program reports
 * the number of mismatches of both variants, so the variant the device must use is measured,
 * not assumed.    usage: ./expf_check   (prints "fma: <mismatches>  nofma: <mismatches>")
 * Measured in the build container (Xeon, glibc 2.35): fma: 0  nofma: 2  -> the device uses the
 * fully fused form and is bit-identical to host expf for every f32 input.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const uint64_t T[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};

static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline double asdouble(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t asuint64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

static inline float expf_restated(float x, int use_fma) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
    if (abstop >= (asuint(88.0f) >> 20)) {
        if (asuint(x) == asuint(-INFINITY)) return 0.0f;
        if (abstop >= (asuint(INFINITY) >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double xd = (double)x;
    double z = InvLn2N * xd;
    /* GCC's -ffp-contract=fast fuses the product into BOTH of its uses in glibc's FMA build
       (measured: with only the polynomial fused, 2 of 2^32 inputs differ from libm). */
    double kd = use_fma ? fma(InvLn2N, xd, SHIFT) : z + SHIFT;
    uint64_t ki = asuint64(kd);
    kd -= SHIFT;
    double r = use_fma ? fma(InvLn2N, xd, -kd) : z - kd;
    uint64_t t = T[ki % 32];
    t += ki << (52 - 5);
    double s = asdouble(t);
    double y;
    if (use_fma) {
        z = fma(C0, r, C1);
        double r2 = r * r;
        y = fma(C2, r, 1.0);
        y = fma(z, r2, y);
    } else {
        z = C0 * r + C1;
        double r2 = r * r;
        y = C2 * r + 1.0;
        y = z * r2 + y;
    }
    y = y * s;
    return (float)y;
}

int main(void) {
    unsigned long long bad_fma = 0, bad_nofma = 0;
#pragma omp parallel for reduction(+ : bad_fma, bad_nofma) schedule(static)
    for (long long u = 0; u < (1LL << 32); u++) {
        uint32_t bits = (uint32_t)u; float x; memcpy(&x, &bits, 4);
        if (x != x) continue;                                   /* NaN payloads: not compared */
        float ref = expf(x);
        float a = expf_restated(x, 1), b = expf_restated(x, 0);
        if (asuint(a) != asuint(ref)) bad_fma++;
        if (asuint(b) != asuint(ref)) bad_nofma++;
    }
    printf("fma: %llu  nofma: %llu\n", bad_fma, bad_nofma);
    return 0;
}
