/*
 * quant_check.c — TEST INFRASTRUCTURE.  Checks the division-free fast path of the device quantiser
 * (lm.rs_amd/csrc/lmrs_device_math.h: quant_q8_try + quant_group_sane) against the reference arithmetic
 * q = (x / scale).round() as i8  (reference src/quantization.rs:62-63) on random and adversarial inputs.
 * The fast path multiplies by 1/scale and falls back to the exact IEEE division whenever the product lies
 * within 1e-4 of a rounding boundary (k + 0.5), which covers the worst-case error of the product (3.1e-5 with a 1-ulp reciprocal), or the
 * group maximum is not a normal number in [1e-30, 1e30].  (Round 4: the window is 4e-5 - kQuantDevMax = 0.49996 - still above the 3.1e-5 bound.)
 * Also checks, exhaustively, the division-free group scale m / 127 (div127_sane).  |x| <= wmax as in a real group (x is an element of it).
 *   usage: ./quant_check        (prints the number of mismatches: must be 0)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int exact_q(float x, float scale) {
    float q = roundf(x / scale);
    if (q != q) return 0;
    if (q < -128.0f) q = -128.0f;
    if (q > 127.0f) q = 127.0f;
    return (int)q;
}
/* device form: the candidate (int)rint(x * inv) stands unless the lane raises `slow`; wmax is the group maximum */
static inline int fast_q(float x, float inv, float scale, float wmax) {
    int slow = !(wmax > 1.0e-30f && wmax < 1.0e30f);
    const float r = x * inv;
    const float n = rintf(r);
    slow |= fabsf(r - n) > 0.49996f;
    if (slow) return exact_q(x, scale);
    return (int)n;      /* v_cvt_i32_f32; |n| <= 127 here, asserted below */
}
/* Q4_0 activation quantiser (reference src/quantization.rs:69-95): scale = wmax / -8, nibble = clamp(round(x/scale + 8) as u8, 0, 15) */
static inline unsigned exact_q4(float x, float scale) {
    float q = roundf(x / scale + 8.0f);
    if (q != q) return 0;
    if (q < 0.0f) q = 0.0f;
    if (q > 15.0f) q = 15.0f;
    return (unsigned)q;
}
static inline unsigned fast_q4(float x, float inv, float scale, float wmax) {
    int slow = !(wmax > 1.0e-30f && wmax < 1.0e30f);
    const float s = x * inv + 8.0f;
    const float n = rintf(s);
    slow |= fabsf(s - n) > 0.49996f;
    if (slow) return exact_q4(x, scale);
    const unsigned q = (unsigned)(int)n;
    return q < 15u ? q : 15u;
}
static uint64_t s = 88172645463325252ull;
static inline uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
static inline float urand(void) { return (float)(rnd() >> 8) / 16777216.0f; }

int main(void) {
    unsigned long long bad = 0, n = 0;
    for (long it = 0; it < 400000000L; ++it) {
        float wmax, x;
        const uint32_t mode = rnd() & 7;
        if (mode < 5) { wmax = ldexpf(0.5f + urand(), (int)(rnd() % 40) - 30); x = (2.0f * urand() - 1.0f) * wmax; }
        else if (mode == 5) { wmax = ldexpf(0.5f + urand(), (int)(rnd() % 250) - 140); x = (2.0f * urand() - 1.0f) * wmax; }   /* huge / denormal scales */
        else {                                      /* adversarial: x right at a rounding boundary (k + 0.5) * scale, +- a few ulp */
            wmax = ldexpf(0.5f + urand(), (int)(rnd() % 40) - 30);
            const float sc = wmax / 127.0f; const int k = (int)(rnd() % 127);
            x = ((float)k + 0.5f) * sc; uint32_t u; memcpy(&u, &x, 4); u += (rnd() % 9) - 4; memcpy(&x, &u, 4);
            if (rnd() & 1) x = -x;
        }
        const float scale = wmax / 127.0f;
        const float inv = 1.0f / scale;
        if (fabsf(x) > wmax) x = copysignf(wmax, x);
        /* the device takes inv from v_rcp_f32 (1 ulp): every neighbour of the correctly rounded reciprocal must work */
        const int e = exact_q(x, scale);
        if (e != fast_q(x, inv, scale, wmax) || e != fast_q(x, nextafterf(inv, INFINITY), scale, wmax) || e != fast_q(x, nextafterf(inv, 0.0f), scale, wmax)) bad++;
        n++;
    }
    /* the same for Q4_0 */
    for (long it = 0; it < 200000000L; ++it) {
        float wmax, x;
        const uint32_t mode = rnd() & 7;
        if (mode < 5) { wmax = ldexpf(0.5f + urand(), (int)(rnd() % 40) - 30); x = (2.0f * urand() - 1.0f) * wmax; }
        else if (mode == 5) { wmax = ldexpf(0.5f + urand(), (int)(rnd() % 250) - 140); x = (2.0f * urand() - 1.0f) * wmax; }
        else {
            wmax = ldexpf(0.5f + urand(), (int)(rnd() % 40) - 30);
            const float sc = wmax / -8.0f; const int k = (int)(rnd() % 17) - 8;       /* x/scale + 8 = k + 8.5 +- a few ulp */
            x = ((float)k + 0.5f) * sc; uint32_t u; memcpy(&u, &x, 4); u += (rnd() % 9) - 4; memcpy(&x, &u, 4);
            if (rnd() & 1) x = -x;
        }
        if (fabsf(x) > wmax) x = copysignf(wmax, x);
        const float scale = wmax / -8.0f, inv = 1.0f / scale;
        const unsigned e = exact_q4(x, scale);
        if (e != fast_q4(x, inv, scale, wmax) || e != fast_q4(x, nextafterf(inv, -INFINITY), scale, wmax) || e != fast_q4(x, nextafterf(inv, 0.0f), scale, wmax)) bad++;
        n++;
    }
    /* degenerate groups */
    const float zs[] = {0.0f, -0.0f, 1e-45f, 3e-39f, INFINITY, NAN};
    for (unsigned i = 0; i < sizeof zs / sizeof *zs; ++i)
        for (unsigned j = 0; j < sizeof zs / sizeof *zs; ++j) { if (exact_q(zs[j], zs[i] / 127.0f) != fast_q(zs[j], 1.0f / (zs[i] / 127.0f), zs[i] / 127.0f, zs[i])) bad++; n++;
              if (exact_q4(zs[j], zs[i] / -8.0f) != fast_q4(zs[j], 1.0f / (zs[i] / -8.0f), zs[i] / -8.0f, zs[i])) bad++; n++; }
    /* the scale without the division sequence (lmrs_stage.h div127_sane): y = RN(1/127), q0 = m y, r = fma(-127, q0, m), q = fma(r, y, q0)
     * against m / 127.0f for EVERY float of the range quant_group_sane admits (and 16 neighbours either side) */
    {
        const float y = 0x1.020408p-7f, lo_f = 1.0e-30f, hi_f = 1.0e30f;
        uint32_t lo, hi; memcpy(&lo, &lo_f, 4); memcpy(&hi, &hi_f, 4);
        unsigned long long dbad = 0;
        for (uint32_t b = lo - 16; b <= hi + 16; ++b) {
            float m; memcpy(&m, &b, 4);
            const float q0 = m * y;
            const float r = fmaf(-127.0f, q0, m);
            const float q = fmaf(r, y, q0), e = m / 127.0f;
            if (memcmp(&q, &e, 4)) dbad++;
        }
        printf("div127: %u cases, %llu mismatches\n", hi - lo + 33, dbad);
        bad += dbad;
    }
    printf("cases: %llu  mismatches: %llu\n", n, bad);
    return bad != 0;
}
