/* tanh_check.c - test infrastructure (never linked or run by the product).
 *
 * Gemma-2's score soft-cap (reference src/transformer.rs:520-522), logit soft-cap (:377-379) and tanh-GELU (:614) call Rust's
 * f64::tanh, i.e. the host libm's tanh, and round the result to f32.  The HIP kernels call the device math library's f64 tanh.
 * Unlike expf (expf_check.c: the device RESTATES glibc's algorithm, so the restatement can be swept on the host over all 2^32
 * inputs), tanh is not restated: the two implementations can only be compared on the results.  This program checks a dump of
 * device results against the host libm:
 *
 *     python -c "import lmrs_amd, numpy as np; x = ...; np.stack([x, lmrs_amd.tanh_cast(x, C)], 1).astype('<f4').tofile('dump.bin')"
 *     oracle/tanh_check dump.bin C          # C = 1 (soft-caps) or 0.7978845608028654 (GELU)
 *
 * dump.bin = n records of (f32 x, f32 y_device); y must equal (float)tanh(C * (double)x) bit for bit (NaN matches NaN).
 * tests/test_gpu_parity.py::test_tanh_matches_host_libm runs the same comparison in-process on the GPU box over every f32 of the
 * four binades [0.25, 4) (both signs), every 61st f32 of the whole line and the edge cases: 0 mismatches on MI355X / ROCm 7.2 /
 * glibc 2.35 (the double results differ by an ulp now and then; none of the swept inputs sits close enough to an f32 rounding
 * boundary for that to survive the cast). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s <dump.bin of (f32 x, f32 y_device) records> <c>\n", argv[0]); return 2; }
    const double c = atof(argv[2]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    float rec[2 * 4096];
    size_t n = 0, bad = 0, got;
    while ((got = fread(rec, 8, 4096, f)) > 0) {
        for (size_t i = 0; i < got; i++) {
            const float x = rec[2 * i], y = rec[2 * i + 1], h = (float)tanh(c * (double)x);
            uint32_t yb, hb;
            memcpy(&yb, &y, 4); memcpy(&hb, &h, 4);
            if (yb != hb && !(isnan(y) && isnan(h))) {
                if (bad < 10) printf("x = %.9g (0x%08x): device %.9g host %.9g\n", x, *(uint32_t*)&rec[2 * i], y, h);
                ++bad;
            }
        }
        n += got;
    }
    fclose(f);
    printf("%zu inputs, %zu mismatches (c = %.17g)\n", n, bad, c);
    return bad != 0;
}
