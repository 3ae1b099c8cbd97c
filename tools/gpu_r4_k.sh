#!/bin/bash
# round 4, GPU call K: Q8_0 activation quantiser candidates by magic-number add (packed) against rint + convert
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4k; mkdir -p $O
timeout 600 python tools/ab_bench.py "rint+cvt@lm.rs_amd/liblmrs_hip_cvt.so" "magic-add" "rint+cvt(again)@lm.rs_amd/liblmrs_hip_cvt.so" "magic-add(again)" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 600 python tools/ab_bench.py --model llama-3.2-3b "rint+cvt@lm.rs_amd/liblmrs_hip_cvt.so" "magic-add" > $O/ab_3b.txt 2>&1; cat $O/ab_3b.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "quantize or matmul or golden or mini or random_geom or rounding" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
