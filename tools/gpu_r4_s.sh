#!/bin/bash
# round 4, GPU call S: 256 x 128 GEMM tiles, CLIP softmax launch (balanced exps, pipelined chain)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4s; mkdir -p $O
for n in 512 256; do for v in 0 1; do echo "LMRS_GEMM_TM256=$v" >> $O/prefill.txt; LMRS_GEMM_TM256=$v timeout 300 python tools/prefill_rate.py llama-3.2-1b $n >> $O/prefill.txt 2>&1; done; done
LMRS_GEMM_TM256=0 timeout 300 python tools/vision_rate.py 2 24 > $O/vision_tm128.txt 2>&1; timeout 300 python tools/vision_rate.py 2 24 > $O/vision.txt 2>&1
cat $O/prefill.txt; grep tower $O/vision_tm128.txt $O/vision.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or prefill or vision or fill_kv or configs4 or image or processor or chat" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
