#!/bin/bash
# round 4, GPU call I: top-p with the filter on the device; full GPU tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sample or topp or chat" > $O/pytest_sampler.txt 2>&1; tail -5 $O/pytest_sampler.txt
timeout 300 python tools/sampler_rate.py > $O/sampler_rate.txt 2>&1; cat $O/sampler_rate.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
