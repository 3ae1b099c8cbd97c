#!/bin/bash
# round 4, final artefacts: every profile of tools/make_profiles.sh under the final sources, then the whole GPU test tier
cd "$GRAFT_REPO_ROOT"
bash tools/make_profiles.sh r4 > gpurun_out/make_profiles_r4.log 2>&1; tail -12 gpurun_out/make_profiles_r4.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_r4_final.txt 2>&1; tail -5 gpurun_out/pytest_r4_final.txt
ls gpurun_out/art
