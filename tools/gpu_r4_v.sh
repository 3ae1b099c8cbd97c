#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4v; mkdir -p $O
timeout 300 python tools/timeline.py gemma-2-2b 30 q4_0 > $O/tl_gemma.txt 2>&1
tail -34 $O/tl_gemma.txt
