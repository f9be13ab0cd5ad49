#!/bin/bash
# Round 4: the host shell end to end on the final tree (device JPEG; single pipeline = spinning waits by default)
O=gpurun_out/r4_e2e2; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
CONFIGS=device ALGOS=farn,tvl1 timeout 600 python scripts/e2e_cli_rate.py 1920 1080 1537 > $O/e2e_1080p_1537frames.log 2>&1; tail -3 $O/e2e_1080p_1537frames.log
CONFIGS=device ALGOS=farn,tvl1 timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 > $O/e2e_224x64clips.log 2>&1; tail -3 $O/e2e_224x64clips.log
