#!/bin/bash
# Round 4: JPEG tests (stream growth, capacity status), then the host shell end to end with blocking / spinning waits
O=gpurun_out/r4_e2e; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 600 python -m pytest tests/test_jpeg_gpu.py tests/test_async_gpu.py tests/test_segments_gpu.py -x -q > $O/pytest_jpeg.log 2>&1; tail -4 $O/pytest_jpeg.log
for spin in 0 1; do
  if [ $spin = 1 ]; then export DF_SPIN_WAIT=1; else unset DF_SPIN_WAIT; fi
  CONFIGS=device ALGOS=farn,tvl1 timeout 600 python scripts/e2e_cli_rate.py 1920 1080 513 > $O/e2e_1080p_spin$spin.log 2>&1
  tail -4 $O/e2e_1080p_spin$spin.log
done
unset DF_SPIN_WAIT
CONFIGS=device ALGOS=tvl1 timeout 600 python scripts/e2e_cli_rate.py 224 224 300 64 > $O/e2e_224x64.log 2>&1; tail -3 $O/e2e_224x64.log
