#!/bin/bash
# Round 4: 128-column strips for wide levels vs 64-column strips everywhere: parity, rate
O=gpurun_out/r4_farn14; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py tests/test_segments_gpu.py tests/test_edge_sizes_gpu.py -x -q > $O/pytest_farn.log 2>&1; tail -3 $O/pytest_farn.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in wide narrow wide narrow; do
  L=denseflow_amd/lib; [ $m = narrow ] && L=build/variants/narrow
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 0 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*\|"last_flow_checksum":"[0-9a-f]*"' | paste - - - >> $O/rates.txt
done
cat $O/rates.txt
