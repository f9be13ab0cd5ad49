#!/bin/bash
# Round 4: Farneback row-stream strips shifted so that a strip's footprint starts on a 128-byte line: parity, rate, HBM bytes
O=gpurun_out/r4_farn16; mkdir -p $O; export TMPDIR=/tmp; R=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py tests/test_segments_gpu.py tests/test_edge_sizes_gpu.py -x -q > $O/pytest_farn.log 2>&1; tail -3 $O/pytest_farn.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in shift noshift shift noshift; do
  L=denseflow_amd/lib; [ $m = noshift ] && L=build/variants/noshift
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 0 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*\|"last_flow_checksum":"[0-9a-f]*"' | paste - - - >> $O/rates.txt
done
cat $O/rates.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/$c -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 ) > $O/$c.log 2>&1
  python scripts/sq_summary.py $O/$c farn_iter > $O/$c.json 2>&1; grep -A2 "false" $O/$c.json | head -4; rm -rf $O/$c
done
