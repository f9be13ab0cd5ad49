#!/bin/bash
# Round 4: Farneback row stream with 12-row steps (ring of 24 rows) against 6-row steps: parity of both, rates
O=gpurun_out/r4_farn17; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 600 python -m pytest tests/test_farneback_gpu.py -x -q > $O/pytest_rb6.log 2>&1; tail -2 $O/pytest_rb6.log
DFX_LIBRARY=$PWD/build/variants/rb12w3/libdfx.so timeout 600 python -m pytest tests/test_farneback_gpu.py tests/test_edge_sizes_gpu.py -x -q > $O/pytest_rb12.log 2>&1; tail -2 $O/pytest_rb12.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in rb6 rb12w4 rb12w3 rb6 rb12w4 rb12w3; do
  L=build/variants/$m; [ $m = rb6 ] && L=denseflow_amd/lib
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 0 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*\|"last_flow_checksum":"[0-9a-f]*"' | paste - - - >> $O/rates.txt
done
cat $O/rates.txt
