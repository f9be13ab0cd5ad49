#!/bin/bash
# Round 4: the row-stream Farneback iteration kernel: segment length (generations of workgroups per launch)
O=gpurun_out/r4_farn9; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in full wps4 p1only p23only full wps4 p1only p23only; do
  L=build/variants/$m; v=0
  [ $m = full ] && L=denseflow_amd/lib
  [ $m = tile ] && { L=denseflow_amd/lib; v=32; }
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 $v 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*' | paste - - >> $O/rates.txt
done
cat $O/rates.txt
