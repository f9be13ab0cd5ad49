#!/bin/bash
# Round 4: row-stream Farneback kernel: only R0 + flows a step ahead (windows loaded when used) at 4 / 5 workgroups per CU
O=gpurun_out/r4_farn11; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in full nowin4 nowin5 full nowin4 nowin5; do
  L=build/variants/$m; [ $m = full ] && L=denseflow_amd/lib
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 0 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*\|"last_flow_checksum":"[0-9a-f]*"' | paste - - - >> $O/rates.txt
done
cat $O/rates.txt
