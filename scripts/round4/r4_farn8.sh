#!/bin/bash
# Round 4: the fused Farneback iteration kernel at ONE workgroup per CU (LDS padded) — do the two resident workgroups share or just coexist?
O=gpurun_out/r4_farn8; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in full onewg mask1 mask1_onewg mask6_onewg full onewg; do
  if [ $m = full ]; then L=denseflow_amd/lib; else L=build/variants/$m; fi
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 0 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*' | paste - - >> $O/rates.txt
done
cat $O/rates.txt
