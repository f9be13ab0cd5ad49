#!/bin/bash
# Round 4: does an initial stagger between the two workgroups of a CU break the phase lockstep of the fused iteration kernel?
O=gpurun_out/r4_farn7; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in full stag6000 stag14000 stag24000 full stag6000 stag14000 stag24000; do
  if [ $m = full ]; then L=denseflow_amd/lib; else L=build/variants/$m; fi
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 0 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*' | paste - - >> $O/rates.txt
done
cat $O/rates.txt
