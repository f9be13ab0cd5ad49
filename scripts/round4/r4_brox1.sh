#!/bin/bash
# Round 4: Brox stage 1 with 8-byte loads for the adjacent taps of a bilinear sample: parity, rate against the old library
O=gpurun_out/r4_brox1; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_brox_gpu.py -x -q > $O/pytest_brox.log 2>&1; tail -2 $O/pytest_brox.log
python scripts/make_raw_clip.py 1920 1080 2 66 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for m in new old new old; do
  L=denseflow_amd/lib; [ $m = old ] && L=build/variants/brox_old
  echo -n "$m " >> $O/rates.txt
  LD_LIBRARY_PATH=$L ./build/dfx_prof brox 1920 1080 /tmp/clip1080.raw 66 1 2 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"device_ms_per_pair":[0-9.]*\|"last_flow_checksum":"[0-9a-f]*"' | paste - - - >> $O/rates.txt
done
cat $O/rates.txt
