#!/bin/bash
# Round 4: parity, rates (both kernels alternating), SQ + TA counters of the default kernel; torch-free harness, batch 129
O=gpurun_out/r4_farn5; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py -x -q > $O/pytest_farn.log 2>&1; tail -3 $O/pytest_farn.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for v in 0 32 16 0 32 16; do ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 $v >> $O/rates.txt 2>> $O/err.log; done
grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*\|"last_flow_checksum":"[0-9a-f]*"' $O/rates.txt | paste - - -
R=/root/repo
run() { n=$1; v=$2; shift 2
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n farn_iter > $O/$n.json 2>&1; echo "== $n"; grep -v "^{\|^}\|^ }" $O/$n.json | head -20; rm -rf $O/$n; }
for v in ${VARIANTS:-0}; do
  run sq_v$v $v SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
  run ta_v$v $v TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum
done
