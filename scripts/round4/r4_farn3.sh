#!/bin/bash
# Round 4: parity of the Farneback kernels, then rate + SQ counters of the default iteration kernel (torch-free harness, batch 129)
O=gpurun_out/r4_farn3; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py -x -q > $O/pytest_farn.log 2>&1; tail -3 $O/pytest_farn.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for v in 0 16 0 16; do ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 $v >> $O/rates.txt 2>> $O/err.log; done
cat $O/rates.txt
R=/root/repo
for v in ${VARIANTS:-0}; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/$O/sq_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/sq_v$v.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/$O/sq2_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/sq2_v$v.log 2>&1
  for d in sq_v$v sq2_v$v; do echo "== $d"; python scripts/sq_summary.py $O/$d farn_iter > $O/$d.json 2>&1; head -c 1200 $O/$d.json; rm -rf $O/$d; done
done
