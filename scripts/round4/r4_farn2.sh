#!/bin/bash
# Round 4, GPU call 3: the torch-free harness (tools/dfx_prof) under rocprofv3 at the bench's own batch (129 pairs):
# recomputed-M Farneback iteration kernel vs the M-in-HBM kernel — rates, SQ counters, HBM traffic (separate --pmc passes).
O=gpurun_out/r4_farn2; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
ls -la /tmp/clip1080.raw > $O/clip.txt
for v in 0 16 0 16; do ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 $v >> $O/rates.txt 2>> $O/err.log; done
cat $O/rates.txt
R=/root/repo
for v in 0 16; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/stats_v$v.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/$O/sq_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/sq_v$v.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/$O/sq2_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/sq2_v$v.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/fetch_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/fetch_v$v.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/write_v$v -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/write_v$v.log 2>&1
done
for d in sq_v0 sq_v16 sq2_v0 sq2_v16 fetch_v0 fetch_v16 write_v0 write_v16; do echo "== $d"; python scripts/sq_summary.py $O/$d farn_iter > $O/$d.json 2>&1; head -c 1500 $O/$d.json; done
# keep only the summaries (CSV dumps are large)
for d in stats_v0 stats_v16; do find $O/$d -name "*kernel_stats.csv" -exec cp {} $O/$d.kernel_stats.csv \; ; done
rm -rf $O/sq_v0 $O/sq_v16 $O/sq2_v0 $O/sq2_v16 $O/fetch_v0 $O/fetch_v16 $O/write_v0 $O/write_v16 $O/stats_v0 $O/stats_v16
