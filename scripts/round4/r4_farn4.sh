#!/bin/bash
# Round 4: memory-pipeline counters (TA / TCP / TCC) of the two Farneback iteration kernels, torch-free harness, batch 129
O=gpurun_out/r4_farn4; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -c . $O/counters_list.txt
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
R=/root/repo
run() { # name variant counters...
  n=$1; v=$2; shift 2
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 0 $v ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n farn_iter > $O/$n.json 2>&1; echo "== $n"; head -c 1500 $O/$n.json; rm -rf $O/$n
}
for v in 0 16; do
  run ta_v$v $v TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
  run tcp_v$v $v TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
  run tcc_v$v $v TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
done
