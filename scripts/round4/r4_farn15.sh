#!/bin/bash
# Round 4: counters of the FINAL Farneback iteration kernel (row stream, fused init / merge), torch-free harness, batch 129
O=gpurun_out/r4_farn15; mkdir -p $O; export TMPDIR=/tmp; R=/root/repo
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
run() { n=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n farn_iter > $O/$n.json 2>&1; echo "== $n"; grep -v "^{\|^}\|^ }" $O/$n.json | head -24; rm -rf $O/$n; }
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run ta TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
