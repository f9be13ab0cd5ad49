#!/bin/bash
# round 2, GPU call 1: parity of the packed-math TVL1 step kernel, A/B sweep, SQ counters old vs new
mkdir -p gpurun_out/r2a; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a
cd $R
( timeout 600 python -m pytest tests/test_tvl1_gpu.py tests/test_edge_sizes_gpu.py -m gpu -x -q ) > $O/pytest_tvl1.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_tvl1.log
( SWEEP="2:4:0:0,0:4:0:0,0:4:0:48,0:5:0:48,0:6:0:48,0:3:0:0,0:5:0:0,0:6:0:0,0:4:64:0" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_1080p.log 2>&1; echo "sweep rc=$?"; cat $O/sweep_1080p.log
( DFX_LIBRARY=$R/denseflow_amd/lib/variants/libdfx_hypbranch.so SWEEP="0:4:0:0,0:4:0:48" timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_1080p_hypbranch.log 2>&1; echo "sweep hypbranch rc=$?"; cat $O/sweep_1080p_hypbranch.log
( SWEEP="2:4:0:0,0:4:0:0,0:4:0:16,0:4:0:24,0:3:0:0" timeout 600 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep_224.log 2>&1; echo "sweep224 rc=$?"; cat $O/sweep_224.log
cd /tmp
for V in "2:4:16:0 old" "0:4:16:0 new"; do set -- $V
 for P in "A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "B SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  T=${P%% *}; C=${P#* }
  ( SWEEP="$1" timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq_$2_$T -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 17 ) > $O/sq_$2_$T.log 2>&1; echo "sq $2 $T rc=$?"
  python $R/scripts/sq_summary.py $O/sq_$2_$T step_fused > $O/sq_$2_$T.json 2>>$O/sq_$2_$T.log; rm -rf $O/sq_$2_$T
 done
done
cat $O/sq_old_A.json $O/sq_new_A.json | head -80
