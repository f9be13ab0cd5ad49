#!/bin/bash
# round 2, GPU call 17: (a) do the 15 planes of a TVL1 pair collide on HBM channels?  plane stride + 0 / 64 / 448 / 1088
# floats; (b) one bounded attempt at the SQ counters of the Farneback iteration kernel (hung twice before)
mkdir -p gpurun_out/r2q; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q
cd $R
( SWEEP="0:4:0:0:3:0:1:16:0,0:4:0:0:3:0:1:16:64,0:4:0:0:3:0:1:16:448,0:4:0:0:3:0:1:16:1088,0:4:0:0:3:0:1:16:0,0:4:0:0:3:0:1:16:448" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_skew.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sweep_skew.log | cut -c1-420
cd /tmp
( ALGO=farn SWEEP="0:0:8:0" timeout -s KILL 75 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_farn -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 9 ) > $O/sq_farn_A.log 2>&1; echo "sq farn A rc=$?"
python $R/scripts/sq_summary.py $O/sq_farn k_farn_iteration > $O/sq_farn_final_A.json 2>>$O/sq_farn_A.log; rm -rf $O/sq_farn; head -24 $O/sq_farn_final_A.json
( ALGO=farn SWEEP="0:0:8:0" timeout -s KILL 75 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $O/sq_farnB -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 9 ) > $O/sq_farn_B.log 2>&1; echo "sq farn B rc=$?"
python $R/scripts/sq_summary.py $O/sq_farnB k_farn_iteration > $O/sq_farn_final_B.json 2>>$O/sq_farn_B.log; rm -rf $O/sq_farnB; head -24 $O/sq_farn_final_B.json
