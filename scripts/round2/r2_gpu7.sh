#!/bin/bash
# round 2, GPU call 7: split tests, shell A/B with the collector thread, bench legs, PMC traffic + SQ counters for all three
mkdir -p gpurun_out/r2g; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g
cd $R
( timeout 900 python -m pytest tests/test_host_shell.py tests/test_async_gpu.py -m gpu -q ) > $O/pytest_shell.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_shell.log
for S in 0 1 0 1; do
 if [ $S = 1 ]; then export DF_SYNC_FLOW=1; else unset DF_SYNC_FLOW; fi
 ( ALGOS=tvl1,farn timeout 600 python scripts/e2e_cli_rate.py 1920 1080 513 ) > $O/e2e_1080p_sync$S.log 2>&1; echo "e2e 1080p sync=$S"; grep "device" $O/e2e_1080p_sync$S.log
 ( ALGOS=tvl1 timeout 600 python scripts/e2e_cli_rate.py 224 224 300 32 ) > $O/e2e_224_sync$S.log 2>&1; echo "e2e 224 sync=$S"; grep "device" $O/e2e_224_sync$S.log
done
unset DF_SYNC_FLOW
( timeout 900 python bench.py ) > $O/bench_tvl1.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_tvl1.log | cut -c1-600
ALGOS="tvl1 farn brox" bash scripts/gpu_pmc.sh > $O/gpu_pmc.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json; tail -40 $O/gpu_pmc.log | cut -c1-200
cd /tmp
for A in farn brox; do
 for P in "A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "B SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  T=${P%% *}; C=${P#* }
  ( timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq_${A}_$T -o p -- python $R/bench.py --algo $A --steps 1 --warmup 0 --frames 34 --max-batch 16 --no-cpu-baseline --no-pcie ) > $O/sq_${A}_$T.log 2>&1; echo "sq $A $T rc=$?"
  python $R/scripts/sq_summary.py $O/sq_${A}_$T k_farn_iteration k_brox_sor_fused k_brox_stage1 > $O/sq_${A}_$T.json 2>>$O/sq_${A}_$T.log; rm -rf $O/sq_${A}_$T
 done
done
cat $O/sq_farn_A.json | head -30
