#!/bin/bash
# round 2, GPU call 4: wave-priority experiment on the one-tile kernel; persistent kernel diagnostics (grid, SQ counters)
mkdir -p gpurun_out/r2d; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d
cd $R
for P in 0 1 2; do
( DFX_TVL1_PRIO=$P EPS=1e-9 SWEEP="0:4:0:0" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/noconv_prio$P.log 2>&1; echo "noconv prio$P rc=$?"; grep -v amdgpu.ids $O/noconv_prio$P.log
( DFX_TVL1_PRIO=$P SWEEP="0:4:0:0" timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_prio$P.log 2>&1; grep -v amdgpu.ids $O/sweep_prio$P.log
done
for G in 0 1 2 3 4; do
( DFX_VERBOSE=1 DFX_TVL1_MAP=1 DFX_TVL1_PERS_WGS=$G EPS=1e-9 SWEEP="3:4:0:0" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/noconv_pers_wgs$G.log 2>&1; echo "pers wgs=$G rc=$?"; grep -v amdgpu.ids $O/noconv_pers_wgs$G.log
done
( DFX_VERBOSE=1 DFX_TVL1_MAP=1 DFX_TVL1_PERS_WGS=3 EPS=1e-9 SWEEP="3:4:0:322" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/noconv_pers322_wgs3.log 2>&1; grep -v amdgpu.ids $O/noconv_pers322_wgs3.log
cd /tmp
for V in "3:4:16:0 pers8" "3:4:16:322 pers4" "0:4:16:0 tile"; do set -- $V
  ( DFX_TVL1_MAP=1 EPS=1e-9 SWEEP="$1" timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_$2 -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 17 ) > $O/sq_$2.log 2>&1; echo "sq $2 rc=$?"
  python $R/scripts/sq_summary.py $O/sq_$2 step_ > $O/sq_$2.json 2>>$O/sq_$2.log; rm -rf $O/sq_$2; cat $O/sq_$2.json
done
