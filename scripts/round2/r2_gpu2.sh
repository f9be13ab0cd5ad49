#!/bin/bash
# round 2, GPU call 2: tile variants, Infinity-Cache sub-batching probe, memory-only / compute-only builds
mkdir -p gpurun_out/r2b; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b
cd $R
( timeout 600 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q -k "fused_kernel or bit_exact" ) > $O/pytest_tvl1.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_tvl1.log
( SWEEP="0:4:0:0,0:4:0:488,0:5:0:488,0:6:0:488,0:4:0:324,0:3:0:488" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_1080p.log 2>&1; echo "sweep rc=$?"; cat $O/sweep_1080p.log
( NSCALES=1 SWEEP="0:4:129:0,0:4:64:0,0:4:32:0,0:4:16:0,0:4:12:0,0:4:8:0,0:4:6:0,0:4:4:0,0:4:8:488,0:4:16:488" timeout 600 python scripts/sweep_tvl1.py 786 442 130 ) > $O/sweep_L4_subbatch.log 2>&1; echo "sweep L4 rc=$?"; cat $O/sweep_L4_subbatch.log
( NSCALES=1 SWEEP="0:4:129:0,0:4:8:0,0:4:4:0,0:4:2:0,0:4:1:0" timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_L0_subbatch.log 2>&1; echo "sweep L0 rc=$?"; cat $O/sweep_L0_subbatch.log
( EPS=1e-9 SWEEP="0:4:0:0,0:4:0:488" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_noconv.log 2>&1; echo "noconv rc=$?"; cat $O/sweep_noconv.log
for D in 1 2; do
( EPS=1e-9 DFX_LIBRARY=$R/denseflow_amd/lib/variants/libdfx_dbg$D.so SWEEP="0:4:0:0,0:4:0:488" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_dbg$D.log 2>&1; echo "dbg$D rc=$?"; cat $O/sweep_dbg$D.log
done
