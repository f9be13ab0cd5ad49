#!/bin/bash
# round 2, GPU call 3: persistent prefetching TVL1 step kernel — parity, sweep, non-converging full-step timing
mkdir -p gpurun_out/r2c; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c
cd $R
( timeout 600 python -m pytest tests/test_tvl1_gpu.py -m gpu -x -q -k "fused_kernel" ) > $O/pytest_tvl1.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_tvl1.log
( SWEEP="0:4:0:0,3:4:0:0,3:4:0:322,3:3:0:0,3:5:0:0,3:6:0:0,3:5:0:322" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_1080p.log 2>&1; echo "sweep rc=$?"; cat $O/sweep_1080p.log
( DFX_TVL1_MAP=1 SWEEP="3:4:0:0,3:4:0:322" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_1080p_map1.log 2>&1; echo "sweep map1 rc=$?"; cat $O/sweep_1080p_map1.log
( EPS=1e-9 SWEEP="3:4:0:0,3:4:0:322" SWEEP_LEVELS=1 timeout 600 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_noconv.log 2>&1; echo "noconv rc=$?"; cat $O/sweep_noconv.log
( SWEEP="0:4:0:0,3:4:0:0,3:4:0:322" timeout 600 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep_224.log 2>&1; echo "sweep224 rc=$?"; cat $O/sweep_224.log
