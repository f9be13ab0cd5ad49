#!/bin/bash
# round 2, GPU call 11: the new default variants (TVL1 tile geometry 3, Brox one-barrier SOR with dead-band skipping,
# Farneback zero-weight tap skipping + 16-row polynomial expansion): full GPU suite under the new defaults (the variant
# tests run every A/B mode), then in-process A/B sweeps (bit-identity checked against the first configuration).
mkdir -p gpurun_out/r2k; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k
cd $R
( timeout 900 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
# TVL1 1080p: geometry 0 / 1 / 2 / 3, twice (fields impl:K:B:TH:geom)
( SWEEP="0:4:0:0:0,0:4:0:0:1,0:4:0:0:2,0:4:0:0:3,0:4:0:0:0,0:4:0:0:3" SWEEP_LEVELS=1 timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_tvl1_geom.log 2>&1; echo "tvl1 geom rc=$?"; grep -v amdgpu.ids $O/sweep_tvl1_geom.log | cut -c1-420
# TVL1 1080p, whole 300-frame clip: batch 129 (default) / 150 / 299
( SWEEP="0:4:0:0:3,0:4:150:0:3,0:4:299:0:3,0:4:0:0:3" timeout 300 python scripts/sweep_tvl1.py 1920 1080 300 ) > $O/sweep_tvl1_batch.log 2>&1; echo "tvl1 batch rc=$?"; grep -v amdgpu.ids $O/sweep_tvl1_batch.log | cut -c1-220
( SWEEP="0:4:0:0:0,0:4:0:0:3,0:4:0:0:0,0:4:0:0:3" timeout 200 python scripts/sweep_tvl1.py 224 224 300 ) > $O/sweep_tvl1_224.log 2>&1; echo "tvl1 224 rc=$?"; grep -v amdgpu.ids $O/sweep_tvl1_224.log | cut -c1-220
# Farneback 1080p: (skip0, polyrows) = (0,0) (1,0) (0,16) (1,16), twice the ends (fields ...:geom:sor:skip0:polyrows)
( ALGO=farn SWEEP="0:0:0:0:0:0:0:0,0:0:0:0:0:0:1:0,0:0:0:0:0:0:0:16,0:0:0:0:0:0:1:16,0:0:0:0:0:0:0:0,0:0:0:0:0:0:1:16" timeout 300 python scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sweep_farn.log 2>&1; echo "farn rc=$?"; grep -v amdgpu.ids $O/sweep_farn.log | cut -c1-200
# Brox 1080p and 3840x2160: SOR mode 0 / 1 / 2
( ALGO=brox SWEEP="0:0:0:0:0:0,0:0:0:0:0:1,0:0:0:0:0:2,0:0:0:0:0:0,0:0:0:0:0:2" timeout 300 python scripts/sweep_tvl1.py 1920 1080 33 ) > $O/sweep_brox.log 2>&1; echo "brox rc=$?"; grep -v amdgpu.ids $O/sweep_brox.log | cut -c1-200
( ALGO=brox SWEEP="0:0:0:0:0:0,0:0:0:0:0:2,0:0:0:0:0:1" timeout 300 python scripts/sweep_tvl1.py 3840 2160 9 ) > $O/sweep_brox_4k.log 2>&1; echo "brox 4k rc=$?"; grep -v amdgpu.ids $O/sweep_brox_4k.log | cut -c1-200
# headline line under the new defaults (no CPU leg here: call 12 runs the full default command)
timeout -s KILL 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_tvl1_1080p.json; echo "bench rc=$?"; cut -c1-700 $O/bench_tvl1_1080p.json
