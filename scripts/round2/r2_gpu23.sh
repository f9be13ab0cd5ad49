#!/bin/bash
# round 2, GPU call 23: end-to-end host-shell rates and the 3840x2160 TVL1 / Farneback lines with the final kernels
mkdir -p gpurun_out/r2w; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w
cd $R
( ALGOS=tvl1,farn timeout 400 python scripts/e2e_cli_rate.py 1920 1080 513 ) > $O/e2e_1080p.log 2>&1; echo "e2e 1080p rc=$?"; grep -v amdgpu.ids $O/e2e_1080p.log
( ALGOS=tvl1 timeout 400 python scripts/e2e_cli_rate.py 224 224 300 32 ) > $O/e2e_224.log 2>&1; echo "e2e 224 rc=$?"; grep -v amdgpu.ids $O/e2e_224.log
timeout -s KILL 300 python bench.py --algo tvl1 --width 3840 --height 2160 --frames 34 --steps 2 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 > $O/bench_tvl1_4k.json; echo "tvl1 4k rc=$?"; cut -c1-200 $O/bench_tvl1_4k.json
timeout -s KILL 300 python bench.py --algo farn --width 3840 --height 2160 --frames 34 --steps 2 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 > $O/bench_farn_4k.json; echo "farn 4k rc=$?"; cut -c1-200 $O/bench_farn_4k.json
