#!/bin/bash
# round 2, GPU call 25: BASELINE configs 5 and 4 at their stated sizes on one GPU: 3840x2160 x 300 frames -a=brox -s=2
# (device-resident bench line), and one GPU's share of the 512-clip list (64 clips 224x224 x 300 frames through the CLI)
mkdir -p gpurun_out/r2x; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2x
cd $R
timeout -s KILL 400 python bench.py --algo brox --width 3840 --height 2160 --step 2 --frames 300 --steps 1 --warmup 1 --no-cpu-baseline --no-pcie 2>$O/brox_4k_300.err | tail -1 > $O/bench_brox_4k_s2_300frames.json; echo "brox 4k x300 rc=$?"; cut -c1-330 $O/bench_brox_4k_s2_300frames.json; tail -2 $O/brox_4k_300.err | cut -c1-200
( ALGOS=tvl1 timeout 400 python scripts/e2e_cli_rate.py 224 224 300 64 ) > $O/e2e_224_64clips.log 2>&1; echo "e2e 64 clips rc=$?"; grep -v amdgpu.ids $O/e2e_224_64clips.log
