#!/bin/bash
# round 2, GPU call 19: two PROCESSES sharing the GPU reached 426 pairs/s where two handles in one process reached 400.
# How far does that go (1-4 processes), and can one process get it (more hardware queues)?
mkdir -p gpurun_out/r2s; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s
cd $R
export DFX_BENCH_SHARE_GPU=1
for N in 1 2 3 4 2; do
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N bench.py --gpus $N --steps 3 --warmup 1 --frames 120 --no-cpu-baseline --no-pcie ) > $O/procs_$N.log 2>&1; echo "procs=$N rc=$? $(tail -1 $O/procs_$N.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "pairs/s", d["config"]["pairs_per_step"], "pairs/step")')"
done
unset DFX_BENCH_SHARE_GPU
( LANES="1,2,3,2" timeout 300 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_default.log 2>&1; grep -v amdgpu.ids $O/lanes_default.log | cut -c1-160
( GPU_MAX_HW_QUEUES=8 LANES="1,2,3,2" timeout 300 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_hwq8.log 2>&1; echo "GPU_MAX_HW_QUEUES=8"; grep -v amdgpu.ids $O/lanes_hwq8.log | cut -c1-160
( GPU_MAX_HW_QUEUES=1 LANES="1,2" timeout 300 python scripts/multi_lane_probe.py 1920 1080 240 ) > $O/lanes_hwq1.log 2>&1; echo "GPU_MAX_HW_QUEUES=1"; grep -v amdgpu.ids $O/lanes_hwq1.log | cut -c1-160
