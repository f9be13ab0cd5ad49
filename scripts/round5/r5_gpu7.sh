#!/bin/bash
# round 5, GPU call 7: fuse_k on content that keeps the fine levels iterating (HardClip, no early exit)
O=gpurun_out/r5_7; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
for k in 2 3 4 5 6; do
  python bench.py --clip hard --frames 66 --fuse-k $k --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/hard_k$k.json 2> $O/hard_k$k.err
  python bench.py --tvl1-epsilon 0 --frames 33 --fuse-k $k --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/noexit_k$k.json 2> $O/noexit_k$k.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_7/hard_k$k.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/r5_7/noexit_k$k.json").read().strip().splitlines()[-1])
print("fuse_k $k  hard", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1), " noexit", round(b["value"],2), "launch us", round(b["roofline"]["avg_launch_us"],1))
PY
done
