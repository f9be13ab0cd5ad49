#!/bin/bash
# round 2, GPU call 18: the multi-rank GPU path of bench.py on a 1-GPU box (two processes, gloo rendezvous, both ranks on
# device 0 via DFX_BENCH_SHARE_GPU=1): weak (--split none) and strong (--split clip) lines.  NOT a scaling measurement.
mkdir -p gpurun_out/r2r; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r
cd $R
export DFX_BENCH_SHARE_GPU=1
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --frames 120 ) > $O/two_ranks_weak.log 2>&1; echo "weak rc=$?"; tail -1 $O/two_ranks_weak.log | cut -c1-600
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 2 --warmup 1 --frames 120 --split clip ) > $O/two_ranks_strong.log 2>&1; echo "strong rc=$?"; tail -1 $O/two_ranks_strong.log | cut -c1-600
unset DFX_BENCH_SHARE_GPU
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --steps 2 --warmup 1 --frames 120 --no-cpu-baseline --no-pcie ) > $O/one_rank_launcher.log 2>&1; echo "one rank rc=$?"; tail -1 $O/one_rank_launcher.log | cut -c1-300
