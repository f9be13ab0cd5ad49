#!/bin/bash
# round 3, GPU session 1: host-link probe, the changed async/bench paths, the new bench line
O=gpurun_out/r3a; mkdir -p $O
build/pcie_probe > $O/pcie_probe.txt 2>&1; cat $O/pcie_probe.txt
timeout 900 python -m pytest tests/test_async_gpu.py tests/test_bench_two_ranks_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json
