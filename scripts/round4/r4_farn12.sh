#!/bin/bash
# Round 4: Farneback with the level initialisation inside the first iteration: parity, rate
O=gpurun_out/r4_farn12; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py tests/test_segments_gpu.py tests/test_edge_sizes_gpu.py -x -q > $O/pytest_farn.log 2>&1; tail -3 $O/pytest_farn.log
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
for v in 0 16 0 16; do ./build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 3 0 $v 2>> $O/err.log | grep -o '"pairs_per_s":[0-9.]*\|"avg_launch_us":[0-9.]*\|"kernel_launches":[0-9]*\|"last_flow_checksum":"[0-9a-f]*"' | paste - - - - >> $O/rates.txt; done
cat $O/rates.txt
python bench.py --algo farn --no-others --no-cpu-baseline --no-live-pmc --no-pcie 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench farn', d['value'], d['roofline']['avg_launch_us'])"
