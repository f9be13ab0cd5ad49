#!/bin/bash
# round 5, GPU call 5: Brox SOR with a guarded LDS tile (A/B against the clamped-index build), JPEG landing buffers, helper thread
O=gpurun_out/r5_5; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
make -s host > $O/make_host.log 2>&1
timeout 900 python -m pytest tests/test_brox_gpu.py tests/test_jpeg_gpu.py tests/test_async_gpu.py tests/test_segments_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for r in 1 2; do for v in guard noguard; do
  L=""; [ $v = noguard ] && L=/root/repo/build/variants/libdfx_brox_noguard.so
  DFX_LIBRARY=$L python bench.py --algo brox --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_brox_$v.json 2> $O/bench_brox_$v.err
  DFX_LIBRARY=$L python bench.py --algo brox --width 3840 --height 2160 --frames 34 --step 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_brox4k_$v.json 2>> $O/bench_brox_$v.err
  python - <<PY
import json
a=json.loads(open("gpurun_out/r5_5/bench_brox_$v.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/r5_5/bench_brox4k_$v.json").read().strip().splitlines()[-1])
print("$v 1080p", round(a["value"],1), "launch us", round(a["roofline"]["avg_launch_us"],1), " 4K -s=2", round(b["value"],2))
PY
done; done
python scripts/round5/e2e_stages.py 1920 1080 1537 farn jpg > $O/e2e_stages_farn_jpg.log 2>&1; grep "run\|stages" $O/e2e_stages_farn_jpg.log
python scripts/e2e_trace.py 1920 1080 1537 farn 2>&1 | grep "calc:" | head -6
