#!/bin/bash
# round 5, GPU call 3: the PNG scheme on the device, the fixed tests, fuse_k with the new arithmetic, the shell's stages
O=gpurun_out/r5_3; mkdir -p $O; export TMPDIR=/tmp
cd /root/repo
make -s host > $O/make_host.log 2>&1
timeout 900 python -m pytest tests/test_png_planes_gpu.py tests/test_device_math_gpu.py tests/test_bench_shaped_batch_gpu.py tests/test_host_shell.py tests/test_quant_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for k in 3 4 5 6; do
  python bench.py --fuse-k $k --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others --no-parity > $O/bench_k$k.json 2> $O/bench_k$k.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_3/bench_k$k.json").read().strip().splitlines()[-1])
print("fuse_k $k", round(d["value"],1), "pairs/s  launch us", round(d["roofline"]["avg_launch_us"],1), "noop", d["config"]["noop_step_fraction"])
PY
done
python scripts/round5/e2e_stages.py 1920 1080 1537 farn jpg > $O/e2e_stages_farn_jpg.log 2>&1; cat $O/e2e_stages_farn_jpg.log
python scripts/round5/e2e_stages.py 1920 1080 513 farn png > $O/e2e_stages_farn_png_device.log 2>&1; tail -4 $O/e2e_stages_farn_png_device.log
python scripts/round5/e2e_stages.py 1920 1080 513 farn png DF_HOST_PNG=1 > $O/e2e_stages_farn_png_host.log 2>&1; tail -4 $O/e2e_stages_farn_png_host.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; grep "leg " $O/bench_default.err; wc -c $O/bench_default.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_3/bench_default.json").read().strip().splitlines()[-1])
print("tvl1", d["value"], d["roofline"]["frac"], d.get("parity_check"))
c=d["config"]; print({k:c[k] for k in c if "png" in k or "hard" in k or "noexit" in k})
for leg in c["other_workloads"]: print(leg.get("key"), leg.get("pairs_per_s"), leg.get("error"), leg.get("pcie_inclusive"))
PY
