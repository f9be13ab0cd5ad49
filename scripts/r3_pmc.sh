#!/bin/bash
# round 3: PMC HBM traffic of the final kernels (batch 16 like round 2: rocprofv3 --pmc segfaults within a second on the
# 130-frame / batch-129 workload, gpurun_out/final/pmc.log), plus one batch-64 pass for Farneback as a linearity check
O=gpurun_out/final; mkdir -p $O
ALGOS="tvl1 farn brox" PMC_BATCH=16 PMC_TIMEOUT=240 bash scripts/gpu_pmc.sh > $O/pmc16.log 2>&1; tail -3 $O/pmc16.log; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_batch16.json
ALGOS="farn" PMC_BATCH=64 PMC_TIMEOUT=240 bash scripts/gpu_pmc.sh > $O/pmc64.log 2>&1; tail -3 $O/pmc64.log; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_farn_batch64.json
python - <<'PY'
import json
a=json.load(open('gpurun_out/final/pmc_traffic_batch16.json')); b=json.load(open('gpurun_out/final/pmc_traffic_farn_batch64.json'))
for k,v in a.items(): print(k, 'MB per pair and launch', round(v['hbm_bytes_per_launch_per_pair']/1e6,2), round(v.get('companion_hbm_bytes_per_launch_per_pair',0)/1e6,2))
for k,v in b.items(): print('batch64', k, round(v['hbm_bytes_per_launch_per_pair']/1e6,2))
PY
