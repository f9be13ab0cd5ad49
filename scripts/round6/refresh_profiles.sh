#!/bin/bash
# copy a closing run's files (gpurun_out/r6_final, scripts/round6/r6_final.sh) into profiles/round6 and recompute profiles/pmc_traffic.json
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_final; P=profiles/round6
cp $O/bench_default.json $O/bench_farn_1080p.json $O/bench_brox_1080p.json $O/bench_*_kernel_stats.csv $O/full_clip_parity.txt $O/pytest_gpu.log $O/smoke.log $O/fetch_brox.json $O/fetch_farn.json $O/write_brox.json $O/write_farn.json $P/
cp $O/fetch_tvl1_v*.json $O/write_tvl1_v*.json $O/head_ab_*.json $O/sqA_tvl1_*.json $O/sqB_tvl1_*.json $O/timeline_v*.md $O/timeline_v*_dispatches.csv $P/tvl1_head/
cp $O/brox_sync_*.json $O/brox_4k_*.json $P/brox/
python3 - <<'PY'
import json
P='profiles/'
d=json.load(open(P+'pmc_traffic.json'))
f=json.load(open(P+'round6/fetch_brox.json')); w=json.load(open(P+'round6/write_brox.json'))
ks=[k for k in f if 'k_brox' in k]
n=sum(f[k]['dispatches'] for k in ks); F=sum(f[k]['dispatches']*f[k].get('FETCH_SIZE',0) for k in ks); W=sum(w[k]['dispatches']*w[k].get('WRITE_SIZE',0) for k in ks)
b=d['brox']; b['launches']=n; b['FETCH_SIZE_KB_per_launch']=F/n; b['WRITE_SIZE_KB_per_launch']=W/n
b['hbm_bytes_per_launch']=(2*F+W)*1024/n; b['hbm_bytes_per_launch_per_pair']=b['hbm_bytes_per_launch']/65
f2=json.load(open(P+'round6/fetch_farn.json')); w2=json.load(open(P+'round6/write_farn.json'))
ks=[k for k in f2 if 'k_farn_iter_stream' in k]
n=sum(f2[k]['dispatches'] for k in ks); F=sum(f2[k]['dispatches']*f2[k]['FETCH_SIZE'] for k in ks); W=sum(w2[k]['dispatches']*w2[k]['WRITE_SIZE'] for k in ks)
a=d['farn']; a['launches']=n; a['FETCH_SIZE_KB_per_launch']=F/n; a['WRITE_SIZE_KB_per_launch']=W/n; a['hbm_bytes_per_launch']=(2*F+W)*1024/n; a['hbm_bytes_per_launch_per_pair']=a['hbm_bytes_per_launch']/129
f3=json.load(open(P+'round6/tvl1_head/fetch_tvl1_v0.json')); w3=json.load(open(P+'round6/tvl1_head/write_tvl1_v0.json'))
t=d['tvl1']
for key,pre in (('void k_tvl1_step_fused<true, 0>',''),('void k_tvl1_warp_head<0>','companion_')):
    t[pre+'launches']=f3[key]['dispatches']; t[pre+'FETCH_SIZE_KB_per_launch']=f3[key]['FETCH_SIZE']; t[pre+'WRITE_SIZE_KB_per_launch']=w3[key]['WRITE_SIZE']
    v=(2*f3[key]['FETCH_SIZE']+w3[key]['WRITE_SIZE'])*1024
    if pre=='': t['hbm_bytes_per_launch']=v
    t[pre+'hbm_bytes_per_launch_per_pair']=v/129
json.dump(d,open(P+'pmc_traffic.json','w'),indent=1)
print('tvl1', t['hbm_bytes_per_launch_per_pair']/1e6, t['companion_hbm_bytes_per_launch_per_pair']/1e6, 'farn', a['hbm_bytes_per_launch_per_pair']/1e6, 'brox', b['hbm_bytes_per_launch_per_pair']/1e6)
PY
