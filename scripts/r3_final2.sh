#!/bin/bash
# round 3, closing GPU session on the final tree (after: Brox parity-split LDS tile, round-2 Brox kernel removed, JPEG encoders
# on libjpeg's integer transform, PNG writer on libpng's rules): GPU suite, smoke, the driver-shaped bench line, per-algorithm
# bench lines, kernel statistics, end-to-end shell rate, SQ counters of the Brox kernels
O=gpurun_out/final2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > $O/bench_tvl1_1080p.json 2> $O/bench_tvl1_1080p.err; echo "bench rc=$?"
timeout 300 python bench.py --algo farn --steps 3 --no-others > $O/bench_farn_1080p.json 2>/dev/null; echo "farn rc=$?"
timeout 300 python bench.py --algo brox --steps 2 --no-others > $O/bench_brox_1080p.json 2>/dev/null; echo "brox rc=$?"
timeout 300 python bench.py --algo brox --width 3840 --height 2160 --frames 300 --step 2 --steps 1 --no-others --no-cpu-baseline > $O/bench_brox_4k_s2_300frames.json 2>/dev/null; echo "brox4k rc=$?"
timeout 300 python bench.py --algo tvl1 --width 224 --height 224 --steps 5 --no-others --no-cpu-baseline > $O/bench_tvl1_224x224.json 2>/dev/null; echo "224 rc=$?"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import sys,json
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['config'].get('pcie_inclusive',{})
print(sys.argv[1].split('/')[-1], round(d['value'],1), 'pairs/s frac', round(d['roofline']['frac'],3), 'f32', round(p.get('value',0),1), 'jpeg', round(p.get('jpeg_files_out',{}).get('value',0),1))
PY
done
for A in tvl1 farn brox; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$A -- python $R/bench.py --algo $A --steps 2 --no-cpu-baseline --no-others --no-pcie > /dev/null 2>&1 )
  f=$(find $O/stats_$A -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_${A}_1080p_kernel_stats.csv && python scripts/kstats.py $f | head -6; rm -rf $O/stats_$A
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_jpeg -- python $R/scripts/jpeg_rate.py > $R/$O/jpeg_rate.txt 2>&1 )
f=$(find $O/stats_jpeg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/jpeg_out_farn_1080p_kernel_stats.csv && python scripts/kstats.py $f | grep -i "jpeg\|farn_iteration" | head -8; rm -rf $O/stats_jpeg
CONFIGS="dev-bound/host-jpeg,device" timeout 900 python scripts/e2e_cli_rate.py 1920 1080 513 2>&1 | grep -v amdgpu.ids | tee $O/e2e_1080p_513frames.log
( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/$O/sq_brox -- python $R/bench.py --algo brox --steps 1 --warmup 0 --frames 34 --max-batch 16 --no-cpu-baseline --no-others --no-pcie > $R/$O/sq_brox.log 2>&1 ); echo "sq rc=$?"
python scripts/sq_summary.py $O/sq_brox k_brox_sor k_brox_stage1 > $O/sq_brox_sor_pk.json 2>/dev/null; head -40 $O/sq_brox_sor_pk.json; rm -rf $O/sq_brox
