#!/bin/bash
# Round 6, closing run on the final tree: GPU suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats of
# bench.py per algorithm (the summaries roofline.avg_launch_us must agree with), the warp-and-head kernel against the
# two-launch form (bench lines, per-dispatch timelines, SQ counters, HBM traffic), the Brox SOR's two synchronisation
# forms and its streaming form (bench lines), every flow of the headline clip against the oracle.
set -u
R=$GRAFT_REPO_ROOT; O=gpurun_out/r6_final; mkdir -p $R/$O; export TMPDIR=/tmp
cd $R; make -s host > $O/make_host.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "bytes=$(wc -c < $O/bench_default.json)"
for a in tvl1 farn brox; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$a -o p -- python $R/bench.py --algo $a --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-pcie --no-live-pmc --no-parity ) > $O/bench_${a}_1080p_profiled.json 2> $O/stats_$a.err
  find $O/stats_$a -name "*kernel_stats.csv" -exec cp {} $O/bench_${a}_1080p_kernel_stats.csv \;
  rm -rf $O/stats_$a
  python scripts/kstats.py $O/bench_${a}_1080p_kernel_stats.csv | head -4
done
python bench.py --algo farn --no-others --no-cpu-baseline > $O/bench_farn_1080p.json 2>> $O/err.log
python bench.py --algo brox --no-others --no-cpu-baseline > $O/bench_brox_1080p.json 2>> $O/err.log
# the warp-and-head kernel against the two launches it replaces, alternating
for rep in 1 2; do for v in 0 64; do
  python bench.py --variant $v --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/head_ab_v${v}_$rep.json 2>> $O/err.log
done; done
cd /tmp
for v in 0 64; do
  ( SWEEP="0:4:0:$v" timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$v -o t -- python $R/scripts/sweep_tvl1.py 1920 1080 130 ) > $R/$O/trace_$v.log 2>&1
  F=$(find $R/$O/trace_$v -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python $R/scripts/tvl1_timeline.py "$F" $R/$O/timeline_v${v}_dispatches.csv > $R/$O/timeline_v$v.md 2>$R/$O/timeline_v$v.err; rm -rf $R/$O/trace_$v
done
cd $R
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err
run() { n=$1; algo=$2; nf=$3; var=$4; shift 4
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof $algo 1920 1080 /tmp/clip1080.raw $nf 1 1 0 $var 0 ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n > $O/$n.json 2>&1; rm -rf $O/$n; }
for v in 0 64; do
  run sqA_tvl1_v$v tvl1 130 $v SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
  run sqB_tvl1_v$v tvl1 130 $v SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  run fetch_tvl1_v$v tvl1 130 $v FETCH_SIZE
  run write_tvl1_v$v tvl1 130 $v WRITE_SIZE
done
for a in farn brox; do nf=130; [ $a = brox ] && nf=66
  run fetch_$a $a $nf 0 FETCH_SIZE; run write_$a $a $nf 0 WRITE_SIZE
done
# Brox fused SOR: streaming (default: persistent workgroups + LDS-DMA prefetch) vs one workgroup per tile (256) vs the latter
# with band-wise progress counters instead of barriers (128)
for rep in 1 2; do for v in 0 256 128; do
  python bench.py --algo brox --frames 131 --variant $v --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/brox_sync_v${v}_$rep.json 2>> $O/err.log
done; done
for v in 0 256; do
  python bench.py --algo brox --width 3840 --height 2160 --frames 66 --step 2 --variant $v --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-live-pmc --no-others > $O/brox_4k_v${v}.json 2>> $O/err.log
done
timeout 900 python scripts/round6/full_clip_parity.py 300 > $O/full_clip_parity.txt 2>> $O/err.log; cat $O/full_clip_parity.txt
python - <<'PY'
import json, glob
O="gpurun_out/r6_final"
d=json.loads(open(O+"/bench_default.json").read().strip().splitlines()[-1])
print("tvl1", d["value"], {k:v for k,v in d["roofline"].items() if k in ("bound","frac","traffic_frac","valu_frac","useful_frac","avg_launch_us","shader_GHz")}, d.get("parity_check"))
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list,str))})
print(d.get("cpu_baseline"))
for a in ("farn","brox"):
    x=json.loads(open(f"{O}/bench_{a}_1080p.json").read().strip().splitlines()[-1]); print(a, x["value"], {k:v for k,v in x["roofline"].items() if k in ("bound","frac","traffic_frac","avg_launch_us")}, x.get("parity_check"))
for f in sorted(glob.glob(O+"/head_ab_*.json"))+sorted(glob.glob(O+"/brox_sync_*.json"))+sorted(glob.glob(O+"/brox_4k_v*.json")):
    x=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(x["value"],2), x.get("parity_check",{}).get("max_abs"))
for v in (0,64):
    f=json.load(open(f"{O}/fetch_tvl1_v{v}.json")); w=json.load(open(f"{O}/write_tvl1_v{v}.json"))
    for k in f:
        if "k_tvl1_step" in k or "warp" in k: print(v, k[:44], f[k]["dispatches"], "MB/pair/launch", round((2*f[k].get("FETCH_SIZE",0)+w[k].get("WRITE_SIZE",0))*1024/129/1e6,2))
PY
