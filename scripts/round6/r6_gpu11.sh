#!/bin/bash
# round 6, GPU call 11: (a) TVL1 step kernel on 64 x 48 tiles (6 waves, two workgroups per CU) against 64 x 32 (4 waves,
# three per CU); (b) SQ counters of the two synchronisation forms of the Brox fused SOR
set -u
R=$GRAFT_REPO_ROOT; O=gpurun_out/r6_11; mkdir -p $R/$O; export TMPDIR=/tmp
cd $R; make -s host > $O/make_host.log 2>&1
DFX_LIBRARY=$R/build/variants/libdfx_t48.so timeout 900 python -m pytest tests/test_tvl1_gpu.py -q -m gpu -x -k "single_pair or golden or tile_geometry or head or batched" 2>&1 | tail -4
b() { env $2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-live-pmc --no-others $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"],1), "pairs/s  frac", round(d["roofline"]["frac"],3), "parity", d.get("parity_check",{}).get("max_abs"))
PY
}
for rep in 1 2; do b t32_$rep X=1 ""; b t48_$rep DFX_LIBRARY=$R/build/variants/libdfx_t48.so ""; done
b t32_hard X=1 "--clip hard --frames 66 --no-parity"; b t48_hard DFX_LIBRARY=$R/build/variants/libdfx_t48.so "--clip hard --frames 66 --no-parity"
b t32_224 X=1 "--width 224 --height 224 --clips 64 --no-parity"; b t48_224 DFX_LIBRARY=$R/build/variants/libdfx_t48.so "--width 224 --height 224 --clips 64 --no-parity"
python scripts/make_raw_clip.py 1920 1080 2 66 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
run() { n=$1; var=$2; shift 2
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$n -o p -- $R/build/dfx_prof brox 1920 1080 /tmp/clip1080.raw 66 1 1 0 $var 0 ) > $O/$n.log 2>&1
  python scripts/sq_summary.py $O/$n k_brox_sor > $O/$n.json 2>&1; rm -rf $O/$n; }
for v in 0 128; do
  run sqA_brox_v$v $v SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
  run sqB_brox_v$v $v SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
done
python - <<'PY'
import json
O="gpurun_out/r6_11"
for n in ("sqA_brox_v0","sqA_brox_v128","sqB_brox_v0","sqB_brox_v128"):
    d=json.load(open(f"{O}/{n}.json"))
    for k,v in d.items(): print(n, k[:40], {c: (round(x,4) if isinstance(x,float) and x<10 else round(x)) for c,x in v.items() if c!="note"})
PY
