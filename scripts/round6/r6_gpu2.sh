#!/bin/bash
# round 6, GPU call 2: the fixed tile copy (parity), per-dispatch timelines of one 129-pair batch with / without the head
# kernel, SQ counters of the head kernel
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_2; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_tvl1_gpu.py tests/test_content_classes_gpu.py tests/test_bench_shaped_batch_gpu.py -q -m gpu -x -k "tvl1 or content" 2>&1 | tail -8 > $O/pytest.log
cat $O/pytest.log
cd /tmp
for v in 0 64; do
  ( SWEEP="0:4:0:$v" timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o t -- python $R/scripts/sweep_tvl1.py 1920 1080 130 ) > $O/trace_$v.log 2>&1; echo "trace $v rc=$?"
  F=$(find $O/trace_$v -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python $R/scripts/tvl1_timeline.py "$F" $O/timeline_v${v}_dispatches.csv > $O/timeline_v$v.md 2>$O/timeline_v$v.err; rm -rf $O/trace_$v; cat $O/timeline_v$v.md
done
for P in "A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "B SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  T=${P%% *}; C=${P#* }
  ( SWEEP="0:4:0:0" timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq_$T -o p -- python $R/scripts/sweep_tvl1.py 1920 1080 130 ) > $O/sq_$T.log 2>&1; echo "sq $T rc=$?"
  python $R/scripts/sq_summary.py $O/sq_$T k_tvl1 > $O/sq_head_$T.json 2>>$O/sq_$T.log; rm -rf $O/sq_$T
  python - <<PY
import json
d=json.load(open("$O/sq_head_$T.json"))
for k,v in d.items(): print(k, {a:(round(b,4) if isinstance(b,float) and b<10 else b) for a,b in v.items() if "/" in a or a in ("SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_LDS_BANK_CONFLICT","SQ_ACTIVE_INST_LDS","SQ_WAVE_CYCLES","dispatches")})
PY
done
