#!/bin/bash
# Round 4: does dfx_params.blocking_sync take effect inside a torch-hosted process (bench.py)?  user+sys CPU vs wall, spin vs blocking
O=gpurun_out/r4_block_torch; mkdir -p $O; cd /root/repo
for mode in spin blocking; do
  flag=""; [ $mode = blocking ] && flag="--blocking-sync"
  for a in tvl1 farn; do
    ( TIMEFORMAT="$mode $a wall %R s user %U s sys %S s"; time python bench.py --algo $a --steps 6 --warmup 1 --no-others --no-pcie --no-cpu-baseline --no-live-pmc $flag > $O/bench_${a}_$mode.json 2>> $O/err.log ) 2>> $O/times.txt
    python -c "import json;d=json.load(open('$O/bench_${a}_$mode.json'));print('$mode $a', d['value'], d['ms_per_step']*d['steps']/1e3, 's timed')" >> $O/times.txt
  done
done
cat $O/times.txt
