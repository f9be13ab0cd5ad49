#!/bin/bash
# Round 4: kernel-time breakdown of the final Farneback path (torch-free harness, batch 129, one timed pass)
O=gpurun_out/r4_farn13; mkdir -p $O; export TMPDIR=/tmp; R=/root/repo
cd /root/repo
python scripts/make_raw_clip.py 1920 1080 2 130 /tmp/clip1080.raw 2> $O/mk.err || { tail -3 $O/mk.err; exit 1; }
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o p -- $R/build/dfx_prof farn 1920 1080 /tmp/clip1080.raw 130 1 1 ) > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/farn_kernel_stats_harness.csv \; ; rm -rf $O/stats
cat $O/farn_kernel_stats_harness.csv | cut -d, -f1-5 | cut -c1-120
