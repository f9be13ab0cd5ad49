#!/bin/bash
mkdir -p gpurun_out /tmp/dbg
export TMPDIR=/tmp
make host > /dev/null 2>&1
python - <<'PY'
import sys, numpy as np
sys.path.insert(0,'.')
from denseflow_amd.synth import SynthClip
fr=SynthClip(224,224,1).frames(4)
with open('/tmp/dbg/clip.y4m','wb') as f:
    f.write(b"YUV4MPEG2 W224 H224 F30:1 Ip A1:1 Cmono\n")
    for x in fr: f.write(b"FRAME\n"); f.write(x.tobytes())
open('/tmp/dbg/list.txt','w').write('/tmp/dbg/clip.y4m\n')
PY
echo "== A: pinned, single file"; DF_TRACE=1 timeout -s KILL 40 ./build/denseflow /tmp/dbg/clip.y4m -o=/tmp/dbg/outA -a=tvl1 -s=1 -b=20 2>&1 | tail -20; echo "rc=$?"
echo "== B: no pinned, single file"; DF_NO_PINNED=1 DF_TRACE=1 timeout -s KILL 40 ./build/denseflow /tmp/dbg/clip.y4m -o=/tmp/dbg/outB -a=tvl1 -s=1 -b=20 2>&1 | tail -20; echo "rc=$?"
echo "== C: pinned, list.txt"; DF_TRACE=1 timeout -s KILL 40 ./build/denseflow /tmp/dbg/list.txt -o=/tmp/dbg/outC -a=tvl1 -s=1 -b=20 2>&1 | tail -20; echo "rc=$?"
ls /tmp/dbg/outA/clip /tmp/dbg/outB/clip 2>/dev/null | head
