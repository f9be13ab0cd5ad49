#!/bin/bash
# Run the CPU oracle (test infrastructure) under AddressSanitizer + UBSan on a range of frame sizes: the checker
# itself must not owe its answers to out-of-bounds reads or undefined behaviour.  CPU only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
( cd $R/oracle && gcc -O1 -g -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -D_GNU_SOURCE \
    -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $T/liboracle_asan.so \
    tvl1_oracle.c farneback_oracle.c brox_oracle.c quant_oracle.c prepare_oracle.c -lm )
cat > $T/run.py <<PY
import sys
sys.path.insert(0, "$R")
from oracle import oracle_py as O
O._LIB_PATH = "$T/liboracle_asan.so"
import numpy as np
from denseflow_amd.synth import SynthClip
for (w, h) in [(64, 48), (97, 61), (16, 16), (33, 40), (8, 64), (130, 70)]:
    c = SynthClip(max(w, 32), max(h, 32), 3)
    f0 = np.ascontiguousarray(c.frame(0)[:h, :w]); f1 = np.ascontiguousarray(c.frame(1)[:h, :w])
    a = O.tvl1_calc(f0, f1, threads=2); O.farneback_calc(f0, f1, threads=2); O.brox_calc(f0, f1, threads=2)
    O.flow_to_u8(a, -20, 20)
for (sw, sh, dw, dh, ch) in [(64, 48, 32, 24, 1), (7, 5, 3, 2, 3), (33, 17, 64, 64, 1), (1, 1, 5, 4, 3), (100, 70, 224, 224, 1)]:
    src = np.random.default_rng(0).integers(0, 256, (sh, sw) if ch == 1 else (sh, sw, 3), dtype=np.uint8)
    O.prepare_frame(src, dw, dh)
print("oracle: no sanitizer report")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python $T/run.py
rm -rf $T
