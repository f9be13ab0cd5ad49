#!/usr/bin/env python3
"""PMC calibration workload: streaming kernels with a KNOWN byte count (denseflow_amd/csrc/selftest.hip dfxi_calib).
Run under rocprofv3 --pmc FETCH_SIZE (then --pmc WRITE_SIZE): the counter value per dispatch divided by the true
bytes gives the correction factor for that access width on this gfx950 / rocprofv3 (MI355X_MICROARCH.md §HBM says
FETCH_SIZE reads 1/2 for 16 B/lane streams and leaves every other width uncalibrated).
Usage: python scripts/pmc_calibrate.py [GiB]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import denseflow_amd  # noqa: E402

L = denseflow_amd.load_library()
L.dfxi_calib.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int]
L.dfxi_calib.restype = C.c_int
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
nbytes = int(gib * (1 << 30)) // 4096 * 4096
for kind, name in ((0, "copy 4 B/lane"), (1, "copy 16 B/lane"), (2, "read 4 B/lane")):
    rc = L.dfxi_calib(0, kind, nbytes, 3)
    print(f"{name}: {nbytes} bytes read{'' if kind == 2 else ' + written'} per launch, 3 launches, rc={rc}", flush=True)
