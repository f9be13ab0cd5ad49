#!/bin/bash
O=gpurun_out/r4_jpegtest; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_jpeg_gpu.py tests/test_async_gpu.py tests/test_segments_gpu.py -x -q > $O/pytest_jpeg.log 2>&1; tail -30 $O/pytest_jpeg.log
