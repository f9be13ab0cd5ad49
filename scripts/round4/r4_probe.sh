#!/bin/bash
# Round 4, GPU call 1: is there any OpenCV (cv2 / libopencv / a wheel) on the MI355X box, or any way to get one?
# VERDICT r3 "Next round" item 1.  Transcript -> gpurun_out/r4_probe.txt -> profiles/round4/cv2_probe_gpu_box.txt
out=gpurun_out/r4_probe.txt
mkdir -p gpurun_out
{
echo "## host"; hostname; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -8
echo "## python -c 'import cv2'"; python -c "import cv2; print(cv2.__version__, cv2.__file__)" 2>&1 | tail -2
echo "## /opt/conda python"; ls /opt/conda/bin/python* 2>&1 | head -3; /opt/conda/bin/python -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1
echo "## pip config"; pip config list 2>&1 | grep -v WARNING
echo "## pip download opencv-contrib-python-headless"; timeout 60 pip download --no-deps -d /tmp/whl opencv-contrib-python-headless 2>&1 | grep -v WARNING | tail -4
echo "## pip download with index re-enabled"; PIP_NO_INDEX=0 timeout 60 pip download --no-deps -d /tmp/whl --index-url https://pypi.org/simple opencv-contrib-python-headless 2>&1 | grep -v WARNING | tail -4
echo "## pip index versions"; PIP_NO_INDEX=0 timeout 60 pip index versions opencv-python 2>&1 | grep -v WARNING | tail -3
echo "## conda"; timeout 60 /opt/conda/bin/conda search opencv 2>&1 | tail -3
echo "## network"; timeout 10 curl -sS -o /dev/null -w "%{http_code}\n" https://pypi.org/simple/ 2>&1 | tail -1; timeout 5 getent hosts pypi.org 2>&1 | tail -1
echo "## wheelhouses"; ls /opt/wheelhouse 2>/dev/null | grep -i -E "opencv|cv2" ; find / -xdev \( -name "*.whl" -o -name "*.tar.bz2" -o -name "*.conda" \) 2>/dev/null | grep -i -E "opencv|cv2" | head
echo "## any opencv on disk"; find / -xdev \( -iname "*opencv*" -o -name "cv2*" -o -name "libopencv*" \) -not -path "/proc/*" 2>/dev/null | grep -v -E "^/root/repo|gpurun|graft" | head -20
echo "## apt"; apt list --installed 2>/dev/null | grep -i -E "opencv|ffmpeg|libav" | head
echo "## done"
} > $out 2>&1
cat $out
