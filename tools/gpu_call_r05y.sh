#!/bin/bash
# Round 5, GPU call y: the frame's intensity pyramid + derivative / gate images as one launch: equality tests, the RGB-D + SO(3) ATE gate, A/B of the
# reference-default bench line
TAG=${1:-r05y}
mkdir -p gpurun_out
timeout 300 python -m pytest "tests/test_gpu_switches.py::test_fused_rgb_pyramid_equals_the_single_kernels" ${MF_R05Y_MORE_TESTS} \
   -q -m gpu -n 4 -x > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log | cut -c1-200
rd() {
  n=$1; shift
  timeout 200 python bench.py --icp-weight 20 --so3 --frame-cache /tmp/mf_frames --no-variants --no-host-input --no-cpu-baseline "$@" > gpurun_out/${TAG}_rd_$n.json 2> gpurun_out/${TAG}_rd_$n.err
  python - "$n" gpurun_out/${TAG}_rd_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(f"reference-default {sys.argv[1]:10s} {d['value']:7.1f} frames/s  {d['ms_per_step']*1e3:.1f} us")
except Exception as e:
    print("rd", sys.argv[1], "FAILED", e)
PY
}
rd fused
rd single --param fusedRgbPyramid=0
