#!/bin/bash
# Round 6, GPU call zh: "so3Stream" -- the SO(3) pre-alignment beside the depth filter and the pyramids.  Tests, A/B at the reference's GUI defaults
TAG=${1:-r06zh}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_rgbd.py tests/test_gpu_multimodel.py tests/test_gpu_api.py -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity_long.py -q -m gpu -k "reference_defaults or rgbd or so3" 2>&1 | tail -3
B="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --no-roofline --min-seconds 1.0 --icp-weight 20 --so3"
for v in 1 0 1 0 1 0; do
  timeout 300 python bench.py $B --param so3Stream=$v > gpurun_out/${TAG}_rd_$v.json 2> gpurun_out/${TAG}_rd_$v.err
  python - gpurun_out/${TAG}_rd_$v.json "reference defaults so3Stream=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"] * 1e3, 1), "us")
PY
done
