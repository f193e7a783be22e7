#!/bin/bash
# Round 6, GPU call u: "objectStream" -- the batched object passes beside the background's chain.  Bit-identity tests, then A/B on configs[4] and S2
TAG=${1:-r06u}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multimodel.py tests/test_gpu_switches.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python -m pytest "tests/test_gpu_parity_long.py::test_config4_dense_maps_tracked" "tests/test_gpu_parity_long.py::test_config4_dense_maps" -q -m gpu -x -n 2 2>&1 | tail -3
for v in 1 0 1 0; do
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 --param objectStream=$v > gpurun_out/${TAG}_c4_os$v.json 2> gpurun_out/${TAG}_c4_os$v.err
  python - gpurun_out/${TAG}_c4_os$v.json "c4 tracked objectStream=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms", "frame roofline", round(d.get("roofline_frame", {}).get("frac", 0), 3))
PY
done
for v in 1 0; do
  timeout 400 python bench.py --config 4 --static-objects --frame-cache $CACHE --min-seconds 0.5 --param objectStream=$v > gpurun_out/${TAG}_c4s_os$v.json 2> gpurun_out/${TAG}_c4s_os$v.err
  python - gpurun_out/${TAG}_c4s_os$v.json "c4 static objectStream=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms")
PY
done
for v in 1 0 1 0; do
  timeout 400 python bench.py --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline --param objectStream=$v > gpurun_out/${TAG}_2s_os$v.json 2> gpurun_out/${TAG}_2s_os$v.err
  python - gpurun_out/${TAG}_2s_os$v.json "2s objectStream=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms")
PY
done
