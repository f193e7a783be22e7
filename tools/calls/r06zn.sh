#!/bin/bash
# Round 6, GPU call zn: the background's first index pass beside the label stage ("indexPassOverlapElements").  Tests, A/B on configs[4] and S2
TAG=${1:-r06zn}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multimodel.py tests/test_gpu_switches.py tests/test_gpu_sharded.py -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity_long.py -q -m gpu 2>&1 | tail -2
for v in 1048576 2000000000 1048576 2000000000; do
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 --param indexPassOverlapElements=$v > gpurun_out/${TAG}_c4_$v.json 2> gpurun_out/${TAG}_c4_$v.err
  python - gpurun_out/${TAG}_c4_$v.json "c4 tracked indexPassOverlapElements=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms")
PY
done
for v in 0 1048576 0 1048576; do
  timeout 400 python bench.py --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline --param indexPassOverlapElements=$v > gpurun_out/${TAG}_2s_$v.json 2> gpurun_out/${TAG}_2s_$v.err
  python - gpurun_out/${TAG}_2s_$v.json "2s indexPassOverlapElements=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 4), "ms")
PY
done
