#!/bin/bash
# Round 6, GPU call zk: 16 x 32 pixel tiles (as many pixels as a tile workgroup has threads) against the 16 x 24 default
TAG=${1:-r06zk}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
B="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 1.0"
for h in 24 32 24 32; do
  timeout 300 python bench.py $B --param tileHeight=$h > gpurun_out/${TAG}_v_$h.json 2> gpurun_out/${TAG}_v_$h.err
  python - gpurun_out/${TAG}_v_$h.json "configs[1] tileHeight=$h" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms"]
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"] * 1e3, 1), "us  predict", round(st["IndexMap::ACTIVE"] * 1e3, 1))
PY
done
for h in 24 32 24 32; do
  timeout 300 python bench.py --config 4n $B --param tileHeight=$h > gpurun_out/${TAG}_4n_$h.json 2> gpurun_out/${TAG}_4n_$h.err
  python - gpurun_out/${TAG}_4n_$h.json "1280x960 natural map tileHeight=$h" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"] * 1e3, 1), "us")
PY
done
for h in 24 32 24 32; do
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 --param tileHeight=$h > gpurun_out/${TAG}_c4_$h.json 2> gpurun_out/${TAG}_c4_$h.err
  python - gpurun_out/${TAG}_c4_$h.json "c4 tracked tileHeight=$h" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms")
PY
done
