#!/bin/bash
# Round 6, GPU call zl: launch shape of the tile pass again, now with 16 x 24 tiles: lanes per sprite, threads per workgroup (configs[1])
TAG=${1:-r06zl}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
B="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 1.0"
for sh in "4 512" "2 512" "8 512" "4 384" "4 1024" "4 512" "2 512" "8 512" "4 384"; do
  set -- $sh
  timeout 300 python bench.py $B --param spriteLanes=$1 --param tileThreads=$2 > gpurun_out/${TAG}_$1_$2.json 2> gpurun_out/${TAG}_$1_$2.err
  python - gpurun_out/${TAG}_$1_$2.json "lanes $1 threads $2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms"]
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"] * 1e3, 1), "us  predict", round(st["IndexMap::ACTIVE"] * 1e3, 1))
PY
done
