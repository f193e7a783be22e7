#!/bin/bash
# Round 6, GPU call zq2: the compaction grid chosen per launch (1 024 workgroups from 1.5 M elements on; tools/ab/tree_old = the tree before)
TAG=${1:-r06zq2}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
REPO=$(pwd)
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -2
run() { # name dir args
  n=$1; d=$2; shift; shift
  (cd $d && timeout 400 python bench.py "$@" > $REPO/gpurun_out/${TAG}_$n.json 2> $REPO/gpurun_out/${TAG}_$n.err)
  python - gpurun_out/${TAG}_$n.json "$n" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d.get("stage_ms") or d.get("roofline", {}).get("stage_ms", {})
print(f"{sys.argv[2]:14s}", round(d["value"], 1), "frames/s", round(d["ms_per_step"] * 1e3, 1), "us   pre", round(st.get("Preprocess", 0) * 1e3, 1))
PY
}
V="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 1.0"
for i in 1 2; do
  run vga_new . $V; run vga_old tools/ab/tree_old $V
  run 2s_new . --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline; run 2s_old tools/ab/tree_old --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline
  run c4_new . --config 4 --frame-cache $CACHE --min-seconds 0.5; run c4_old tools/ab/tree_old --config 4 --frame-cache $CACHE --min-seconds 0.5
  run 4n_new . --config 4n $V; run 4n_old tools/ab/tree_old --config 4n $V
done
