#!/bin/bash
# Round 6, GPU call zg: k_lab_models clears its tables with the whole wavefront.  Label / multi-model tests, S2 before / after (tools/ab/tree_old = the tree before)
TAG=${1:-r06zg}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_labels.py tests/test_gpu_multimodel.py tests/test_gpu_sharded.py -q -m gpu 2>&1 | tail -3
for t in new old new old; do
  D=$REPO; [ $t = old ] && D=$REPO/tools/ab/tree_old
  (cd $D && timeout 400 python bench.py --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline > $REPO/gpurun_out/${TAG}_2s_$t.json 2> $REPO/gpurun_out/${TAG}_2s_$t.err)
  python - gpurun_out/${TAG}_2s_$t.json "2s $t" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 4), "ms")
PY
done
