#!/bin/bash
# Round 6, GPU call zb: "batchSolveInPixelPass" -- one launch per iteration of the batched Gauss-Newton loop.  Tests, then A/B on S2 and configs[4]
TAG=${1:-r06zb}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multimodel.py tests/test_gpu_switches.py tests/test_gpu_sharded.py -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity_long.py -q -m gpu 2>&1 | tail -3
for v in 1 0 1 0; do
  timeout 400 python bench.py --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline --param batchSolveInPixelPass=$v > gpurun_out/${TAG}_2s_$v.json 2> gpurun_out/${TAG}_2s_$v.err
  python - gpurun_out/${TAG}_2s_$v.json "2s batchSolveInPixelPass=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 4), "ms")
PY
done
for v in 1 0 1 0; do
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 --param batchSolveInPixelPass=$v > gpurun_out/${TAG}_c4_$v.json 2> gpurun_out/${TAG}_c4_$v.err
  python - gpurun_out/${TAG}_c4_$v.json "c4 tracked batchSolveInPixelPass=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms", "odom", round(d["stage_ms"]["odom"], 3))
PY
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/bench.py --config 2s --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --steps 200 --warmup 60 > /tmp/prof_k.log 2>&1
cp $(find /tmp/prof_k -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_2s_kernel_stats.csv
head -8 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_2s_kernel_stats.csv | cut -c1-110
