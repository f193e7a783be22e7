#!/bin/bash
# Round 6, GPU call zf: per-model grids of the batched Gauss-Newton pixel pass (the background: many small chunks).  Whole suite, A/B on S2 / configs[4]
TAG=${1:-r06zf}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/${TAG}_pytest.log
for v in 0 1 0 1; do
  timeout 400 python bench.py --config 2s --frame-cache $CACHE --min-seconds 1.0 --no-cpu-baseline --param batchGridUniform=$v > gpurun_out/${TAG}_2s_$v.json 2> gpurun_out/${TAG}_2s_$v.err
  python - gpurun_out/${TAG}_2s_$v.json "2s batchGridUniform=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 4), "ms")
PY
done
for v in 0 1 0 1; do
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 --param batchGridUniform=$v > gpurun_out/${TAG}_c4_$v.json 2> gpurun_out/${TAG}_c4_$v.err
  python - gpurun_out/${TAG}_c4_$v.json "c4 tracked batchGridUniform=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms", "odom", round(d["stage_ms"]["odom"], 3))
PY
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/bench.py --config 2s --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --steps 200 --warmup 60 > /tmp/prof_k.log 2>&1
cp $(find /tmp/prof_k -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_2s_kernel_stats.csv
head -4 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_2s_kernel_stats.csv | cut -c1-110
