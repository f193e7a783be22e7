#!/bin/bash
# Round 6, GPU call b: Model::clean in place (k_cull_clean / k_clean_runs / append): whole GPU suite, configs[4] bench line + kernel trace of its dense frames
TAG=${1:-r06b}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -q -m gpu -x --durations=8 ) > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -16 gpurun_out/${TAG}_pytest.log | cut -c1-300
timeout 500 python bench.py --config 4 --frame-cache $CACHE > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench c4 rc=$?"; tail -3 gpurun_out/${TAG}_bench_c4.err | cut -c1-300
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench_c4.json"))
    print("c4 value", d["value"], "ms", d["ms_per_step"], "stages", {k: round(v, 3) for k, v in d["stage_ms"].items()})
    print("surfels", d["config"]["surfels"], "->", d["config"]["surfels_at_end"])
except Exception as e:
    print("bench parse failed", e)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4
C4="--config 4 --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --steps 20"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o s -- python $REPO/bench.py $C4 > /tmp/prof_c4.log 2>&1
python $REPO/tools/c4_dense_summary.py $(find /tmp/prof_c4 -name "*kernel_trace.csv" | head -1) > $REPO/gpurun_out/${TAG}_c4_kernel_stats.csv
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_c4_kernel_stats_raw.csv
cd $REPO
head -40 gpurun_out/${TAG}_c4_kernel_stats.csv | cut -c1-150
