#!/bin/bash
# Round 6, GPU call q: the printed figures of the tracked S3 parity test; what a compaction costs (configs[4], one forced every 10 frames)
TAG=${1:-r06q}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_long.py::test_config4_dense_maps_tracked -q -m gpu -n 0 -x -s > gpurun_out/${TAG}_c4t.log 2>&1
echo "c4 tracked rc=$?"; grep -v "^ \{4,\}\|^E \|^>" gpurun_out/${TAG}_c4t.log | tail -16 | cut -c1-500
timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 --static-objects --param densifyEvery=10 > gpurun_out/${TAG}_c4_densify.json 2> gpurun_out/${TAG}_c4_densify.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06q_c4_densify.json"))
print("c4 static, a compaction forced every 10 frames:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms; compactions", d["compactions"])
print("   compaction pass (mean over 10 instrumented frames, i.e. one compaction):", d["roofline_passes"]["passes"].get("compaction"), {k: round(v["ms"], 3) for k, v in d["roofline_passes"]["passes"].items()})
PY
