#!/bin/bash
# Round 6, GPU call a: the tree at the start of the round + the tracked S3 parity test: whole GPU suite, the default bench line
TAG=${1:-r06a}
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -q -m gpu -x ) > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity_long.py::test_config4_dense_maps_tracked -q -m gpu -n 0 -x -s > gpurun_out/${TAG}_c4t.log 2>&1
echo "c4 tracked rc=$?"; grep -v "^ \{4,\}\|^E \|^>" gpurun_out/${TAG}_c4t.log | tail -30 | cut -c1-400
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06a_bench.json"))
    print("value", d["value"], "roofline", d["roofline"]["frac"], {k: v.get("value") for k, v in (d.get("variants") or {}).items()})
except Exception as e:
    print("bench parse failed", e)
PY
