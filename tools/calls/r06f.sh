#!/bin/bash
# Round 6, GPU call f: slab culling of the batched Gauss-Newton pixel pass; whole GPU suite + the default bench line with its variants
TAG=${1:-r06f}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -q -m gpu -x --durations=6 ) > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -14 gpurun_out/${TAG}_pytest.log | cut -c1-300
timeout 600 python bench.py --frame-cache $CACHE > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err | cut -c1-300
for sc in 1 0; do
  timeout 300 python bench.py --config 2s --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --param slabCulling=$sc > gpurun_out/${TAG}_2s_sc$sc.json 2> gpurun_out/${TAG}_2s_sc$sc.err
done
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06f_bench.json"))
    print("value", d["value"], "roofline", d["roofline"]["frac"], {k: round(v.get("value"), 1) for k, v in (d.get("variants") or {}).items()})
    c4 = d["variants"]["config4_stress"]
    print("c4 passes", {k: round(v["ms"], 3) for k, v in c4["roofline_passes"]["passes"].items()}, "tracked", c4["tracked_models"], "compactions", c4["compactions"])
    print("c4 stages", {k: round(v, 3) for k, v in c4["stage_ms"].items() if v})
except Exception as e:
    print("bench parse failed", e)
for sc in (1, 0):
    try:
        d = json.load(open(f"gpurun_out/r06f_2s_sc{sc}.json")); print("2s slabCulling", sc, d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in (d.get("stage_ms") or {}).items() if v})
    except Exception as e:
        print("2s", sc, "FAILED", e)
PY
