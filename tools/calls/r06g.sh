#!/bin/bash
# Round 6, GPU call g: slab culling with the cheap rectangle (level 0, object models only): S2 on one GPU with and without, configs[4] tracked, the two tests
TAG=${1:-r06g}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_multimodel.py tests/test_gpu_parity_long.py::test_s2_eight_objects_tracked_teacher_forced -q -m gpu -x > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log | cut -c1-300
for sc in 1 0; do
  timeout 300 python bench.py --config 2s --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --param slabCulling=$sc > gpurun_out/${TAG}_2s_sc$sc.json 2> gpurun_out/${TAG}_2s_sc$sc.err
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 1.0 --param slabCulling=$sc > gpurun_out/${TAG}_c4_sc$sc.json 2> gpurun_out/${TAG}_c4_sc$sc.err
done
python - <<'PY'
import json
for sc in (1, 0):
    for n in ("2s", "c4"):
        try:
            d = json.load(open(f"gpurun_out/r06g_{n}_sc{sc}.json")); print(n, "slabCulling", sc, round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 3) for k, v in (d.get("stage_ms") or {}).items() if v})
        except Exception as e:
            print(n, sc, "FAILED", e)
PY
