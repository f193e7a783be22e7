#!/bin/bash
# round 3, call e: loop-level shard calls (mf_track_models / mf_fuse_models / mf_predict_models): parity + the one-GPU scene rate
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "sharded or multimodel or launch_switches or api" > gpurun_out/r03e_pytest.log 2>&1; tail -3 gpurun_out/r03e_pytest.log
timeout 400 python bench.py --config 3 --steps 100 --warmup 20 > gpurun_out/r03e_bench_config3.json 2> gpurun_out/r03e_bench_config3.err; cut -c1-420 gpurun_out/r03e_bench_config3.json; tail -2 gpurun_out/r03e_bench_config3.err
timeout 400 python bench.py --force-sharded-scene --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03e_bench_scene.json 2> gpurun_out/r03e_bench_scene.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03e_bench_scene.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('sharded_scene'), indent=0)[:1500])
PY
