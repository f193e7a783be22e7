#!/bin/bash
# Round 5, GPU call f: size-based choice of the clean form (two launches below 6 M elements, one launch above) -- parity subset + the two bench lines.
TAG=${1:-r05f}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_surfel_passes.py tests/test_gpu_glsl_passes.py tests/test_gpu_multimodel.py tests/test_gpu_pipeline.py tests/test_gpu_api.py \
   tests/test_gpu_labels.py tests/test_gpu_sharded.py tests/test_gpu_switches.py \
   "tests/test_gpu_parity_long.py::test_config4_dense_maps" "tests/test_gpu_parity_long.py::test_s2_eight_objects_tracked_teacher_forced" \
   "tests/test_gpu_parity_long.py::test_s2_eight_objects_standing_own_filters" "tests/test_gpu_parity_long.py::test_s2_eight_objects_tracked" \
   -x -q -m gpu -n 8 --durations=6 > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --config 4 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench c4 rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench default rc=$?"
cut -c1-500 gpurun_out/${TAG}_bench_c4.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05f_bench.json'))
print(d['value'], {k:round(v*1e3,1) for k,v in d['roofline']['stage_ms'].items() if v})
for k,x in (d.get('variants') or {}).items(): print('  variant',k, round(x['value'],1), x.get('ms_per_step'))
print('host_input', d['host_input']['value'])
PY
tail -3 gpurun_out/${TAG}_bench.err
tail -14 gpurun_out/${TAG}_pytest.log | cut -c1-220
