#!/bin/bash
# round 4, second call: per-level pixel slots + single-wavefront Gauss-Newton finish, model recycling / retired-log arena, asynchronous
# host-pointer frames -- correctness subset first, then the figures they move.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_multimodel.py tests/test_gpu_api.py tests/test_gpu_facade.py \
    tests/test_gpu_gn_graph.py tests/test_gpu_sharded.py tests/test_gpu_rgbd.py tests/test_gpu_parity_long.py -m gpu -q --durations=12 -s \
    -k "not long_horizon_ate and not standing and not config4 and not eight_objects_tracked or teacher_forced" > gpurun_out/r04b_pytest.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04b_pytest.log | tail -12
grep -n "differ in colour\|slot \|oracle \[" gpurun_out/r04b_pytest.log | head -20 | cut -c1-400
grep -n "teacher-forced:" gpurun_out/r04b_pytest.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; tail -2 gpurun_out/r04b_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b_bench.json'))
print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline']['us_per_launch'],{k:(round(v['us'],3),round(v['frac'],4)) for k,v in (d['roofline']['levels'] or {}).items()})
print('host_input',d['host_input']['value'],'variant',d['variants']['reference_default']['value'] if d.get('variants') else None)
PY
bash tools/kstats.sh r04b_c1 2>&1 | head -24
bash tools/kstats.sh r04b_2s --config 2s 2>&1 | head -50
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r04b_bench_2s.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b_bench_2s.json'))
print('2s value',d['value'],'ms',d['ms_per_step'],'models',d['config']['models'],'host_input',d['host_input'],'stages',{k:round(v,4) for k,v in d['roofline']['stage_ms'].items()})
PY
timeout 120 python tools/icp_prof.py > gpurun_out/r04b_icp_prof.txt 2>&1; tail -45 gpurun_out/r04b_icp_prof.txt
