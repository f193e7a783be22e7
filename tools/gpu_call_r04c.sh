#!/bin/bash
# round 4, third call: re-check after the L0 / L1 grid changes, readlane finish, pipelined tile pass and contraction-free pose_derive
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_multimodel.py tests/test_gpu_surfel_passes.py tests/test_gpu_gn_graph.py \
    tests/test_gpu_parity_long.py tests/test_gpu_f4_bound.py -m gpu -q --durations=8 -s \
    -k "not long_horizon_ate and not standing and not config4 and not eight_objects_tracked or teacher_forced" > gpurun_out/r04c_pytest.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04c_pytest.log | tail -12
grep -n "differ in colour\|   slot \|             oracle" gpurun_out/r04c_pytest.log | head -12 | cut -c1-300
grep -n "teacher-forced:" gpurun_out/r04c_pytest.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 --no-variants > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; tail -2 gpurun_out/r04c_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04c_bench.json'))
print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline']['us_per_launch'],{k:(round(v['us'],3),round(v['frac'],4)) for k,v in (d['roofline']['levels'] or {}).items()})
print('host_input',d['host_input']['value'], 'stages', {k:round(v,4) for k,v in d['roofline']['stage_ms'].items() if v})
PY
bash tools/kstats.sh r04c_c1 --no-variants 2>&1 | head -24
timeout 120 python tools/icp_prof.py > gpurun_out/r04c_icp_prof.txt 2>&1; tail -45 gpurun_out/r04c_icp_prof.txt
timeout 120 python tools/splat_prof.py > gpurun_out/r04c_splat_prof.txt 2>&1; tail -15 gpurun_out/r04c_splat_prof.txt
