#!/bin/bash
# round 4, first call: the new -m gpu tests (switches / stand-ins / teacher-forced 8-object parity / RGB-D + SO(3) horizon), the bench line
# with its `variants` entry, kernel traces of the reference-default (RGB-D + SO(3)) frame and of the 18-model frame, the ICP stamps.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity_long.py -m gpu -q --durations=30 -s \
    -k "switches or teacher_forced or reference_defaults" > gpurun_out/r04a_pytest.log 2>&1; tail -45 gpurun_out/r04a_pytest.log | cut -c1-260
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; cut -c1-300 gpurun_out/r04a_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench.json'))
print('value',d['value'],'roofline',d['roofline']['frac'],d['roofline']['us_per_launch'],{k:(v['us'],v['frac']) for k,v in d['roofline']['levels'].items()})
print('host_input',d['host_input']['value'],'variants',json.dumps(d.get('variants'))[:1500])
PY
bash tools/kstats.sh r04a_rgbd --icp-weight 20 --so3
bash tools/kstats.sh r04a_2s --config 2s
cp $(find /tmp/prof_k -name '*kernel_trace.csv' | head -1) /tmp/r04a_2s_trace.csv 2>/dev/null && python tools/trace_gaps.py /tmp/r04a_2s_trace.csv > gpurun_out/r04a_2s_trace_gaps.txt 2>&1; head -60 gpurun_out/r04a_2s_trace_gaps.txt
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r04a_bench_2s.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench_2s.json'))
print('2s value',d['value'],'ms',d['ms_per_step'],'models',d['config']['models'],'stages',{k:round(v,4) for k,v in d['roofline']['stage_ms'].items()})
PY
timeout 120 python tools/icp_prof.py > gpurun_out/r04a_icp_prof.txt 2>&1; tail -25 gpurun_out/r04a_icp_prof.txt
