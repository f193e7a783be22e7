#!/bin/bash
# round 4: the whole-frame hipGraph of the host-pointer path on the hardware -- parity with the eager launches, then what it does to host_input
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_pipeline.py tests/test_gpu_facade.py tests/test_gpu_switches.py tests/test_gpu_f4_bound.py tests/test_gpu_gn_graph.py -m gpu -q -s --durations=6 > gpurun_out/r04h_pytest.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04h_pytest.log | tail -6
grep -n "^E  " gpurun_out/r04h_pytest.log | head -8 | cut -c1-250
for fg in 1 0; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --param frameGraph=$fg > gpurun_out/r04h_bench_fg$fg.json 2>gpurun_out/r04h_bench_fg$fg.err
done
python - <<'PY'
import json
for n in (1,0):
    d=json.load(open(f'gpurun_out/r04h_bench_fg{n}.json')); h=d['host_input']; v=d['variants']['reference_default'] if d.get('variants') else None
    print('frameGraph',n,'value',round(d['value'],1),'host_input',round(h['value'],1),'host ms/call',round(h['host_ms_per_call'],4),'variant',v and round(v['value'],1))
PY
timeout 200 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r04h_bench_1280x960.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h_bench_1280x960.json')); print('1280x960 value',round(d['value'],1),'host_input',round(d['host_input']['value'],1))
PY
