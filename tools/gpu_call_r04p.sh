#!/bin/bash
# round 4, closing call: the host-path tests and the default bench line at the final defaults (hostLockstep on, eager launches)
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_gpu_api.py -m gpu -q -k "asynchronous or frame_graph" 2>&1 | tail -2
timeout 90 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > gpurun_out/r04p_bench.json 2> gpurun_out/r04p_bench.err; tail -1 gpurun_out/r04p_bench.err | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04p_bench.json')); h=d['host_input']
print('value',round(d['value'],1),'frac',d['roofline'] and round(d['roofline']['frac'],4),'host_input',round(h['value'],1),{k:round(v,1) for k,v in h['host_us_inside_the_call'].items()})
PY
