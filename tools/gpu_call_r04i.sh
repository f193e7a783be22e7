#!/bin/bash
# round 4: where the host-pointer call spends its host time, frame graph on / off (steady state: captures excluded)
mkdir -p gpurun_out
for fg in 1 0; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-variants --param frameGraph=$fg > gpurun_out/r04i_bench_fg$fg.json 2>gpurun_out/r04i_bench_fg$fg.err
done
python - <<'PY'
import json
for n in (1,0):
    d=json.load(open(f'gpurun_out/r04i_bench_fg{n}.json')); h=d['host_input']
    print('frameGraph',n,'value',round(d['value'],1),'host_input',round(h['value'],1),'host ms/call',round(h['host_ms_per_call'],4),{k:round(v,1) for k,v in h['host_us_inside_the_call'].items()})
PY
