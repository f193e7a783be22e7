#!/bin/bash
# round 4, last call: the multi-model tests after the lane-split object scatter + the own-filters gate, the 2s line, host-input A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multimodel.py tests/test_gpu_sharded.py tests/test_gpu_parity_long.py tests/test_gpu_switches.py -m gpu -q -s --durations=5 \
    -k "not long_horizon and not config4" > gpurun_out/r04g_pytest.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR\|own filters, 20\|teacher-forced:" gpurun_out/r04g_pytest.log | cut -c1-300 | tail -8
grep -n "^E  " gpurun_out/r04g_pytest.log | head -6 | cut -c1-250
bash tools/kstats.sh r04g_2s --config 2s 2>&1 | head -16
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r04g_bench_2s.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g_bench_2s.json')); print('2s',d['value'],d['ms_per_step'],d['config']['models'],'host_input',d['host_input'] and d['host_input']['value'])
PY
timeout 200 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-roofline --param hostInputAsync=0 > gpurun_out/r04g_bench_blocking.json 2>/dev/null
timeout 200 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-roofline > gpurun_out/r04g_bench_async.json 2>/dev/null
python - <<'PY'
import json
for n in ('blocking','async'):
    d=json.load(open(f'gpurun_out/r04g_bench_{n}.json')); h=d['host_input']; print(n,'value',round(d['value'],1),'host_input',round(h['value'],1),'host ms/call',round(h['host_ms_per_call'],4))
PY
