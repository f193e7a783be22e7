#!/bin/bash
# round 4, fifth call: the own-filters multi-model gate, the host-pointer rate with the copy helper thread
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_long.py tests/test_gpu_api.py tests/test_gpu_facade.py -m gpu -q -s --durations=5 -k "own_filters or eight_objects_tracked or api or facade" > gpurun_out/r04e_pytest.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR\|own filters, 20\|teacher-forced:\|comparable\|identical on" gpurun_out/r04e_pytest.log | cut -c1-300 | tail -12
grep -n "^E " gpurun_out/r04e_pytest.log | head -10 | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err; tail -2 gpurun_out/r04e_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e_bench.json'))
print('value',d['value'],'host_input',d['host_input']['value'],d['host_input']['ms_per_step'])
PY
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r04e_bench_2s.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e_bench_2s.json'))
print('2s value',d['value'],'ms',d['ms_per_step'],'models',d['config']['models'],'host_input',d['host_input'] and d['host_input']['value'])
PY
python - <<'PY'
# where the host-pointer call spends its time: the staging copy alone, pageable numpy -> pinned, one and two threads
import time, numpy as np, ctypes, threading
n=7*640*480
src=[np.random.randint(0,255,n,dtype=np.uint8) for _ in range(64)]
import torch
dst=torch.empty(n,dtype=torch.uint8).pin_memory()
d=dst.numpy()
t0=time.perf_counter()
for k in range(256): np.copyto(d, src[k%64])
t=(time.perf_counter()-t0)/256
print('staging copy of 2.15 MB, one thread: %.1f us (%.1f GB/s)'%(t*1e6, n/t/1e9))
PY
