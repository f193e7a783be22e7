#!/bin/bash
# round 4, last call (≈ 6 GPU-minutes): the packed single upload on the hardware (async ≡ blocking ≡ serial upload, graph ≡ eager), a timeline of
# the host-pointer path (kernel + memory-copy trace), and host_input with the upload overlapped / serial
REPO=$(pwd); mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_api.py -m gpu -q -k "asynchronous or frame_graph" > gpurun_out/r04k_pytest.log 2>&1; tail -2 gpurun_out/r04k_pytest.log | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h && timeout 150 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_h -o h -- \
    python $REPO/bench.py --steps 150 --warmup 40 --min-seconds 0 --frames 24 --gen-workers 1 --no-cpu-baseline --no-roofline --no-variants > /tmp/prof_h.log 2>&1 )
python tools/host_trace_summary.py /tmp/prof_h 150 > gpurun_out/r04k_host_trace.json 2> gpurun_out/r04k_host_trace.err; tail -3 gpurun_out/r04k_host_trace.err
head -c 1500 $(find /tmp/prof_h -name '*memory_copy_trace.csv' | head -1) > gpurun_out/r04k_memory_copy_head.csv 2>/dev/null
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r04k_host_trace.json'))
    for k in ('device_resident','host_input'):
        s=d[k]; print(k, 'period', s['period_us_median'], 'busy', round(s['kernel_busy_us_mean'],1), 'gap->next', s['gap_to_next_frame_us_median'], 'launches', s['launches_per_frame'])
    print('delta', {k:v for k,v in d['delta_us'].items() if abs(v)>0.3}); print('uploads', d['uploads']); print(d['copy_directions_seen'])
except Exception as e: print('trace summary failed', e)
PY
for v in 0 1; do
  timeout 100 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-roofline --param hostUploadOnMain=$v > gpurun_out/r04k_bench_main$v.json 2>/dev/null
  python - $v <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r04k_bench_main{sys.argv[1]}.json')); h=d['host_input']
    print('uploadOnMain',sys.argv[1],'value',round(d['value'],1),'host_input',round(h['value'],1),{k:round(v,1) for k,v in h['host_us_inside_the_call'].items()})
except Exception as e: print('bench failed', e)
PY
done
