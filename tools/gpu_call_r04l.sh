#!/bin/bash
# round 4: the host-pointer path's timeline once more, keeping the traced run's own bench line and a six-frame merged timeline; the same
# with eager launches; an untraced host_input over 600 frames
REPO=$(pwd); mkdir -p gpurun_out
for fg in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h$fg && timeout 150 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_h$fg -o h -- \
      python $REPO/bench.py --steps 150 --warmup 40 --min-seconds 0 --frames 24 --gen-workers 1 --no-cpu-baseline --no-roofline --no-variants --param frameGraph=$fg \
      > $REPO/gpurun_out/r04l_traced_bench_fg$fg.json 2> /tmp/prof_h$fg.err )
  python tools/host_trace_summary.py /tmp/prof_h$fg 150 gpurun_out/r04l_timeline_fg$fg.csv > gpurun_out/r04l_host_trace_fg$fg.json 2> gpurun_out/r04l_host_trace_fg$fg.err; tail -2 gpurun_out/r04l_host_trace_fg$fg.err
  python - $fg <<'PY'
import json,sys
fg=sys.argv[1]
try:
    d=json.load(open(f'gpurun_out/r04l_host_trace_fg{fg}.json'))
    for k in ('device_resident','host_input'):
        s=d[k]; print('fg',fg,k,'period',s['period_us_median'],'busy',round(s['kernel_busy_us_mean'],1),'gap->next',s['gap_to_next_frame_us_median'],'span/frame',s.get('span_us_per_frame'))
    print('  uploads',d['uploads']['us_mean'],d['uploads'].get('start_us_into_the_frame_it_runs_under'),d['uploads']['kernels_they_ran_under'])
    b=json.load(open(f'gpurun_out/r04l_traced_bench_fg{fg}.json')); h=b['host_input']
    print('  traced bench: value',round(b['value'],1),'host_input',round(h['value'],1),{k:round(v,1) for k,v in h['host_us_inside_the_call'].items()})
except Exception as e: print('failed',e)
PY
done
timeout 100 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-roofline --host-input-frames 600 > gpurun_out/r04l_bench_600.json 2>/dev/null
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r04l_bench_600.json')); h=d['host_input']
    print('600 host frames: value',round(d['value'],1),'host_input',round(h['value'],1),{k:round(v,1) for k,v in h['host_us_inside_the_call'].items()})
except Exception as e: print('bench failed', e)
PY
