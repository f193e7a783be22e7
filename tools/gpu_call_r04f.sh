#!/bin/bash
# round 4, final call: the whole -m gpu suite at HEAD with its prints (-s), the bench lines at HEAD, the reference-default kernel trace after the
# small-launch fusion, and the N = 1 rehearsal of the multi-GPU code path (sharded scene with the packed collectives)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --durations=12 > gpurun_out/r04f_pytest.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04f_pytest.log | tail -6
grep -n "teacher-forced:\|RGB-D + SO(3), \|frames: ATE RMSE\|own filters, 20\|configs\[0\] stand-in\|configs\[2\] stand-in" gpurun_out/r04f_pytest.log | cut -c1-420
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; tail -2 gpurun_out/r04f_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04f_bench.json'))
print('value',d['value'],'frac',d['roofline']['frac'],'host_input',d['host_input']['value'],d['host_input'].get('host_ms_per_call'),'variant',d['variants']['reference_default']['value'])
PY
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r04f_bench_2s.json 2>/dev/null
timeout 400 python bench.py --force-sharded-scene --steps 20 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/r04f_bench_scene_n1.json 2> gpurun_out/r04f_bench_scene_n1.err; tail -2 gpurun_out/r04f_bench_scene_n1.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04f_bench_2s.json')); print('2s',d['value'],d['ms_per_step'],d['config']['models'],'host_input',d['host_input'] and d['host_input']['value'])
d=json.load(open('gpurun_out/r04f_bench_scene_n1.json')); print('scene n1',json.dumps(d.get('sharded_scene'))[:600])
PY
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_r
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- python $REPO/bench.py --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --no-variants --steps 600 --warmup 60 --icp-weight 20 --so3 > /tmp/prof_r.log 2>&1
cp $(find /tmp/prof_r -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/r04f_rgbd_kernel_stats.csv; cd $REPO; head -14 gpurun_out/r04f_rgbd_kernel_stats.csv | cut -c1-110
