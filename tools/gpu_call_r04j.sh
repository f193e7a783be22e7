#!/bin/bash
# round 4, closing call: the single-model long-horizon tests and the RGB-D tests through the frame graph (default on), then the bench line at HEAD
mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_gpu_rgbd.py tests/test_gpu_parity_long.py tests/test_gpu_emu_agrees.py -m gpu -q -s --durations=5 -k "rgbd or long_horizon or emu or so3 or frame_to_frame or rgb" > gpurun_out/r04j_pytest.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR\|frames: ATE RMSE\|RGB-D + SO(3), " gpurun_out/r04j_pytest.log | cut -c1-330 | tail -8
grep -n "^E  " gpurun_out/r04j_pytest.log | head -6 | cut -c1-250
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04j_bench.json 2> gpurun_out/r04j_bench.err; tail -2 gpurun_out/r04j_bench.err | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j_bench.json')); h=d['host_input']
print('value',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'host_input',round(h['value'],1),h.get('host_us_inside_the_call'),'variant',round(d['variants']['reference_default']['value'],1),'cpu',round(d['cpu_baseline']['value'],2))
PY
