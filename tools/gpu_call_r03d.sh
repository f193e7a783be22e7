#!/bin/bash
# round 3, call d: branch-free bilateral (parity + time), bench line with the measured copy ceiling
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "bilateral or preproc or pipeline or kernels" > gpurun_out/r03d_pytest.log 2>&1; tail -3 gpurun_out/r03d_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err; cut -c1-300 gpurun_out/r03d_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('us/launch', r['us_per_launch'], 'frac', r['frac'], 'ceiling', r.get('measured_ceiling')); print({k:round(v,4) for k,v in r['stage_ms'].items()})
PY
