#!/bin/bash
# round 3, call h: lanes per sprite in the tile z-test (1 / 2 / 4 / 8 / 16): parity at the default, stage times for each
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "surfel_passes or glsl or multimodel" > gpurun_out/r03j_pytest.log 2>&1; tail -2 gpurun_out/r03j_pytest.log
for L in 1 2 4 8; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-input --frame-cache /tmp/mf_frames --param spriteLanes=$L > gpurun_out/r03j_bench_$L.json 2> gpurun_out/r03j_bench_$L.err
python - $L <<'PY'
import json,sys
L=sys.argv[1]
d=json.loads(open(f'gpurun_out/r03j_bench_{L}.json').read().strip().splitlines()[-1])
r=d['roofline']; print('lanes', L, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'IndexMap::ACTIVE (splat) ms', round(r['stage_ms']['IndexMap::ACTIVE'],4), 'surfels', d['config'].get('surfels'))
PY
done
PYTHONPATH=. python tools/splat_prof.py 300 2>&1 | tail -14
