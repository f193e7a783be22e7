#!/bin/bash
# round 3, call k: threads per tile workgroup (256 / 512 / 1024) x lanes per sprite: bit-identity test, stage times
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "launch_shapes or tiled_splat or overflow" > gpurun_out/r03k_pytest.log 2>&1; tail -2 gpurun_out/r03k_pytest.log
for T in 256 512 1024; do for L in 1 4; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-input --frame-cache /tmp/mf_frames --param spriteLanes=$L --param tileThreads=$T > gpurun_out/r03k_bench_${T}_$L.json 2> gpurun_out/r03k_bench_${T}_$L.err
python - $T $L <<'PY'
import json,sys
T,L=sys.argv[1:3]
d=json.loads(open(f'gpurun_out/r03k_bench_{T}_{L}.json').read().strip().splitlines()[-1])
r=d['roofline']; print('threads', T, 'lanes', L, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'IndexMap::ACTIVE (splat) ms', round(r['stage_ms']['IndexMap::ACTIVE'],4), 'surfels', d['config'].get('surfels'))
PY
done; done
