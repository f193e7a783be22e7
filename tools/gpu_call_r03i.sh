#!/bin/bash
# round 3, call h: lanes per sprite in the tile z-test (1 / 2 / 4 / 8 / 16): parity at the default, stage times for each
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "surfel_passes or pipeline or glsl" > gpurun_out/r03i_pytest.log 2>&1; tail -2 gpurun_out/r03i_pytest.log
for L in 1 4; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-input --frame-cache /tmp/mf_frames --param spriteLanes=$L > gpurun_out/r03i_bench_$L.json 2> gpurun_out/r03i_bench_$L.err
python - $L <<'PY'
import json,sys
L=sys.argv[1]
d=json.loads(open(f'gpurun_out/r03i_bench_{L}.json').read().strip().splitlines()[-1])
r=d['roofline']; print('lanes', L, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'IndexMap::ACTIVE (splat) ms', round(r['stage_ms']['IndexMap::ACTIVE'],4), 'surfels', d['config'].get('surfels'))
PY
done
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
B="--frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --steps 120 --warmup 20"
rm -rf /tmp/prof_q; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_q -o q -- python $REPO/bench.py $B > /tmp/prof_q.log 2>&1
tail -2 /tmp/prof_q.log | cut -c1-200
python $REPO/tools/pmc_summary.py $(find /tmp/prof_q -name "*counter_collection.csv" | head -1) 100 > $REPO/gpurun_out/r03i_pmc_sq.csv
grep -i "splat_tile\|clean_flags\|index_resolve" $REPO/gpurun_out/r03i_pmc_sq.csv
