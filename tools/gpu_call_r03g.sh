#!/bin/bash
# round 3, call f: XCD-contiguous tile order for k_bilateral / k_frame_pyramid: parity, time, FETCH_SIZE
mkdir -p gpurun_out
REPO=$(pwd)
timeout 600 python -m pytest tests -m gpu -x -q -k "surfel_passes or pipeline or multimodel or glsl" > gpurun_out/r03g_pytest.log 2>&1; tail -2 gpurun_out/r03g_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frame-cache /tmp/mf_frames > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('fps', d['value'], 'ms', d['ms_per_step'], 'us/launch', r['us_per_launch']); print({k:round(v,4) for k,v in r['stage_ms'].items()})
PY
cd /tmp && export TMPDIR=/tmp
B="--frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --steps 200 --warmup 20"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py $B > /tmp/prof_f.log 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) 150 > $REPO/gpurun_out/r03g_pmc_fetch.csv
grep -i "splat\|global_tile\|bilateral" $REPO/gpurun_out/r03g_pmc_fetch.csv
rm -rf /tmp/prof_s; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/r03g_kernel_stats.csv; grep -i "splat\|clean_flags" $REPO/gpurun_out/r03g_kernel_stats.csv | cut -c1-100
