#!/bin/bash
# Round 6, GPU call h: kernel trace of S2 on one GPU (--config 2s)
TAG=${1:-r06h}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 300 python bench.py --config 2s --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline > gpurun_out/${TAG}_2s.json 2> gpurun_out/${TAG}_2s.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_2s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_2s -o s -- python $REPO/bench.py --config 2s --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-variants --no-host-input --no-cpu-baseline --no-roofline --steps 120 --warmup 60 > /tmp/prof_2s.log 2>&1
cp $(find /tmp/prof_2s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_2s_kernel_stats.csv
cd $REPO
python -c "
import json; d = json.load(open('gpurun_out/${TAG}_2s.json')); print('2s', d['value'], d['ms_per_step'], d.get('stage_ms'))"
head -45 gpurun_out/${TAG}_2s_kernel_stats.csv | cut -c1-140
