#!/bin/bash
# Round 5, GPU call w: the 18-model scene (--config 2s) on the final tree: bench line + rocprofv3 kernel table
TAG=${1:-r05}
mkdir -p gpurun_out
timeout 300 python bench.py --config 2s --no-cpu-baseline > gpurun_out/${TAG}_bench_2s.json 2> gpurun_out/${TAG}_bench_2s.err
echo "bench 2s rc=$?"; cut -c1-260 gpurun_out/${TAG}_bench_2s.json
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_2s -o m -- python $REPO/bench.py --config 2s --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --steps 240 --warmup 60 > /tmp/prof_2s.log 2>&1
cp $(find /tmp/prof_2s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_2s_kernel_stats.csv
head -12 $REPO/gpurun_out/${TAG}_2s_kernel_stats.csv | cut -c1-150
