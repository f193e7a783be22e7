#!/bin/bash
# Round 5, first GPU call: configs[4] as specified (dense maps) -- parity test, bench line, rocprofv3 kernel trace.
TAG=${1:-r05a}
REPO=$(pwd)
mkdir -p gpurun_out
(lscpu | head -20; nproc; free -g | head -2) > gpurun_out/${TAG}_gpu_box_host.txt
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity_long.py::test_config4_dense_maps -x -q -s -m gpu -n 0 > gpurun_out/${TAG}_pytest_c4.log 2>&1
echo "pytest c4 rc=$? $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/${TAG}_pytest_c4.log
t0=$(date +%s)
timeout 600 python bench.py --config 4 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench c4 rc=$? $(( $(date +%s) - t0 )) s"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o s -- python $REPO/bench.py --config 4 --frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0 --steps 20 > /tmp/prof_c4.log 2>&1
echo "rocprof c4 rc=$?"
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_c4_kernel_stats.csv
cd $REPO
tail -5 /tmp/prof_c4.log | cut -c1-300
cut -c1-1500 gpurun_out/${TAG}_bench_c4.json
tail -5 gpurun_out/${TAG}_bench_c4.err
head -40 gpurun_out/${TAG}_c4_kernel_stats.csv | cut -c1-150
tail -30 gpurun_out/${TAG}_pytest_c4.log | cut -c1-300
