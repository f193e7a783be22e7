#!/bin/bash
# Collects the measurement artefacts of one round on the GPU box (run through gpurun from the repo root):
#   gpurun_out/<tag>_bench.json            bench.py line (N = 1, default flags: what the driver runs)
#   gpurun_out/<tag>_kernel_stats.csv      timeout 300 rocprofv3 --kernel-trace --stats summary of the same workload
#   gpurun_out/<tag>_pmc_fetch.csv / _pmc_write.csv   per-kernel FETCH_SIZE / WRITE_SIZE (separate --pmc passes, no trace domain
#                                          besides --kernel-trace), averaged per kernel by tools/pmc_summary.py
#   gpurun_out/<tag>_calib_fetch.csv / _calib_write.csv   the same two passes over tools/micro/fetch_calib (known byte counts)
#   profiles/<tag>_pmc.json                corrected bytes per launch per kernel (tools/make_pmc_json.py) -- copy back by hand
TAG=${1:-r02}
REPO=$(pwd)
mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w /tmp/prof_cf /tmp/prof_cw
B="--frames 60 --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline"   # no fork under the profiler
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 200 --warmup 60 $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py --steps 20 --warmup 60 $B > /tmp/prof_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $REPO/bench.py --steps 20 --warmup 60 $B > /tmp/prof_w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_cf -o f -- $REPO/tools/micro/fetch_calib > /tmp/prof_cf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_cw -o w -- $REPO/tools/micro/fetch_calib > /tmp/prof_cw.log 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) 400 > $REPO/gpurun_out/${TAG}_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) 400 > $REPO/gpurun_out/${TAG}_pmc_write.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cf -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cw -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_write.csv
cd $REPO
mkdir -p gpurun_out/profiles_out
python tools/make_pmc_json.py ${TAG} gpurun_out/${TAG}_pmc_fetch.csv gpurun_out/${TAG}_pmc_write.csv gpurun_out/${TAG}_calib_fetch.csv gpurun_out/${TAG}_calib_write.csv "$(cat .tree_id 2>/dev/null)" && cp profiles/${TAG}_pmc.json gpurun_out/
tail -3 /tmp/prof_f.log | cut -c1-200
cut -c1-400 gpurun_out/${TAG}_bench.json
(lscpu | head -20; nproc) > gpurun_out/${TAG}_gpu_box_host.txt
