#!/bin/bash
# Measurement artefacts of round 3 (run through gpurun from the repo root).  Every pass runs the SAME workload as the timing line: the
# 600-frame S1 stream of configs[1], rendered once (bench.py --frame-cache) and re-used by the profiler passes, 60 warm-up + 600 timed frames:
#   gpurun_out/<tag>_bench.json            bench.py as the driver runs it (--steps 20 --warmup 5)
#   gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>_pmc_fetch.csv / _pmc_write.csv     separate --pmc FETCH_SIZE / WRITE_SIZE passes (no other trace domain), per-kernel
#                                          means over the last 4000 launches of each kernel (the map at its working size)
#   gpurun_out/<tag>_calib_*.csv           the same two passes over tools/micro/fetch_calib (known byte counts)
#   gpurun_out/<tag>_pmc.json              corrected bytes per launch per kernel (tools/make_pmc_json.py) -> copy to profiles/
TAG=${1:-r03}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
(cd tools/micro && [ -x fetch_calib ] || timeout 120 hipcc --offload-arch=gfx950 -O3 -Wno-unused-result fetch_calib.hip -o fetch_calib)
timeout 400 python bench.py --steps 20 --warmup 5 --frame-cache $CACHE > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w /tmp/prof_cf /tmp/prof_cw
B="--frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --steps 600 --warmup 60"   # no fork under the profiler
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py $B > /tmp/prof_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $REPO/bench.py $B > /tmp/prof_w.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_cf -o f -- $REPO/tools/micro/fetch_calib > /tmp/prof_cf.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_cw -o w -- $REPO/tools/micro/fetch_calib > /tmp/prof_cw.log 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_pmc_write.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cf -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cw -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_write.csv
cd $REPO
python tools/make_pmc_json.py ${TAG} gpurun_out/${TAG}_pmc_fetch.csv gpurun_out/${TAG}_pmc_write.csv gpurun_out/${TAG}_calib_fetch.csv gpurun_out/${TAG}_calib_write.csv "$(git rev-parse --short HEAD 2>/dev/null || cat .tree_id 2>/dev/null)" && cp profiles/${TAG}_pmc.json gpurun_out/
tail -3 /tmp/prof_f.log | cut -c1-200
cut -c1-400 gpurun_out/${TAG}_bench.json
(lscpu | head -20; nproc) > gpurun_out/${TAG}_gpu_box_host.txt
head -16 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-120
