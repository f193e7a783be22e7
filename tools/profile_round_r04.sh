#!/bin/bash
# Measurement artefacts of round 4 (run through gpurun from the repo root).  Every pass of the single-model part runs the SAME workload as the
# timing line: the 600-frame S1 stream of configs[1], rendered once (bench.py --frame-cache) and re-used by the profiler passes.
#   gpurun_out/<tag>_bench.json                bench.py as the driver runs it (--steps 20 --warmup 5), with its `variants` entry
#   gpurun_out/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats, configs[1]
#   gpurun_out/<tag>_rgbd_kernel_stats.csv     the same for the reference-default variant (--icp-weight 20 --so3): k_rgbd_iter / k_rgb_step / k_so3_iter
#   gpurun_out/<tag>_2s_kernel_stats.csv       the same for the multi-model frame (--config 2s)
#   gpurun_out/<tag>_pmc_fetch.csv / _pmc_write.csv / _calib_*.csv / <tag>_pmc.json      separate --pmc FETCH_SIZE / WRITE_SIZE passes + calibration
#   gpurun_out/<tag>_rgbd_pmc_fetch.csv / _rgbd_pmc_write.csv                            the same two passes over the reference-default variant
TAG=${1:-r04}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
(cd tools/micro && [ -x fetch_calib ] || timeout 120 hipcc --offload-arch=gfx950 -O3 -Wno-unused-result fetch_calib.hip -o fetch_calib)
timeout 400 python bench.py --steps 20 --warmup 5 --frame-cache $CACHE > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w /tmp/prof_cf /tmp/prof_cw /tmp/prof_r /tmp/prof_rf /tmp/prof_rw /tmp/prof_2s
B="--frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --no-variants --steps 600 --warmup 60"   # no fork under the profiler
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- python $REPO/bench.py $B --icp-weight 20 --so3 > /tmp/prof_r.log 2>&1
cp $(find /tmp/prof_r -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_rgbd_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_2s -o m -- python $REPO/bench.py --config 2s --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --steps 240 --warmup 60 > /tmp/prof_2s.log 2>&1
cp $(find /tmp/prof_2s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_2s_kernel_stats.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py $B > /tmp/prof_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $REPO/bench.py $B > /tmp/prof_w.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_rf -o f -- python $REPO/bench.py $B --icp-weight 20 --so3 > /tmp/prof_rf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_rw -o w -- python $REPO/bench.py $B --icp-weight 20 --so3 > /tmp/prof_rw.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_cf -o f -- $REPO/tools/micro/fetch_calib > /tmp/prof_cf.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_cw -o w -- $REPO/tools/micro/fetch_calib > /tmp/prof_cw.log 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_pmc_write.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_rf -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_rgbd_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_rw -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_rgbd_pmc_write.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cf -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cw -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_write.csv
cd $REPO
python tools/make_pmc_json.py ${TAG} gpurun_out/${TAG}_pmc_fetch.csv gpurun_out/${TAG}_pmc_write.csv gpurun_out/${TAG}_calib_fetch.csv gpurun_out/${TAG}_calib_write.csv "$(git rev-parse --short HEAD 2>/dev/null || cat .tree_id 2>/dev/null)" && cp profiles/${TAG}_pmc.json gpurun_out/
tail -3 /tmp/prof_f.log | cut -c1-200
cut -c1-400 gpurun_out/${TAG}_bench.json
(lscpu | head -20; nproc) > gpurun_out/${TAG}_gpu_box_host.txt
head -16 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-120
head -12 gpurun_out/${TAG}_rgbd_kernel_stats.csv | cut -c1-120
