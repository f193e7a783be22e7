#!/bin/bash
# Round 5, GPU call e: two-sweep clean with 2048-element chunks.
TAG=${1:-r05e}
REPO=$(pwd)
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_surfel_passes.py tests/test_gpu_glsl_passes.py tests/test_gpu_multimodel.py tests/test_gpu_pipeline.py tests/test_gpu_api.py \
   tests/test_gpu_labels.py tests/test_gpu_sharded.py tests/test_gpu_switches.py \
   "tests/test_gpu_parity_long.py::test_config4_dense_maps" "tests/test_gpu_parity_long.py::test_s2_eight_objects_tracked_teacher_forced" \
   -x -q -m gpu -n 8 --durations=8 > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/${TAG}_pytest.log
t0=$(date +%s)
timeout 600 python bench.py --config 4 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench c4 rc=$? $(( $(date +%s) - t0 )) s"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4 /tmp/prof_rt /tmp/prof_v
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o s -- python $REPO/bench.py --config 4 --frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0.5 --steps 30 > /tmp/prof_c4.log 2>&1
echo "rocprof c4 rc=$?"
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_c4_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o s -- python $REPO/bench.py --frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0 --steps 300 --warmup 60 --no-cpu-baseline --no-host-input --no-roofline --no-variants > /tmp/prof_v.log 2>&1
cp $(find /tmp/prof_v -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
cd $REPO
t0=$(date +%s)
timeout 600 python bench.py --steps 20 --warmup 5 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench default rc=$? $(( $(date +%s) - t0 )) s"
cut -c1-400 gpurun_out/${TAG}_bench_c4.json
tail -3 gpurun_out/${TAG}_bench_c4.err
head -24 gpurun_out/${TAG}_c4_kernel_stats.csv | cut -c1-130
head -20 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-130
cut -c1-300 gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_bench.err
tail -16 gpurun_out/${TAG}_pytest.log | cut -c1-250
