#!/bin/bash
# First GPU call of round 3 (run through gpurun from the repo root; every step under its own timeout, nothing can hang the box: the two
# experimental kernels have bounded spins).  It answers, in this order:
#   1. is HEAD green on hardware (literal fusion weight now the default, the round-3 long-horizon / 8-object / 1280x960 parity tests)?  r03a_pytest.log
#   2. what does a device-wide barrier cost against a dependent launch?   gpurun_out/r03a_micro.txt   
#   3. does the persistent Gauss-Newton launch work on hardware, and is it faster?   gpurun_out/r03a_persist.log
#   4. bench.py (what the driver runs) + A/B of the two switches    gpurun_out/r03a_bench*.json
#   4b. config 2s (multi-model) with the object-model launch switches        gpurun_out/r03a_bench_2s_ab.txt
#   5. kernel trace of the default workload                          gpurun_out/r03a_kernel_stats.csv
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/r03a_pytest.log 2>&1; tail -4 gpurun_out/r03a_pytest.log
grep -E "frames: ATE|^frame +[0-9]+:|identical on|count relative|^FAILED|^ERROR" gpurun_out/r03a_pytest.log | head -40
(cd tools/micro && for f in launch_floor grid_barrier; do [ -x $f ] || timeout 120 hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $f.hip -o $f; done)
(timeout 60 tools/micro/launch_floor; timeout 60 tools/micro/grid_barrier) > gpurun_out/r03a_micro.txt 2>&1; cat gpurun_out/r03a_micro.txt
MF_TEST_PERSISTENT=1 timeout 300 python -m pytest tests/test_gpu_persistent_icp.py -x -q -s > gpurun_out/r03a_persist.log 2>&1; grep -E "iterations|passed|failed|rror" gpurun_out/r03a_persist.log | head
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; cut -c1-260 gpurun_out/r03a_bench.json
timeout 120 python bench.py --no-cpu-baseline --no-host-input --param persistentIcp=1 > gpurun_out/r03a_bench_persist.json 2>> gpurun_out/r03a_bench.err; cut -c1-200 gpurun_out/r03a_bench_persist.json
timeout 120 python bench.py --no-cpu-baseline --param gnLoopGraph=1 > gpurun_out/r03a_bench_graph.json 2>> gpurun_out/r03a_bench.err; cut -c1-200 gpurun_out/r03a_bench_graph.json
# multi-model frames (S2: background + object models on one GPU): the two launch-overhead switches for object models
for P in "" "--param objectSmallGrids=1" "--param objectScatterSplat=1" "--param objectSmallGrids=1 --param objectScatterSplat=1"; do
  timeout 150 python bench.py --config 2s --no-cpu-baseline --no-host-input $P 2>> gpurun_out/r03a_bench.err | cut -c1-220 | tee -a gpurun_out/r03a_bench_2s_ab.txt
done
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_s
B="--frames 60 --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline"   # no fork under the profiler
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 200 --warmup 60 $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/r03a_kernel_stats.csv; cd $REPO; head -8 gpurun_out/r03a_kernel_stats.csv | cut -c1-150
