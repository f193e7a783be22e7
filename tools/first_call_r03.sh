#!/bin/bash
# First GPU call of round 3 (run through gpurun from the repo root; ~12 minutes of box time; every step under its own timeout, nothing
# can hang the box: the two experimental kernels have bounded spins).  It answers, in this order:
#   1. is HEAD green on hardware?                                   gpurun_out/r03a_pytest.log
#   2. what does a device-wide barrier cost against a dependent launch?   gpurun_out/r03a_micro.txt   
#   3. does the persistent Gauss-Newton launch work on hardware, and is it faster?   gpurun_out/r03a_persist.log
#   4. does the hipGraph replay of the loop work on hardware?       gpurun_out/r03a_graph.log
#   5. bench.py (what the driver runs) + A/B of the two switches    gpurun_out/r03a_bench*.json
#   5b. config 2s (multi-model) with the object-model launch switches        gpurun_out/r03a_bench_2s_ab.txt
#   6. kernel trace of the default workload                          gpurun_out/r03a_kernel_stats.csv
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r03a_pytest.log 2>&1; tail -3 gpurun_out/r03a_pytest.log
(cd tools/micro && for f in launch_floor grid_barrier; do [ -x $f ] || timeout 120 hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $f.hip -o $f; done)
(timeout 60 tools/micro/launch_floor; timeout 60 tools/micro/grid_barrier) > gpurun_out/r03a_micro.txt 2>&1; cat gpurun_out/r03a_micro.txt
MF_TEST_PERSISTENT=1 timeout 300 python -m pytest tests/test_gpu_persistent_icp.py -x -q -s > gpurun_out/r03a_persist.log 2>&1; grep -E "iterations|passed|failed|rror" gpurun_out/r03a_persist.log | head
timeout 300 python -m pytest tests/test_gpu_gn_graph.py -q -rxX > gpurun_out/r03a_graph.log 2>&1; tail -4 gpurun_out/r03a_graph.log
# finding F5: the whole suite with the literal fusion weight on both sides (oracle + every context); green => flip both defaults
MF_LITERAL_WEIGHT=1 timeout 600 python -m pytest tests -m gpu -q -k "not facade" > gpurun_out/r03a_pytest_literal_weight.log 2>&1   # (the compiled facade program keeps the library default); tail -2 gpurun_out/r03a_pytest_literal_weight.log
timeout 200 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; cut -c1-260 gpurun_out/r03a_bench.json
timeout 120 python bench.py --no-cpu-baseline --param persistentIcp=1 > gpurun_out/r03a_bench_persist.json 2>> gpurun_out/r03a_bench.err; cut -c1-200 gpurun_out/r03a_bench_persist.json
timeout 120 python bench.py --no-cpu-baseline --param gnLoopGraph=1 > gpurun_out/r03a_bench_graph.json 2>> gpurun_out/r03a_bench.err; cut -c1-200 gpurun_out/r03a_bench_graph.json
# multi-model frames (S2: background + object models on one GPU): the two launch-overhead switches for object models
for P in "" "--param objectSmallGrids=1" "--param objectScatterSplat=1" "--param objectSmallGrids=1 --param objectScatterSplat=1"; do
  timeout 150 python bench.py --config 2s --no-cpu-baseline --no-host-input $P 2>> gpurun_out/r03a_bench.err | cut -c1-220 | tee -a gpurun_out/r03a_bench_2s_ab.txt
done
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_s
B="--frames 60 --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline"   # no fork under the profiler
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 200 --warmup 60 $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/r03a_kernel_stats.csv; cd $REPO; head -8 gpurun_out/r03a_kernel_stats.csv | cut -c1-150
