#!/bin/bash
# round 3, closing call: the tests that touch the tile passes / loop-level calls at HEAD, then the measurement artefacts again (tag r03)
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q -k "pipeline or surfel_passes or multimodel or sharded or glsl or api" > gpurun_out/r03z2_pytest.log 2>&1; tail -3 gpurun_out/r03z2_pytest.log
bash tools/profile_round_r03.sh r03 > gpurun_out/r03_profile_round.log 2>&1; tail -4 gpurun_out/r03_profile_round.log | cut -c1-160
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; cut -c1-200 gpurun_out/r03_bench_final.json
timeout 200 python bench.py --config 2s --steps 20 --warmup 5 --frame-cache /tmp/mf_frames_2s > gpurun_out/r03_bench_2s.json 2>/dev/null; cut -c1-300 gpurun_out/r03_bench_2s.json
