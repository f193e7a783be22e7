#!/bin/bash
# round 4, fourth call: the whole -m gpu suite at HEAD (with durations), then the measurement artefacts (tools/profile_round_r04.sh) and the
# 2s / 1280x960 bench lines
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r04d_pytest.log 2>&1; tail -32 gpurun_out/r04d_pytest.log | cut -c1-200
bash tools/profile_round_r04.sh r04
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r04_bench_2s.json 2>/dev/null; cut -c1-330 gpurun_out/r04_bench_2s.json
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 > gpurun_out/r04_bench_1280x960.json 2>/dev/null; cut -c1-330 gpurun_out/r04_bench_1280x960.json
timeout 120 python tools/icp_prof.py > gpurun_out/r04_icp_prof.txt 2>&1; tail -44 gpurun_out/r04_icp_prof.txt
