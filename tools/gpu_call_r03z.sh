#!/bin/bash
# round 3, final call: the whole -m gpu suite at HEAD, then the measurement artefacts (tools/profile_round_r03.sh)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03z_pytest.log 2>&1; tail -4 gpurun_out/r03z_pytest.log
bash tools/profile_round_r03.sh r03
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; cut -c1-200 gpurun_out/r03_bench_final.json
timeout 300 python bench.py --config 2s --steps 20 --warmup 5 > gpurun_out/r03_bench_2s.json 2>/dev/null; cut -c1-330 gpurun_out/r03_bench_2s.json
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 > gpurun_out/r03_bench_1280x960.json 2>/dev/null; cut -c1-330 gpurun_out/r03_bench_1280x960.json
