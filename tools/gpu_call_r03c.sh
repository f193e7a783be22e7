#!/bin/bash
# Third GPU call of round 3: the suite at HEAD (batched object passes, frame-pyramid contraction fix, index-list splat again), the
# Gauss-Newton prologue microbenchmark, config 2s with the object passes batched / model by model, the kernel trace on r02's arguments.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/r03c_pytest.log 2>&1; tail -4 gpurun_out/r03c_pytest.log
grep -E "frames: ATE|identical on|agreed within|^FAILED|^ERROR" gpurun_out/r03c_pytest.log | head -30
grep -E "ids oracle/hip" gpurun_out/r03c_pytest.log | awk '{for(i=1;i<=NF;i++) if($i=="diff" && $(i-1)=="label") print $(i+1)}' | sort -g | tail -3
(cd tools/micro && timeout 60 ./gn_chain) > gpurun_out/r03c_gn_chain.txt 2>&1; cat gpurun_out/r03c_gn_chain.txt
for P in "" "--param batchObjectPasses=0"; do
  timeout 200 python bench.py --config 2s --no-cpu-baseline --no-host-input $P 2>> gpurun_out/r03c_bench.err | cut -c1-230 | tee -a gpurun_out/r03c_bench_2s_ab.txt
done
timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03c_bench.json 2>> gpurun_out/r03c_bench.err; cut -c1-300 gpurun_out/r03c_bench.json
bash tools/kstats.sh r03c 2>&1 | tail -22
