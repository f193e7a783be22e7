#!/bin/bash
# Second GPU call of round 3: the whole suite at HEAD (exact-normal maps, F4 counters, frameToFrameRGB, object bounding boxes, the re-gated
# long-horizon / 8-object tests), the in-kernel phase stamps of the Gauss-Newton launches, bench.py as the driver runs it, a kernel trace on
# the SAME arguments as profiles/r02_final_kernel_stats.csv (--steps 100 --warmup 60 --frames 60), and a one-GPU rehearsal of the code path
# bench.py takes with WORLD_SIZE > 1 (sharded 8-object scene beside the weak-scaling line).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/r03b_pytest.log 2>&1; tail -4 gpurun_out/r03b_pytest.log
grep -E "frames: ATE|identical on|agreed within|^FAILED|^ERROR|ill iterations hip / oracle (1[0-9]|[1-9]) " gpurun_out/r03b_pytest.log | head -30
timeout 120 python tools/icp_prof.py > gpurun_out/r03b_icp_prof.txt 2>&1; tail -24 gpurun_out/r03b_icp_prof.txt
timeout 240 python bench.py --steps 20 --warmup 5 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; cut -c1-300 gpurun_out/r03b_bench.json
bash tools/kstats.sh r03b 2>&1 | tail -24
timeout 400 python bench.py --force-sharded-scene --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/r03b_bench_scene_n1.json 2> gpurun_out/r03b_bench_scene_n1.err; python -c "
import json; d=json.load(open('gpurun_out/r03b_bench_scene_n1.json')); print('value', d['value'], 'ranks_seen', d['ranks_seen']['world_size'], 'scene', json.dumps(d['sharded_scene'])[:600])" || tail -5 gpurun_out/r03b_bench_scene_n1.err
