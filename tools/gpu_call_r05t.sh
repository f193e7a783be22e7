#!/bin/bash
# Round 5, GPU call t: the two bench lines once more, now that profiles/r05_* (which their roofline entries cite) are the final tree's
TAG=${1:-r05}
mkdir -p gpurun_out
timeout 500 python bench.py --steps 20 --warmup 5 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench default rc=$?"
timeout 500 python bench.py --config 4 --frame-cache /tmp/mf_frames > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench c4 rc=$?"
cut -c1-400 gpurun_out/${TAG}_bench.json; echo; cut -c1-300 gpurun_out/${TAG}_bench_c4.json
