#!/bin/bash
# Round 6, GPU call d: kernel traces of configs[4] (tracked) with the objects' clean in place (bigMapElements = 1 M) and by default
TAG=${1:-r06d}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 300 python bench.py --config 4 --frame-cache $CACHE --min-seconds 0.5 > /dev/null 2>&1   # (renders + caches the frames)
cd /tmp && export TMPDIR=/tmp
for v in big1M default; do
  rm -rf /tmp/prof_$v
  P=""; [ $v = big1M ] && P="--param bigMapElements=1000000"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o s -- python $REPO/bench.py --config 4 --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --steps 20 $P > /tmp/prof_$v.log 2>&1
  python $REPO/tools/c4_dense_summary.py $(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1) > $REPO/gpurun_out/${TAG}_c4_${v}_kernel_stats.csv
  echo "== $v"; head -28 $REPO/gpurun_out/${TAG}_c4_${v}_kernel_stats.csv | cut -c1-110
done
