#!/bin/bash
# Collects the measurement artefacts of one round on the GPU box (run through gpurun from the repo root):
#   gpurun_out/<tag>_bench.json            bench.py line (N = 1)
#   gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the same command
#   gpurun_out/<tag>_pmc_fetch.csv / _pmc_write.csv   per-kernel FETCH_SIZE / WRITE_SIZE (separate --pmc passes, no trace domains
#                                          besides --kernel-trace), averaged per kernel by tools/pmc_summary.py
TAG=${1:-r01}
REPO=$(pwd)
mkdir -p gpurun_out
python bench.py --steps 300 --warmup 30 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_w.log 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_pmc_write.csv
tail -3 /tmp/prof_f.log
cat $REPO/gpurun_out/${TAG}_bench.json | cut -c1-300
