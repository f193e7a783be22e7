#!/bin/bash
# Round 6, GPU call zi: SQ counters of the VGA frame's kernels (is k_splat_tile bound by its VALU work?)
TAG=${1:-r06zi}
CACHE=/tmp/mf_frames
REPO=$(pwd)
mkdir -p gpurun_out
timeout 300 python bench.py --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 0.5 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
B="--frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --no-variants --steps 300 --warmup 300"
for P in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  set -- $P; n=$1; shift
  rm -rf /tmp/prof_$n
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_$n -o p -- python $REPO/bench.py $B > /tmp/prof_$n.log 2>&1
  f=$(find /tmp/prof_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summary.py $f 100 > $REPO/gpurun_out/${TAG}_pmc_$n.csv || echo "counter pass $n failed"
done
grep -i "splat_tile\|bilateral_model\|clean_small_flags\|fuse_data" $REPO/gpurun_out/${TAG}_pmc_sq1.csv | cut -c1-160
