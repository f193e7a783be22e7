#!/bin/bash
# Measurement artefacts of round 6 (run through gpurun from the repo root; one call).
#   gpurun_out/<tag>_pytest.log                the whole -m gpu suite as the driver runs it (pytest.ini: 6 xdist workers), with its wall time
#   gpurun_out/<tag>_bench.json                bench.py as the driver runs it (--steps 20 --warmup 5), with its `variants`
#   gpurun_out/<tag>_bench_c4.json             bench.py --config 4: configs[4] as specified (dense maps)
#   gpurun_out/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats, configs[1]
#   gpurun_out/<tag>_c4_kernel_stats_raw.csv   the same for --config 4 (lead-in launches included)
#   gpurun_out/<tag>_c4_kernel_stats.csv       ... its dense phase only (tools/c4_dense_summary.py over the kernel trace): the per-kernel table DESIGN.md section 4 quotes
#   gpurun_out/<tag>_c4_pmc_fetch.csv / _c4_pmc_write.csv / _calib_*.csv / <tag>_c4_hbm.json   separate --pmc FETCH_SIZE / WRITE_SIZE passes over --config 4
#                                              (dense launches: the last 30 of each kernel) + calibration
#   gpurun_out/<tag>_pmc_fetch.csv / _pmc_write.csv / <tag>_pmc.json     the same two passes over configs[1] (what bench.py's roofline.traffic reads)
TAG=${1:-r06}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
(lscpu | head -20; nproc) > gpurun_out/${TAG}_gpu_box_host.txt
t0=$(date +%s)
timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest -m gpu rc=$? wall $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/${TAG}_pytest.log
(cd tools/micro && [ -x fetch_calib ] || timeout 120 hipcc --offload-arch=gfx950 -O3 -Wno-unused-result fetch_calib.hip -o fetch_calib)
timeout 500 python bench.py --steps 20 --warmup 5 --frame-cache $CACHE > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench default rc=$?"
timeout 500 python bench.py --config 4 --frame-cache $CACHE > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench c4 rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_c4 /tmp/prof_f /tmp/prof_w /tmp/prof_cf /tmp/prof_cw /tmp/prof_4f /tmp/prof_4w
B="--frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --no-variants --steps 600 --warmup 60"   # no fork under the profiler
# (the profiler passes of --config 4 run on ONE stream -- objectStream 0 --: the per-kernel rows are kernels on their own, as DESIGN.md's tables describe
# them; the timing lines above run the default, the object models' passes beside the background's)
C4="--config 4 --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --steps 20 --param objectStream=0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $REPO/bench.py $B > /tmp/prof_s.log 2>&1
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o s -- python $REPO/bench.py $C4 > /tmp/prof_c4.log 2>&1
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_c4_kernel_stats_raw.csv
python $REPO/tools/c4_dense_summary.py $(find /tmp/prof_c4 -name "*kernel_trace.csv" | head -1) > $REPO/gpurun_out/${TAG}_c4_kernel_stats.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_4f -o f -- python $REPO/bench.py $C4 > /tmp/prof_4f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_4w -o w -- python $REPO/bench.py $C4 > /tmp/prof_4w.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py $B > /tmp/prof_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $REPO/bench.py $B > /tmp/prof_w.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_cf -o f -- $REPO/tools/micro/fetch_calib > /tmp/prof_cf.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_cw -o w -- $REPO/tools/micro/fetch_calib > /tmp/prof_cw.log 2>&1
# SQ / TCC / TCP counters of the configs[4] frame (what the waves of each kernel do with their time, L2 hit rate): three more passes
for P in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
         "tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  set -- $P; n=$1; shift
  rm -rf /tmp/prof_$n
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_$n -o p -- python $REPO/bench.py $C4 > /tmp/prof_$n.log 2>&1
  f=$(find /tmp/prof_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summary.py $f 30 > $REPO/gpurun_out/${TAG}_c4_pmc_$n.csv || echo "counter pass $n failed"
done
python $REPO/tools/pmc_summary.py $(find /tmp/prof_4f -name "*counter_collection.csv" | head -1) 30 > $REPO/gpurun_out/${TAG}_c4_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_4w -name "*counter_collection.csv" | head -1) 30 > $REPO/gpurun_out/${TAG}_c4_pmc_write.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_pmc_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) 4000 > $REPO/gpurun_out/${TAG}_pmc_write.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cf -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_fetch.csv
python $REPO/tools/pmc_summary.py $(find /tmp/prof_cw -name "*counter_collection.csv" | head -1) > $REPO/gpurun_out/${TAG}_calib_write.csv
cd $REPO
TREE="$(git rev-parse --short HEAD 2>/dev/null || cat .tree_id 2>/dev/null)"
python tools/make_pmc_json.py ${TAG} gpurun_out/${TAG}_pmc_fetch.csv gpurun_out/${TAG}_pmc_write.csv gpurun_out/${TAG}_calib_fetch.csv gpurun_out/${TAG}_calib_write.csv "$TREE" gpurun_out/${TAG}_pmc.json
python tools/make_pmc_json.py ${TAG}_c4 gpurun_out/${TAG}_c4_pmc_fetch.csv gpurun_out/${TAG}_c4_pmc_write.csv gpurun_out/${TAG}_calib_fetch.csv gpurun_out/${TAG}_calib_write.csv "$TREE" gpurun_out/${TAG}_c4_hbm.json
tail -3 /tmp/prof_4f.log | cut -c1-200
cut -c1-300 gpurun_out/${TAG}_bench.json
cut -c1-300 gpurun_out/${TAG}_bench_c4.json
head -14 gpurun_out/${TAG}_c4_kernel_stats.csv | cut -c1-140
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-120
tail -20 gpurun_out/${TAG}_pytest.log | cut -c1-200
