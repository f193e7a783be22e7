#!/bin/bash
# Round 5, GPU call i: (1) same-box A/B of Model::clean's window taps: grouped (tree) against serial (tools/ab/libmaskfusion_amd_serial.so, built from the
# same tree with -DMF_CLEAN_SERIAL_TAPS); (2) SQ / TCC counter passes over the configs[4] frame.
TAG=${1:-r05i}
mkdir -p gpurun_out
REPO=$GRAFT_REPO_ROOT
c4() {  # name, params...
  n=$1; shift
  timeout 400 python bench.py --config 4 --frame-cache /tmp/mf_frames "$@" > gpurun_out/${TAG}_c4_$n.json 2> gpurun_out/${TAG}_c4_$n.err
  python - "$n" gpurun_out/${TAG}_c4_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    st = d['stage_ms']
    print(f"c4 {sys.argv[1]:18s} {d['value']:7.1f} frames/s  bgFuseClean {st['mmBackgroundFuseClean']:.3f} objFuseClean {st['mmObjectFuseClean']:.3f} Run {st['Run']:.3f} reps {d['config'].get('repetitions')}")
except Exception as e:
    print("c4", sys.argv[1], "FAILED", e)
PY
}
vga() {
  n=$1; shift
  timeout 300 python bench.py --frame-cache /tmp/mf_frames --no-variants --no-host-input --no-cpu-baseline "$@" > gpurun_out/${TAG}_vga_$n.json 2> gpurun_out/${TAG}_vga_$n.err
  python - "$n" gpurun_out/${TAG}_vga_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(f"vga {sys.argv[1]:18s} {d['value']:7.1f} frames/s  {d['ms_per_step']*1e3:.1f} us", {k: round(v * 1e3, 1) for k, v in d['roofline']['stage_ms'].items() if v})
except Exception as e:
    print("vga", sys.argv[1], "FAILED", e)
PY
}
cp maskfusion_amd/libmaskfusion_amd.so /tmp/lib_grouped.so
vga grouped
c4 grouped
cp tools/ab/libmaskfusion_amd_serial.so maskfusion_amd/libmaskfusion_amd.so
vga serial
c4 serial
cp /tmp/lib_grouped.so maskfusion_amd/libmaskfusion_amd.so
vga grouped_again
c4 grouped_again
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/gpurun_out/${TAG}_counters.txt 2>&1
C4="--config 4 --frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0 --steps 20"
pass() {  # name, counters...
  n=$1; shift
  rm -rf /tmp/prof_$n
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_$n -o p -- python $REPO/bench.py $C4 > /tmp/prof_$n.log 2>&1
  f=$(find /tmp/prof_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $REPO/tools/pmc_summary.py $f 30 > $REPO/gpurun_out/${TAG}_c4_pmc_$n.csv; echo "pass $n ok: $(wc -l < $REPO/gpurun_out/${TAG}_c4_pmc_$n.csv) rows"; else echo "pass $n FAILED"; tail -5 /tmp/prof_$n.log; fi
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
cd $REPO
python - gpurun_out/${TAG} <<'PY'
import csv, sys, collections, os
tag = sys.argv[1]
tab = collections.defaultdict(dict)
for n in ("sq1", "sq2", "tcc", "tcp"):
    p = f"{tag}_c4_pmc_{n}.csv"
    if not os.path.exists(p): continue
    for line in list(open(p))[1:]:
        name, counter, launches, mean, _tot = line.rstrip("\n").rsplit(",", 4)
        tab[name.replace("void ", "").replace("mf::", "")][counter] = float(mean)
want = ["k_clean", "k_obj_clean_small_flags", "k_obj_clean_small_compact", "k_obj_splat_scatter", "k_obj_global_scatter", "k_obj_index_scatter", "k_obj_index_scatter2",
        "k_splat_bin", "k_splat_tile<4>", "k_global_tile<4>", "k_index_scatter", "k_index_resolve", "k_fuse_data", "k_icp_iter<512, 3>"]
for k in want:
    if k in tab:
        print(k, {c: (round(v / 1e6, 3) if v > 1e5 else v) for c, v in tab[k].items()})
PY
