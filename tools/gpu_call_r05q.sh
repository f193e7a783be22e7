#!/bin/bash
# Round 5, GPU call q: column-major key image in the FIRST index pass too (its resolve transposes through the LDS; the packed object maps row-major in both
# passes): parity subset, then same-box A/B of "indexKeysColumnMajor" on configs[4] and at VGA, configs[4] kernel trace.
TAG=${1:-r05q}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 700 python -m pytest tests/test_gpu_switches.py tests/test_gpu_surfel_passes.py tests/test_gpu_glsl_passes.py tests/test_gpu_pipeline.py tests/test_gpu_multimodel.py tests/test_gpu_api.py \
   "tests/test_gpu_parity_long.py::test_config4_dense_maps" -x -q -m gpu -n 6 --durations=4 > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/${TAG}_pytest.log
tail -8 gpurun_out/${TAG}_pytest.log | cut -c1-200
c4() {  # name, params...
  n=$1; shift
  timeout 400 python bench.py --config 4 --frame-cache /tmp/mf_frames "$@" > gpurun_out/${TAG}_c4_$n.json 2> gpurun_out/${TAG}_c4_$n.err
  python - "$n" gpurun_out/${TAG}_c4_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    st = d['stage_ms']
    print(f"c4 {sys.argv[1]:18s} {d['value']:7.1f} frames/s  indexMap {st['indexMap']:.3f} bgFuseClean {st['mmBackgroundFuseClean']:.3f} objFuseClean {st['mmObjectFuseClean']:.3f} Run {st['Run']:.3f} reps {d['config'].get('repetitions')}")
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
vga colmajor
vga rowmajor --param indexKeysColumnMajor=0
vga colmajor_again
c4 colmajor
c4 rowmajor --param indexKeysColumnMajor=0
c4 colmajor_again
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o s -- python $REPO/bench.py --config 4 --frame-cache /tmp/mf_frames --gen-workers 1 --min-seconds 0 --steps 20 > /tmp/prof_c4.log 2>&1
python $REPO/tools/c4_dense_summary.py $(find /tmp/prof_c4 -name "*kernel_trace.csv" | head -1) > $REPO/gpurun_out/${TAG}_c4_kernel_stats.csv
python - $REPO/gpurun_out/${TAG}_c4_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:40]:
    if any(k in r['Name'] for k in ('index_scatter', 'index_resolve', 'k_clean(', 'fuse_data')):
        print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d}/{int(r['CallsInTrace']):5d} avg {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:9.1f}")
PY
