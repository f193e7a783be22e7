#!/bin/bash
# Round 5, GPU call s: filtered depth + mask travel with the packed map (clean's mask-decay lookup in the packed, column-major order): parity subset incl. configs[4],
# same-box A/B against HEAD (tools/ab/libmaskfusion_amd_old.so), VGA and configs[4].
TAG=${1:-r05s}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 700 python -m pytest tests/test_gpu_surfel_passes.py tests/test_gpu_glsl_passes.py tests/test_gpu_pipeline.py tests/test_gpu_switches.py tests/test_gpu_multimodel.py tests/test_gpu_api.py tests/test_gpu_sharded.py \
   "tests/test_gpu_parity_long.py::test_config4_dense_maps" "tests/test_gpu_parity_long.py::test_s2_eight_objects_tracked_teacher_forced" -x -q -m gpu -n 6 --durations=4 > gpurun_out/${TAG}_pytest.log 2>&1
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
    print(f"c4 {sys.argv[1]:18s} {d['value']:7.1f} frames/s  objFuseClean {st['mmObjectFuseClean']:.3f} indexMap {st['indexMap']:.3f} bgFuseClean {st['mmBackgroundFuseClean']:.3f} Run {st['Run']:.3f} reps {d['config'].get('repetitions')}")
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
cp maskfusion_amd/libmaskfusion_amd.so /tmp/lib_new.so
vga new
c4 new
cp tools/ab/libmaskfusion_amd_old.so maskfusion_amd/libmaskfusion_amd.so
vga old
c4 old
cp /tmp/lib_new.so maskfusion_amd/libmaskfusion_amd.so
vga new_again
c4 new_again
