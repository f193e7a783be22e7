#!/bin/bash
# Round 5, GPU call g: (1) ticket-counter microbenchmark; (2) the three size forms of the fuse / clean passes + the held-register clean experiment:
# parity subset, configs[4] A/B over clean form x ticket stride; (3) VGA: preprocessing on a (CU-masked) side stream.
TAG=${1:-r05g}
mkdir -p gpurun_out
(cd tools/micro && [ -x ticket_lanes ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-result ticket_lanes.hip -o ticket_lanes)
timeout 120 tools/micro/ticket_lanes > gpurun_out/${TAG}_ticket_lanes.txt 2>&1
cat gpurun_out/${TAG}_ticket_lanes.txt
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_surfel_passes.py "tests/test_gpu_parity_long.py::test_config4_dense_maps" \
   -x -q -m gpu -n 6 --durations=4 > gpurun_out/${TAG}_pytest.log 2>&1
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
    print(f"c4 {sys.argv[1]:18s} {d['value']:7.1f} frames/s  bgFuseClean {st['mmBackgroundFuseClean']:.3f} objFuseClean {st['mmObjectFuseClean']:.3f} Run {st['Run']:.3f} reps {d['config'].get('repetitions')}")
except Exception as e:
    print("c4", sys.argv[1], "FAILED", e)
PY
}
c4 default
c4 stride64k --param cleanTicketStride=16384
c4 held --param cleanHeld=1
c4 held_stride4k --param cleanHeld=1 --param cleanTicketStride=1024
c4 held_stride64k --param cleanHeld=1 --param cleanTicketStride=16384
c4 held_64k_8lanes --param cleanHeld=1 --param cleanTicketStride=16384 --param cleanTicketLanes=8
vga() {
  n=$1; shift
  timeout 300 python bench.py --frame-cache /tmp/mf_frames --no-variants --no-host-input --no-cpu-baseline "$@" > gpurun_out/${TAG}_vga_$n.json 2> gpurun_out/${TAG}_vga_$n.err
  python - "$n" gpurun_out/${TAG}_vga_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(f"vga {sys.argv[1]:18s} {d['value']:7.1f} frames/s  {d['ms_per_step']*1e3:.1f} us")
except Exception as e:
    print("vga", sys.argv[1], "FAILED", e)
PY
}
vga default
vga overlap --param overlapPreprocessing=1
vga overlap_cu32 --param preCUs=32 --param overlapPreprocessing=1
vga overlap_cu64 --param preCUs=64 --param overlapPreprocessing=1
vga overlap_cu16 --param preCUs=16 --param overlapPreprocessing=1
vga default_again
