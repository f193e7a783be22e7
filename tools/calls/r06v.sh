#!/bin/bash
# Round 6, GPU call v: "fusedPreprocessLaunch" -- the depth filter and the model-side pyramid in one launch.  Whole GPU suite, then A/B on configs[1]
TAG=${1:-r06v}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
B="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 1.0"
run() { # name, extra args
  n=$1; shift
  timeout 300 python bench.py $B "$@" > gpurun_out/${TAG}_$n.json 2> gpurun_out/${TAG}_$n.err
  python - "$n" gpurun_out/${TAG}_$n.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    st = d["roofline"]["stage_ms"]
    print(f"{sys.argv[1]:28s} {d['value']:8.1f} frames/s  {d['ms_per_step']*1e3:7.1f} us   pre {st['Preprocess']*1e3:5.1f} odomInit {st['odomInit']*1e3:5.1f} odom {st['odom']*1e3:6.1f} idx {st['indexMap']*1e3:5.1f} fuse {1e3*(st['Fuse::Data']+st['Fuse::Update']):5.1f} clean {st['Fuse::Copy']*1e3:5.1f} predict {st['IndexMap::ACTIVE']*1e3:5.1f}")
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run fused1
run fused0 --param fusedPreprocessLaunch=0
run fused1b
run fused0b --param fusedPreprocessLaunch=0
run fused1c
run fused0c --param fusedPreprocessLaunch=0
