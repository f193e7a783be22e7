#!/bin/bash
# Round 6, GPU call y: whole GPU suite on the tree with the fused preprocessing launch + the frame pyramid's interior path; same-box A/B against the
# tree before them (tools/ab/tree_old = 091fb7e) on configs[1]; bit-level trace of the multi-model switch scenario in both trees
TAG=${1:-r06y}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
REPO=$(pwd)
(cd tools/ab/tree_old && PYTHONPATH=. python ../../state_dump.py > $REPO/gpurun_out/${TAG}_trace_old.txt 2>&1)
PYTHONPATH=. python tools/state_dump.py > gpurun_out/${TAG}_trace_new.txt 2>&1
echo "trace lines differing between the trees: $(diff gpurun_out/${TAG}_trace_old.txt gpurun_out/${TAG}_trace_new.txt | grep -c '^<')"
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
B="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 1.0"
run() { # name, dir, extra args
  n=$1; d=$2; shift; shift
  (cd $d && timeout 300 python bench.py $B "$@" > $REPO/gpurun_out/${TAG}_$n.json 2> $REPO/gpurun_out/${TAG}_$n.err)
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
run new . 
run old tools/ab/tree_old
run new_unfused . --param fusedPreprocessLaunch=0
run new2 .
run old2 tools/ab/tree_old
run new_unfused2 . --param fusedPreprocessLaunch=0
