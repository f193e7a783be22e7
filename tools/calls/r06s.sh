#!/bin/bash
# Round 6, GPU call s: A/B on configs[1] -- the depth filter dispatched without a barrier bit (unorderedPreprocess), threads per tile workgroup
TAG=${1:-r06s}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
B="--frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --min-seconds 1.0"
run() { # name, extra args
  n=$1; shift
  timeout 300 python bench.py $B "$@" > gpurun_out/${TAG}_$n.json 2> gpurun_out/${TAG}_$n.err
  python - "$n" gpurun_out/${TAG}_$n.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    st = d["roofline"]["stage_ms"]
    print(f"{sys.argv[1]:28s} {d['value']:8.1f} frames/s  {d['ms_per_step']*1e3:7.1f} us   pre {st['Preprocess']*1e3:5.1f} odom {st['odom']*1e3:6.1f} idx {st['indexMap']*1e3:5.1f} fuse {1e3*(st['Fuse::Data']+st['Fuse::Update']):5.1f} clean {st['Fuse::Copy']*1e3:5.1f} predict {st['IndexMap::ACTIVE']*1e3:5.1f}")
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run base
run unordered --param unorderedPreprocess=1
run t256 --param tileThreads=256
run t320 --param tileThreads=320
run t384 --param tileThreads=384
run t384_unordered --param tileThreads=384 --param unorderedPreprocess=1
run t256_unordered --param tileThreads=256 --param unorderedPreprocess=1
run base2
# does the filter really overlap the prediction's tail?  kernel trace with timestamps
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_u
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_u -o u -- python $GRAFT_REPO_ROOT/bench.py --frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --no-variants --steps 200 --warmup 60 --param unorderedPreprocess=1 > /tmp/prof_u.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find /tmp/prof_u -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ov = []; gaps = []
for a, b in zip(rows, rows[1:]):
    if "k_splat_tile" in a["Kernel_Name"] and "k_bilateral" in b["Kernel_Name"]:
        ov.append((int(a["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3)
print("k_splat_tile end - next k_bilateral start (us; positive = overlap): n", len(ov), "mean", sum(ov) / max(1, len(ov)), "min", min(ov, default=0), "max", max(ov, default=0))
PY
