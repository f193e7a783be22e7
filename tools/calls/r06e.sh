#!/bin/bash
# Round 6, GPU call c: configs[4] with the objects TRACKED; thread-per-surfel sprite passes for dense object maps; where the in-place clean starts to pay
TAG=${1:-r06e}
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
run() {
  n=$1; shift
  timeout 400 python bench.py --config 4 --frame-cache $CACHE --min-seconds 1.0 "$@" > gpurun_out/${TAG}_c4_$n.json 2> gpurun_out/${TAG}_c4_$n.err
  python - "$n" gpurun_out/${TAG}_c4_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(f"c4 {sys.argv[1]:12s} {d['value']:7.1f} frames/s  {d['ms_per_step']:.3f} ms  tracked {d['tracked_models']} compactions {d['compactions']} runs {d['runs']['visible']:.0f}/{d['runs']['clean_in_place']:.0f}/{d['runs']['background_table']:.0f}")
    print("    passes:", {k: round(v['ms'], 3) for k, v in d['roofline_passes']['passes'].items()})
    print("    stages:", {k: round(v, 3) for k, v in d['stage_ms'].items() if v})
    print("    surfels", d['config']['surfels'], '->', d['config']['surfels_at_end'])
except Exception as e:
    print("c4", sys.argv[1], "FAILED", e); print(open(sys.argv[2].replace('.json', '.err')).read()[-1500:])
PY
}
run tracked
run static --static-objects
run tracked_big1M --param bigMapElements=1000000
run static_big1M --static-objects --param bigMapElements=1000000
timeout 300 python bench.py --config 4n --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline > gpurun_out/${TAG}_4n.json 2> gpurun_out/${TAG}_4n.err
timeout 300 python bench.py --config 4n --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline --param bigMapElements=1000000 > gpurun_out/${TAG}_4n_big1M.json 2> gpurun_out/${TAG}_4n_big1M.err
python - <<'PY'
import json
for n in ("4n", "4n_big1M"):
    try:
        d = json.load(open(f"gpurun_out/r06e_{n}.json")); print(n, d["value"], d["ms_per_step"], d.get("stage_ms"))
    except Exception as e:
        print(n, "FAILED", e)
PY
