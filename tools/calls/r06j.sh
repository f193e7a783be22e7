#!/bin/bash
# Round 6, GPU call j: same-box kernel traces of configs[1], the tree at the start of the round against HEAD
TAG=${1:-r06j}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
timeout 300 python bench.py --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
B="--frame-cache $CACHE --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline --no-variants --steps 600 --warmup 60"
for t in new old; do
  D=$REPO; [ $t = old ] && D=$REPO/tools/ab/tree_old
  rm -rf /tmp/prof_$t
  (cd $D && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -o s -- python bench.py $B > /tmp/prof_$t.log 2>&1)
  cp $(find /tmp/prof_$t -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_${t}_kernel_stats.csv
done
cd $REPO
python3 - <<'PY'
import csv
def load(p):
    return {r["Name"].split("(")[0].replace("void ",""): (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load("gpurun_out/r06j_old_kernel_stats.csv"), load("gpurun_out/r06j_new_kernel_stats.csv")
fa, fb = a["mf::k_bilateral"][0], b["mf::k_bilateral"][0]
rows = sorted(((b.get(k,(0,0))[1]/fb - a.get(k,(0,0))[1]/fa)/1e3, k, a.get(k,(0,0))[1]/fa/1e3, b.get(k,(0,0))[1]/fb/1e3) for k in set(a)|set(b))
for d, k, ta, tb in rows[::-1][:10] + rows[:5]: print(f"{d:+7.2f} us/frame {k[:70]:70s} {ta:7.2f} -> {tb:7.2f}")
print("total", sum(r[2] for r in rows), sum(r[3] for r in rows))
PY
