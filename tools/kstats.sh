#!/bin/bash
# per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats); prints name, calls, avg us
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > /tmp/prof_k.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_k/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0]; c=int(r['Calls']); a=float(r['AverageNs'])/1e3
    if c>=100: print(f"{n:28s} {c:5d} {a:8.2f} us  x{c/210:5.2f}/frame = {a*c/210:7.1f}"); tot+=a*c/210
print("sum/frame", round(tot,1))
PY
grep -o '"value": [0-9.]*' /tmp/prof_k.log | head -1
