#!/bin/bash
# per-kernel average durations of a short bench run (timeout 300 rocprofv3 --kernel-trace --stats); prints name, calls, avg us and copies
# the stats csv to gpurun_out/<tag>_kernel_stats.csv.  usage: tools/kstats.sh <tag> [bench.py args]
REPO=$(pwd); TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $REPO/bench.py --steps 100 --warmup 60 --min-seconds 0 --frames 60 --gen-workers 1 --no-cpu-baseline --no-host-input --no-roofline "$@" > /tmp/prof_k.log 2>&1
mkdir -p $REPO/gpurun_out
F=$(find /tmp/prof_k -name '*kernel_stats.csv' | head -1)
cp "$F" $REPO/gpurun_out/${TAG}_kernel_stats.csv
python - "$F" <<'PY'
import csv,sys
tot=0
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name'].split('(')[0]; c=int(r['Calls']); a=float(r['AverageNs'])/1e3
    if c>=100: print(f"{n:34s} {c:6d} {a:8.2f} us  x{c/160:6.2f}/frame = {a*c/160:7.1f}"); tot+=a*c/160
print("sum/frame", round(tot,1))
PY
grep -o '"value": [0-9.]*' /tmp/prof_k.log | head -1
tail -3 /tmp/prof_k.log | cut -c1-300
