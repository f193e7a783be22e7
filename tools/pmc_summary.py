"""Per-kernel mean of a rocprofv3 --pmc counter_collection CSV.  usage: pmc_summary.py counter_collection.csv [tail]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # only the last `tail` launches of each kernel (steady state)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "?").split("(")[0]
    acc[name][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
print("kernel,counter,launches,mean,total")
for name in sorted(acc, key=lambda n: -sum(sum(v) for v in acc[n].values())):
    for cname, v in acc[name].items():
        if tail:
            v = v[-tail:]
        print(f"{name},{cname},{len(v)},{sum(v) / len(v):.3f},{sum(v):.3f}")
