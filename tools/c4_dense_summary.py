"""Per-kernel durations of the DENSE phase of `bench.py --config 4` from a rocprofv3 --kernel-trace CSV: the launches of the lead-in (small
maps, spawn passes) are left out by keeping, per kernel, the launches whose duration is at least a third of that kernel's longest one -- in the
dense phase every N-sized kernel runs on 26.9 M / 3.4 M surfels, in the lead-in on 10^4..10^6.  Output: a CSV with the columns of rocprofv3's
kernel_stats (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev) so that bench.py's rocprof_rows reads it.
usage: c4_dense_summary.py kernel_trace.csv > profiles/<tag>_c4_kernel_stats.csv"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
for name, d in by.items():
    big = [x for x in d if x * 3 >= max(d)]
    out.append((name, len(big), sum(big), sum(big) / len(big), min(big), max(big), statistics.pstdev(big) if len(big) > 1 else 0.0, len(d)))
tot = sum(o[2] for o in out) or 1
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "CallsInTrace"])
for o in sorted(out, key=lambda o: -o[2]):
    w.writerow([o[0], o[1], o[2], round(o[3], 3), round(100.0 * o[2] / tot, 4), o[4], o[5], round(o[6], 3), o[7]])
