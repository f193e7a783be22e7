"""Timeline analysis of a rocprofv3 --kernel-trace CSV: per-kernel mean duration, mean gap to the previous kernel on the
same queue, and how much of each frame the queues overlap.  usage: trace_gaps.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# skip warm-up half
rows = rows[len(rows) // 2:]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    prev_end = None
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0][:40]
        dur[n].append(e - s)
        if prev_end is not None: gap[n].append(s - prev_end)
        prev_end = e
    tot = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
    nfr = sum(1 for r in rs if "k_frame_advance" in r["Kernel_Name"] or "k_bilateral" in r["Kernel_Name"])
    print(f"queue {q}: {len(rs)} kernels, span {tot:.0f} us, ~{nfr} frames, {tot/max(nfr,1):.1f} us/frame")
    print(f"  {'kernel':40s} {'n/frame':>7s} {'dur us':>8s} {'gap us':>8s} {'sum/frame':>10s}")
    busy = 0; gaps = 0
    for n in sorted(dur, key=lambda k: -sum(dur[k])):
        d = sum(dur[n]) / len(dur[n]) / 1e3
        g = (sum(gap[n]) / len(gap[n]) / 1e3) if gap[n] else 0
        per = len(dur[n]) / max(nfr, 1)
        busy += sum(dur[n]); gaps += sum(gap[n])
        print(f"  {n:40s} {per:7.2f} {d:8.2f} {g:8.2f} {per*(d+g):10.1f}")
    print(f"  busy {busy/1e3/max(nfr,1):.1f} us/frame, gaps {gaps/1e3/max(nfr,1):.1f} us/frame")
