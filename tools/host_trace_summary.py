"""Where the host-pointer path (mf_process_frame) loses against device-resident frames: timeline summary of one
`rocprofv3 --kernel-trace --memory-copy-trace` run of bench.py (the device-resident steps first, its host_input section last).

usage: host_trace_summary.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [n_host_frames] > summary.json

Frames are told apart by their first kernel (k_bilateral).  The last n_host_frames frames of the run are the host-input section, the
frames before the first host-to-device copy of a frame-sized buffer the device-resident one."""
import collections
import csv
import glob
import json
import os
import sys


def col(row, *names):
    for n in names:
        for k in row:
            if k.lower() == n.lower():
                return row[k]
    raise KeyError(names)


def load(pattern):
    files = sorted(glob.glob(os.path.join(sys.argv[1], "**", pattern), recursive=True))
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    n_host = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    kern = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name").split("(")[0].replace("void ", "").replace("mf::", ""),
             col(r, "Queue_Id")) for r in load("*kernel_trace.csv")]
    kern.sort()
    copies = []
    for r in load("*memory_copy_trace.csv"):
        try:
            nbytes = int(col(r, "Bytes", "Size", "Copy_Bytes"))
        except KeyError:
            nbytes = -1
        copies.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Direction"), nbytes))
    copies.sort()
    starts = [i for i, k in enumerate(kern) if k[2].startswith("k_bilateral")]
    if len(starts) < n_host + 40:
        raise SystemExit(f"only {len(starts)} frames in the trace")

    def section(first, last):     # frames first .. last-1 (indices into starts)
        per, busy, tail_gap, dur = [], [], [], collections.defaultdict(list)
        for f in range(first, last):
            a, b = starts[f], starts[f + 1]
            ks = kern[a:b]
            # the pose-log / copy kernels behind a frame belong to it; the period is bilateral to bilateral
            per.append(kern[b][0] - kern[a][0])
            busy.append(sum(e - s for s, e, _, _ in ks))
            tail_gap.append(kern[b][0] - max(e for s, e, _, _ in ks))
            for s, e, n, _ in ks:
                dur[n].append(e - s)
        med = lambda v: sorted(v)[len(v) // 2] / 1e3
        return {"frames": last - first, "period_us_median": med(per), "period_us_mean": sum(per) / len(per) / 1e3,
                "kernel_busy_us_mean": sum(busy) / len(busy) / 1e3, "gap_to_next_frame_us_median": med(tail_gap),
                "launches_per_frame": sum(len(v) for v in dur.values()) / (last - first),
                "kernel_us": {n: round(sum(v) / len(v) / 1e3, 2) for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]}}

    n = len(starts) - 1
    host = section(n - n_host, n)
    # device-resident frames: the n_host frames before the first big host-to-device copy that follows frame 40 (frames are uploaded once, up front)
    t_host0 = kern[starts[n - n_host - 14]][0]
    dev_last = max(i for i, s in enumerate(starts) if kern[s][0] < t_host0) - 2
    dev = section(max(dev_last - n_host, 20), dev_last)
    # the uploads of the host section
    t0, t1 = kern[starts[n - n_host]][0], kern[-1][1]
    ups = [c for c in copies if t0 <= c[0] <= t1 and "HOST_TO_DEVICE" in c[2].upper().replace(" ", "_")]
    under = collections.Counter()
    overlapped = 0
    ki = 0
    for s, e, _, _ in ups:
        names = set()
        for ks, ke, kn, _ in kern[starts[n - n_host]:]:
            if ks > e:
                break
            if ke > s and ks < e:
                names.add(kn)
        if names:
            overlapped += 1
        for kn in names:
            under[kn] += 1
    out = {"device_resident": dev, "host_input": host,
           "delta_us": {k: round(host["kernel_us"].get(k, 0) - dev["kernel_us"].get(k, 0), 2) for k in host["kernel_us"]},
           "uploads": {"count": len(ups), "per_frame": len(ups) / n_host,
                       "us_mean": (sum(e - s for s, e, _, _ in ups) / len(ups) / 1e3) if ups else None,
                       "bytes": sorted({c[3] for c in ups}),
                       "running_under_kernels_frac": overlapped / len(ups) if ups else None,
                       "kernels_they_ran_under": dict(under.most_common(8))},
           "copy_directions_seen": dict(collections.Counter(c[2] for c in copies))}
    # wall-clock of the host section as the timeline sees it, and where in a frame its upload starts / ends
    first, last = starts[n - n_host], starts[n]
    out["host_input"]["span_us_per_frame"] = (kern[last][0] - kern[first][0]) / n_host / 1e3
    rel = []
    fs = [kern[s][0] for s in starts]
    import bisect
    for s, e, _, _ in ups:
        i = bisect.bisect_right(fs, s) - 1
        rel.append(((s - fs[i]) / 1e3, (e - fs[i]) / 1e3))
    if rel:
        rel.sort()
        out["uploads"]["start_us_into_the_frame_it_runs_under"] = {"p10": rel[len(rel) // 10][0], "median": rel[len(rel) // 2][0], "p90": rel[9 * len(rel) // 10][0]}
    if len(sys.argv) > 3:     # merged timeline of six frames in the middle of the host section
        a, b = starts[n - n_host // 2], starts[n - n_host // 2 + 6]
        t_a, t_b = kern[a][0], kern[b][0]
        ev = [(s, e, nme, "q" + str(q)) for s, e, nme, q in kern[a:b]] + [(s, e, "UPLOAD", "copy") for s, e, _, _ in ups if t_a <= s < t_b]
        ev.sort()
        with open(sys.argv[3], "w") as f:
            f.write("start_us,end_us,dur_us,what,where\n")
            for s, e, nme, q in ev:
                f.write(f"{(s - t_a) / 1e3:.2f},{(e - t_a) / 1e3:.2f},{(e - s) / 1e3:.2f},{nme},{q}\n")
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
