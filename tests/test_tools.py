"""Measurement tooling that is not exercised by bench.py itself: known-answer checks, CPU only."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_trace_summary_on_a_fabricated_timeline(tmp_path):
    """tools/host_trace_summary.py (the timeline summary behind DESIGN.md section 5, host-pointer path): a fabricated rocprofv3 trace --
    240 device-resident frames of three kernels, then 160 host frames whose ICP kernel is 2 us longer, with a 5 us bubble and one 50 us
    upload per frame starting 20 us into the frame -- must come back with exactly those figures."""
    kernels = [("void mf::k_bilateral(float const*)", 16000), ("void mf::k_icp_iter<512, 3>(mf::IcpKArgs)", 8000), ("mf::k_clean_flags(mf::CleanArgs)", 24000)]
    t, krows, crows = 1_000_000, [], []
    for f in range(400):
        host = f >= 240
        if host:
            crows.append({"Kind": "MEMORY_COPY", "Direction": "MEMORY_COPY_HOST_TO_DEVICE", "Start_Timestamp": t + 20000, "End_Timestamp": t + 70000})
        for i, (name, dur) in enumerate(kernels):
            d = dur + (2000 if host and "icp" in name else 0)
            krows.append({"Kind": "KERNEL_DISPATCH", "Queue_Id": 1, "Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + d})
            t += d + (5000 if host and i == len(kernels) - 1 else 0)
    for name, rows in (("x_kernel_trace.csv", krows), ("x_memory_copy_trace.csv", crows)):
        with open(tmp_path / name, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0]))
            w.writeheader()
            w.writerows(rows)
    tl = tmp_path / "timeline.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_trace_summary.py"), str(tmp_path), "150", str(tl)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    dev, host = out["device_resident"], out["host_input"]
    assert dev["period_us_median"] == 48.0 and dev["gap_to_next_frame_us_median"] == 0.0 and dev["launches_per_frame"] == 3.0
    assert host["period_us_median"] == 55.0 and host["gap_to_next_frame_us_median"] == 5.0 and abs(host["kernel_busy_us_mean"] - 50.0) < 1e-9
    assert out["delta_us"]["k_icp_iter<512, 3>"] == 2.0 and out["delta_us"]["k_bilateral"] == 0.0
    up = out["uploads"]
    assert abs(up["per_frame"] - 1.0) < 0.01 and up["us_mean"] == 50.0 and up["running_under_kernels_frac"] == 1.0
    assert up["start_us_into_the_frame_it_runs_under"]["median"] == 20.0
    lines = tl.read_text().splitlines()
    assert lines[0] == "start_us,end_us,dur_us,what,where" and sum("UPLOAD" in ln for ln in lines) == 6


def test_bench_reads_a_rounds_final_profile_before_its_intermediate_ones(tmp_path, monkeypatch):
    """bench.py's roofline entries cite the newest committed summary: profiles/rNN_* is round NN's final one, rNN<letter>_* came before it
    (plain reverse alphabetical order put r05e_* ahead of r05_*), and the configs[4] tables are only read when asked for by name."""
    import csv
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()

    def table(name, us):
        with open(prof / name, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            w.writerow(["mf::k_clean(mf::CleanArgs)", 10, 10 * us * 1000, us * 1000, 1.0, 1, 1, 0])
            w.writerow(["void mf::k_icp_iter<512, 3>(mf::IcpKArgs)", 10, 10 * us * 1000, us * 1000, 1.0, 1, 1, 0])
    for name, us in (("r04_kernel_stats.csv", 4), ("r05e_kernel_stats.csv", 5), ("r05_kernel_stats.csv", 6), ("r05e_c4_kernel_stats.csv", 50), ("r05_c4_kernel_stats.csv", 60),
                     ("r05_c4_kernel_stats_raw.csv", 70)):
        table(name, us)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert [os.path.basename(p) for p in bench._newest_first([str(prof / n) for n in ("r04_x.csv", "r05_x.csv", "r05e_x.csv", "r05a_x.csv")])] == \
        ["r05_x.csv", "r05e_x.csv", "r05a_x.csv", "r04_x.csv"]
    rows = bench.rocprof_rows(["k_icp_iter<512, 3>"])
    assert rows["k_icp_iter<512, 3>"]["source"] == "profiles/r05_kernel_stats.csv" and rows["k_icp_iter<512, 3>"]["us"] == 6.0
    rows = bench.rocprof_rows(["k_clean"], pattern="r*_c4_kernel_stats.csv")
    assert rows["k_clean"]["source"] == "profiles/r05_c4_kernel_stats.csv" and rows["k_clean"]["us"] == 60.0


def test_c4_dense_summary_keeps_the_launches_on_the_full_maps(tmp_path):
    """tools/c4_dense_summary.py (the per-kernel table bench.py's roofline_kernels reads for configs[4]): from a fabricated kernel trace in which a
    kernel runs 5 times on a small lead-in map (10-30 us) and 4 times on the full one (~1.3 ms), only the four long launches are averaged; a
    kernel whose launches are all alike keeps them all."""
    trace = tmp_path / "kernel_trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp"])
        t = 1000
        for d in (10_000, 20_000, 30_000, 12_000, 15_000, 1_300_000, 1_310_000, 1_290_000, 1_300_000):
            w.writerow(["KERNEL_DISPATCH", 1, 1, 7, "mf::k_clean(mf::CleanArgs)", 0, t, t + d]); t += d + 5000
        for d in (7_000, 7_500, 8_000):
            w.writerow(["KERNEL_DISPATCH", 1, 1, 8, "void mf::k_icp_iter<256, 1>(mf::IcpKArgs)", 0, t, t + d]); t += d + 5000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c4_dense_summary.py"), str(trace)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = {row["Name"]: row for row in csv.DictReader(r.stdout.splitlines())}
    clean, icp = rows["mf::k_clean(mf::CleanArgs)"], rows["void mf::k_icp_iter<256, 1>(mf::IcpKArgs)"]
    assert int(clean["Calls"]) == 4 and int(clean["CallsInTrace"]) == 9 and abs(float(clean["AverageNs"]) - 1_300_000) < 1
    assert int(icp["Calls"]) == 3 and abs(float(icp["AverageNs"]) - 7_500) < 1
