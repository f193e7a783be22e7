"""Builds profiles/<tag>_pmc.json (what bench.py's roofline.traffic reads) from the per-kernel PMC summaries of one profile round.

usage: make_pmc_json.py <tag> <bench_fetch.csv> <bench_write.csv> <calib_fetch.csv> <calib_write.csv> <tree> [output file name]
(default output profiles/<tag>_pmc.json = what bench.py reads for configs[1]; another workload's summary gets another name)

Inputs are tools/pmc_summary.py outputs (kernel,counter,launches,mean,total) of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
passes (no other trace domain than --kernel-trace).  FETCH_SIZE / WRITE_SIZE are reported in KiB per launch; the correction
factors come from tools/micro/fetch_calib (kernels that move exactly 256 MiB with 4 B and 16 B per lane) run in the same way on the
same box: corrected bytes = reported KiB * 1024 * factor.  The library's streaming kernels use 4 B/lane planar loads (ICP) or
16 B/lane (surfel streams); each kernel's record says which factor was applied.
"""
import csv
import json
import sys

CALIB_BYTES = 256 * 1024 * 1024
WIDE = ("k_fuse_update", "k_clean_compact", "k_index_scatter", "k_splat_bin", "k_clean_flags", "k_fuse_data", "k_index_resolve", "k_index_resolve_transposing", "k_index_resolve_packed", "k_obj_index_resolve_packed", "k_splat_tile",
        "k_model_pyramid", "k_clean", "k_clean_small_flags", "k_clean_small_compact", "k_cull", "k_run_table", "k_global_tile", "k_obj_clean",
        "k_obj_index_scatter", "k_obj_index_scatter2", "k_obj_index_resolve", "k_obj_splat_scatter", "k_obj_global_scatter")   # float4 surfel / map streams


def load(path):
    out = {}
    for line in list(open(path))[1:]:
        # kernel names may contain commas (template arguments): the four numeric / counter columns are taken from the right
        name, _counter, launches, mean, _total = line.rstrip("\n").rsplit(",", 4)
        name = name.replace("void ", "").replace("mf::", "")
        out.setdefault(name, []).append((int(launches), float(mean)))
    return out


def merged(table, prefix):
    """launch-weighted mean over the template instances of one kernel (k_icp_iter<512>, <256>)"""
    n = s = 0
    for name, rows in table.items():
        if name == prefix or name.startswith(prefix + "<"):
            for launches, mean in rows:
                n += launches
                s += launches * mean
    return (s / n, n) if n else (None, 0)


def main():
    tag, bf, bw, cf, cw, tree = sys.argv[1:7]
    out_name = sys.argv[7] if len(sys.argv) > 7 else f"profiles/{tag}_pmc.json"
    F, Wt, CF, CW = load(bf), load(bw), load(cf), load(cw)
    fac = {}
    for key, table in (("read4", CF), ("read16", CF), ("write4", CW), ("write16", CW)):
        mean, _ = merged(table, "calib_" + key)
        fac[key] = CALIB_BYTES / (mean * 1024.0) if mean else None
    kernels = {}
    names = sorted({n.split("<")[0] for n in list(F) + list(Wt)})
    for k in names:
        f, nf = merged(F, k)
        w, nw = merged(Wt, k)
        if f is None and w is None:
            continue
        wide = k in WIDE
        ff, wf = fac["read16" if wide else "read4"], fac["write16" if wide else "write4"]
        kernels[k] = {"fetch_kib_reported": f, "write_kib_reported": w, "launches": max(nf, nw),
                      "fetch_bytes": None if f is None or ff is None else f * 1024.0 * ff,
                      "write_bytes": None if w is None or wf is None else w * 1024.0 * wf,
                      "note": f"{tag}: FETCH_SIZE / WRITE_SIZE KiB per launch x 1024 x calibration factor ({'16' if wide else '4'} B/lane: read x{ff:.3f}, write x{wf:.3f})"}
    out = {"tag": tag, "tree": tree, "calibration": {"bytes_per_launch": CALIB_BYTES, "factors": fac,
                                                      "method": "tools/micro/fetch_calib under the same two rocprofv3 --pmc passes"},
           "kernels": kernels}
    json.dump(out, open(out_name, "w"), indent=1)
    print(json.dumps({"factors": fac, "k_icp_iter": kernels.get("k_icp_iter")}, indent=1))


if __name__ == "__main__":
    main()
