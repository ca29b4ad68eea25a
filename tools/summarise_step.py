#!/usr/bin/env python3
"""Per-step summary of rocprofv3 output for bench.py's timed loop (one HIP stream).

One reconstruction ("step") of the headline pipeline is a fixed kernel sequence that ends with
``bin_final_kernel`` (whose last workgroup also updates the moments); this tool cuts the dispatch list of each profiler run at those markers, keeps the
steady-state steps (the last ones of the run) and reports, per position in the sequence, the median duration
(kernel trace), FETCH_SIZE and WRITE_SIZE (two separate --pmc runs).  HBM bytes = FETCH_SIZE x 1024 x 2
(gfx950 tallies the 128-byte requests of wide coalesced reads as 64 B: MI355X_MICROARCH.md, HBM section)
+ WRITE_SIZE x 1024.

  python3 tools/summarise_step.py <stats_dir> <fetch_dir> <write_dir> <out_dir> <TAG> [suffix]

writes <out>/<TAG>_kernel_stats<suffix>.csv (rocprofv3's own --stats table, verbatim), <out>/<TAG>_step<suffix>.json
(the per-position table) and <out>/traffic_<TAG><suffix>.json (what bench.py reads: bytes per launch under bench.py's
kernel names + bytes_per_recon).
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

MARK = ("bin_final_kernel", "col_div_sp_bin_kernel")     # last launch of a reconstruction (fused tail: the divergence kernel itself)


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def short(name):
    n = name.replace("void ", "")
    if n.startswith("oa::"):
        n = n[4:]
    i = n.find("(")
    return n[:i] if i > 0 else n


def steps_from(rows, key_time):
    """rows: list of dicts with Kernel_Name (+ payload), in dispatch order -> list of steps (lists of rows)."""
    rows = sorted(rows, key=key_time)
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if any(m in r["Kernel_Name"] for m in MARK):
            steps.append(cur)
            cur = []
    return steps


def steady(steps, keep=8):
    """the last `keep` steps of the dominant kernel-name sequence (most dispatches in total): the timed loop's full
    pipeline, not the per-stage timing calls bench.py makes after it"""
    if not steps:
        return []
    weight = {}
    for s in steps:
        sig = tuple(r["Kernel_Name"] for r in s)
        weight[sig] = weight.get(sig, 0) + len(sig)
    sig = max(weight, key=weight.get)
    good = [s for s in steps if tuple(r["Kernel_Name"] for r in s) == sig]
    return good[-keep:]


def trace_table(stats_dir):
    rows = []
    for path in find(stats_dir, "kernel_trace.csv"):
        with open(path, newline="") as f:
            rows += list(csv.DictReader(f))
    st = steady(steps_from(rows, lambda r: int(r["Start_Timestamp"])))
    if not st:
        return []
    out = []
    for i in range(len(st[0])):
        d = [(int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) / 1e3 for s in st]
        r = st[0][i]
        out.append({"kernel": short(r["Kernel_Name"]), "grid": [int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])],
                    "wg": int(r["Workgroup_Size_X"]), "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                    "median_us": statistics.median(d), "min_us": min(d), "n": len(d)})
    return out


def counter_table(pmc_dir, counter):
    rows = []
    for path in find(pmc_dir, "counter_collection.csv"):
        with open(path, newline="") as f:
            rows += [r for r in csv.DictReader(f) if r.get("Counter_Name") == counter]
    st = steady(steps_from(rows, lambda r: int(r["Dispatch_Id"])))
    if not st:
        return []
    return [{"kernel": short(st[0][i]["Kernel_Name"]), "median": statistics.median([float(s[i]["Counter_Value"]) for s in st])}
            for i in range(len(st[0]))]


def bench_keys(table):
    """Map the step's positions onto bench.py's per-kernel names (composites = sums): landmarks are the row R2C, the
    first column pass, the fused row stage and the divergence kernel; whatever sits between them belongs to the
    composite stage in front of the next landmark."""
    names = [r["kernel"] for r in table]

    def first(*prefixes):
        for i, n in enumerate(names):
            if any(n.startswith(p) for p in prefixes):
                return i
        return None
    keys = {}
    r0 = first("row_fft_kernel", "row_r2c_w64_kernel", "row_r2c_w64x2_kernel", "row_r2c_w64r_kernel", "row_r2c_rsplit_kernel", "row_r2c_rs4096_kernel", "row_r2c_rs4096w_kernel", "row_r2c_rs8192_kernel", "row_r2c_rs2048_kernel")
    c0 = first("col_fft_kernel")
    q = first("row_qe8_kernel", "row_qe_pair_kernel", "row_qe_kernel")
    d = first("col_div_kernel", "col_div_sp_kernel", "col_div_sp_bin_kernel")
    fb = first("col_fband_kernel")
    if r0 is not None:
        keys["row_fft_kernel<R2C>"] = [r0]
    if fb is not None and q is not None:                # R-split path: one column kernel between the row pass and the row stage
        keys["legs_cols"] = list(range(fb, q))
        c0 = None
    if c0 is not None and (q is None or c0 < q):
        keys["col_fft_kernel<fwd"] = [c0]
    if c0 is not None and q is not None and q > c0 + 1:
        keys["fwdlegs_cols"] = list(range(c0 + 1, q))
    if q is not None:
        keys["row_qe_kernel"] = [q]
    if q is not None and d is not None and d > q:
        keys["cols_div_bin" if "_bin_kernel" in names[d] else "cols_div"] = list(range(q + 1, d + 1))
    if d is not None and d + 1 < len(names):
        keys["bin_kernel<power>"] = list(range(d + 1, len(names)))
    return keys


def main():
    stats_dir, fetch_dir, write_dir, out, tag = sys.argv[1:6]
    suffix = sys.argv[6] if len(sys.argv) > 6 else ""
    os.makedirs(out, exist_ok=True)
    ks = find(stats_dir, "kernel_stats.csv")
    if ks:
        shutil.copy(ks[0], os.path.join(out, "%s_kernel_stats%s.csv" % (tag, suffix)))
    table = trace_table(stats_dir)
    fetch = counter_table(fetch_dir, "FETCH_SIZE")
    write = counter_table(write_dir, "WRITE_SIZE")
    for i, r in enumerate(table):
        f = fetch[i]["median"] if i < len(fetch) and fetch[i]["kernel"] == r["kernel"] else None
        w = write[i]["median"] if i < len(write) and write[i]["kernel"] == r["kernel"] else None
        r["FETCH_SIZE_KiB"], r["WRITE_SIZE_KiB"] = f, w
        r["hbm_bytes"] = (f * 2048.0 if f is not None else 0.0) + (w * 1024.0 if w is not None else 0.0) if (f is not None or w is not None) else None
        if r["hbm_bytes"]:
            r["hbm_GBs"] = r["hbm_bytes"] / (r["median_us"] * 1e-6) / 1e9
    json.dump({"_how": __doc__, "step": table, "step_kernel_sum_us": sum(r["median_us"] for r in table)},
              open(os.path.join(out, "%s_step%s.json" % (tag, suffix)), "w"), indent=1)
    traffic = {"_how": "tools/summarise_step.py: HBM bytes per launch = FETCH_SIZE x 2048 + WRITE_SIZE x 1024 (KiB counters, separate "
                       "--pmc runs, gfx950 read correction), medians over the steady-state steps of bench.py's timed loop",
               "bytes_per_recon": sum(r["hbm_bytes"] or 0.0 for r in table)}
    for k, ids in bench_keys(table).items():
        vals = [table[i]["hbm_bytes"] for i in ids]
        if all(v is not None for v in vals):
            traffic[k] = sum(vals)
        traffic.setdefault("_us", {})[k] = sum(table[i]["median_us"] for i in ids)
    json.dump(traffic, open(os.path.join(out, "traffic_%s%s.json" % (tag, suffix)), "w"), indent=1)
    for r in table:
        print("%-72s %9.1f us  %s" % (r["kernel"][:72], r["median_us"], ("%.1f MB %.2f TB/s" % (r["hbm_bytes"] / 1e6, r.get("hbm_GBs", 0) / 1e3)) if r["hbm_bytes"] else ""))
    print("step kernel sum %.1f us; bytes/recon %.1f MB" % (sum(r["median_us"] for r in table), traffic["bytes_per_recon"] / 1e6))


if __name__ == "__main__":
    main()
