#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries committed under profiles/.

Recipe (run on the GPU box from the repo root; three SEPARATE profiler runs, counters never combined with
traces other than the kernel trace -- MI355X_MICROARCH.md, HBM / rocprofv3 section):

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --streams 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --streams 1
  python3 tools/summarise_profiles.py gpurun_out/p_stats gpurun_out/p_fetch gpurun_out/p_write gpurun_out/summary TAG

Writes <out>/<TAG>_kernel_stats.csv (the rocprofv3 --stats kernel table, verbatim), <out>/<TAG>_pmc_raw.json
(median counter value per kernel name) and <out>/traffic.json (bytes per launch: FETCH_SIZE x 1024 x 2 -- gfx950
tallies 128-byte requests as 64 -- plus WRITE_SIZE x 1024), keyed the way bench.py looks kernels up.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits


def counter_medians(d, counter):
    vals = {}
    for path in find(d, "counter_collection.csv"):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: (statistics.median(v), len(v)) for k, v in vals.items()}


SHORT = (("row_qe_kernel", "row_qe_kernel"), ("col_legs_kernel", "col_legs_kernel"), ("col_fwdlegs_kernel", "col_fwdlegs_kernel"),
         ("col_div_kernel", "col_div_kernel"),
         ("bin_kernel", "bin_kernel<power>"), ("row_fft_kernel<float, 0", "row_fft_kernel<R2C>"),
         ("col_fft_kernel", "col_fft_kernel"))


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--from-raw":       # re-derive traffic.json from a committed *_pmc_raw.json
        raw, out = json.load(open(sys.argv[2])), sys.argv[3]
    else:
        stats_dir, fetch_dir, write_dir, out, tag = sys.argv[1:6]
        os.makedirs(out, exist_ok=True)
        ks = find(stats_dir, "kernel_stats.csv")
        if ks:
            shutil.copy(ks[0], os.path.join(out, tag + "_kernel_stats.csv"))
        fetch = counter_medians(fetch_dir, "FETCH_SIZE")
        write = counter_medians(write_dir, "WRITE_SIZE")
        raw = {}
        for k in sorted(set(fetch) | set(write)):
            f, nf = fetch.get(k, (0.0, 0))
            w, nw = write.get(k, (0.0, 0))
            raw[k] = {"FETCH_SIZE": f, "n_FETCH_SIZE": nf, "WRITE_SIZE": w, "n_WRITE_SIZE": nw}
        json.dump(raw, open(os.path.join(out, tag + "_pmc_raw.json"), "w"), indent=1)
    detail = {k: {"fetch_bytes_corrected": v["FETCH_SIZE"] * 2048, "write_bytes": v["WRITE_SIZE"] * 1024,
                  "hbm_bytes": v["FETCH_SIZE"] * 2048 + v["WRITE_SIZE"] * 1024} for k, v in raw.items()}
    traffic = {"_how": __doc__.split("Writes")[0].strip(), "detail": detail}
    for needle, short in SHORT:
        # medians per kernel name: the steady-state launches of the timed loop dominate the sample
        hits = [v["hbm_bytes"] for k, v in detail.items() if ("oa::" + needle) in k and "<double" not in k and v["hbm_bytes"] > 1e6]
        if short == "bin_kernel<power>":
            hits = [v["hbm_bytes"] for k, v in detail.items() if "oa::bin_kernel<float, false, true>" in k]
        if hits:
            traffic[short] = max(hits) if short != "col_fft_kernel" else sum(hits) / len(hits)
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    for k in ("row_qe_kernel", "col_legs_kernel", "col_fwdlegs_kernel", "col_div_kernel", "bin_kernel<power>", "row_fft_kernel<R2C>", "col_fft_kernel"):
        print(k, traffic.get(k))


if __name__ == "__main__":
    main()
