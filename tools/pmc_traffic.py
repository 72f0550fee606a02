#!/usr/bin/env python3
"""Summarises rocprofv3 outputs into the small files kept under profiles/.

  kernel stats : <dir>/*kernel_stats.csv           -> profiles/kernel_stats_rNN.csv (fdgs kernels + top others)
  PMC passes   : <fetch dir>, <write dir> (*counter_collection.csv, one counter per pass as
                 /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass)
                 -> profiles/pmc_traffic_rNN.json : per stage, HBM bytes per launch.

Unit / gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB
(bytes = value * 1024); on gfx950 FETCH_SIZE counts 128-B read requests at 64 B, i.e. reports half of the
bytes of wide (16 B / lane) reads -> the read side is doubled.  WRITE_SIZE is used as reported.
Infinity-Cache hits are counted (they are fabric requests), so this is "bytes requested from the memory side".
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGE_OF = {
    "preprocess_fwd_kernel": "preprocess_fwd", "preprocess_bwd_kernel": "preprocess_bwd",
    "blend_fwd_kernel": "blend_fwd", "blend_bwd_kernel": "blend_bwd",
    "radix_hist_kernel": "radix_sort", "radix_scan_kernel": "radix_sort", "radix_scatter_kernel": "radix_sort",
    "tile_bin_lds_kernel<false>": "tile_count", "tile_bin_direct_kernel<false>": "tile_count", "tile_scan_kernel": "tile_scan",
    "tile_bin_lds_kernel<true>": "tile_scatter", "tile_bin_direct_kernel<true>": "tile_scatter", "tile_sort_kernel": "tile_sort",
    "ssim_fwd_kernel": "ssim_fwd", "ssim_bwd_kernel": "ssim_bwd", "sh_bwd_kernel": "sh_bwd", "adam_kernel": "adam", "adam_seg_kernel": "adam", "sh_flush_kernel": "sh_flush", "ssim_fused_kernel": "ssim_fused",
}


def stage_of(kernel_name):
    for k, v in STAGE_OF.items():
        if k in kernel_name:
            return v
    return None


def read_counter(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    tot, launches = defaultdict(float), defaultdict(set)
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                name = r.get("Kernel_Name", "")
                tot[name] += float(r["Counter_Value"])
                launches[name].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    return tot, {k: len(v) for k, v in launches.items()}


def main():
    if len(sys.argv) < 5:
        raise SystemExit("usage: pmc_traffic.py <stats dir> <fetch dir> <write dir> <out prefix e.g. profiles/xxx_r01>")
    stats_dir, fetch_dir, write_dir, prefix = sys.argv[1:5]
    # ---- kernel stats summary ----
    files = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)
    if files:
        rows = list(csv.DictReader(open(files[0])))
        keep = [r for r in rows if "fdgs::" in r["Name"]] + [r for r in rows if "fdgs::" not in r["Name"]][:12]
        with open(prefix.replace("pmc_traffic", "kernel_stats") + ".csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in keep:
                w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r.get("MinNs", ""), r.get("MaxNs", "")])
    # ---- PMC traffic ----
    fetch, nf = read_counter(fetch_dir, "FETCH_SIZE")
    write, nw = read_counter(write_dir, "WRITE_SIZE")
    out = {}
    per_kernel = {}
    for name in set(fetch) | set(write):
        st = stage_of(name)
        n = max(nf.get(name, 0), nw.get(name, 0), 1)
        f_kib, w_kib = fetch.get(name, 0.0) / max(nf.get(name, 1), 1), write.get(name, 0.0) / max(nw.get(name, 1), 1)
        per_kernel[name.split("(")[0]] = {"launches": n, "fetch_kib_per_launch": f_kib, "write_kib_per_launch": w_kib}
        if st is None:
            continue
        e = out.setdefault(st, {"fetch_kib_per_launch_raw": 0.0, "write_kib_per_launch": 0.0, "kernels": []})
        # a stage made of several kernels (radix sort = hist + scan + scatter per pass): sum kernel averages x launches ratio
        e["fetch_kib_per_launch_raw"] += f_kib
        e["write_kib_per_launch"] += w_kib
        e["kernels"].append(name.split("(")[0])
    for st, e in out.items():
        e["hbm_bytes_per_launch"] = int((2.0 * e["fetch_kib_per_launch_raw"] + e["write_kib_per_launch"]) * 1024)
        e["hbm_bytes_per_launch_uncorrected"] = int((e["fetch_kib_per_launch_raw"] + e["write_kib_per_launch"]) * 1024)
    out["_per_kernel"] = per_kernel
    out["_note"] = "bytes = KiB * 1024; read side doubled for gfx950 (FETCH_SIZE counts 128-B requests as 64 B); separate --pmc passes"
    with open(prefix + ".json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
