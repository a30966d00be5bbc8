#!/usr/bin/env python3
"""Turns the output of tools/prof.sh into the two files bench.py and the judge read:

  profiles/<tag>_pmc_hbm_bytes.csv   per-kernel sums of FETCH_SIZE / WRITE_SIZE (KiB), one row per kernel
  profiles/hbm_traffic_<scene>_<h>p_<spp>spp.json   HBM bytes per launch of each path kernel
                                                    (reads x2: gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)

usage: tools/pmc_summary.py gpurun_out/prof_<tag> <tag> <scene> <height> <spp>
The prof.sh PMC passes run `bench.py --steps 1 --warmup 1`: one counting render (COUNT kernels) and
two plain renders (warm-up + timed); per-launch figures are taken over the plain kernels only.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {"k_trace_closest": "trace_closest", "k_trace_shadow": "trace_shadow", "k_shade": "shade",
           "k_tail": "tail", "k_resolve": "resolve", "k_finalize": "finalize"}


def short(name):
    for k, v in KERNELS.items():
        if k in name:
            counting = "<true" in name.replace(" ", "") or "<(bool)1" in name.replace(" ", "")
            return v, counting
    return None, False


def collect(directory, counter):
    files = sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    rows = defaultdict(lambda: [0, 0.0])
    with open(files[-1], newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            e = rows[r["Kernel_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return rows


def main():
    d, tag, scene, height, spp = sys.argv[1:6]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fetch = collect(os.path.join(d, "fetch"), "FETCH_SIZE")
    write = collect(os.path.join(d, "write"), "WRITE_SIZE")
    out_csv = os.path.join(root, "profiles", f"{tag}_pmc_hbm_bytes.csv")
    with open(out_csv, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over: bench.py --steps 1 --warmup 1\n")
        f.write("# unit of sum_KiB: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced reads -> multiply reads by 2\n")
        f.write("pass,counter,kernel,launches,sum_KiB\n")
        for nm, rows in (("fetch", fetch), ("write", write)):
            for k in sorted(rows):
                f.write(f'{nm},{"FETCH_SIZE" if nm == "fetch" else "WRITE_SIZE"},"{k}",{rows[k][0]},{rows[k][1]:.1f}\n')
    per = defaultdict(lambda: {"launches": 0, "read_KiB": 0.0, "write_KiB": 0.0})
    for k, (n, s) in fetch.items():
        nm, counting = short(k)
        if nm and not counting:
            per[nm]["launches"] += n
            per[nm]["read_KiB"] += s
    for k, (n, s) in write.items():
        nm, counting = short(k)
        if nm and not counting:
            per[nm]["write_KiB"] += s
    bpl = {nm: int((2.0 * v["read_KiB"] + v["write_KiB"]) * 1024 / max(v["launches"], 1)) for nm, v in per.items()}
    out_json = os.path.join(root, "profiles", f"hbm_traffic_{scene}_{height}p_{spp}spp.json")
    json.dump({"workload": f"{scene} {height}p {spp}spp",
               "source": f"profiles/{tag}_pmc_hbm_bytes.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 gfx950 correction; tools/pmc_summary.py)",
               "launches_counted": {nm: v["launches"] for nm, v in per.items()},
               "bytes_per_launch": bpl}, open(out_json, "w"), indent=1)
    print(out_csv)
    print(out_json, json.dumps(bpl))


if __name__ == "__main__":
    main()
