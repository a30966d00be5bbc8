#!/usr/bin/env python3
"""Fabric-side bytes per launch of each path kernel from a tools/roofs.sh run (TCC_EA0_RDREQ x 128 B read + write requests x 64 / 32 B:
the calibrated reading) -> profiles/hbm_traffic_<scene>_<h>p_<spp>spp.json, the file bench.py reports as
roofline.traffic_from_profiles.  usage: tools/roofs_traffic.py gpurun_out/roofs_<tag> <scene> <height> <spp>"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {"k_trace_closest": "trace_closest", "k_trace_shadow": "trace_shadow", "k_shade": "shade", "k_tail": "tail", "k_resolve": "resolve", "k_finalize": "finalize"}


def main():
    d, scene, height, spp = sys.argv[1:5]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    agg = collections.defaultdict(float)
    launches = collections.Counter()
    newest = {}  # per counter group: the latest run only (gpurun merges a call's files into a directory that may hold an earlier call's)
    for f in glob.glob(d + "/g*/*/*_counter_collection.csv"):
        g = f[len(d):].split(os.sep)[1]
        if g not in newest or os.path.getmtime(f) > os.path.getmtime(newest[g]):
            newest[g] = f
    for f in newest.values():
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "<true" in name.replace(" ", ""):
                    continue  # the counting build
                for k, v in KERNELS.items():
                    if k in name:
                        agg[(v, row["Counter_Name"])] += float(row["Counter_Value"])
                        if row["Counter_Name"] == "TCC_EA0_RDREQ_sum":
                            launches[v] += 1
    bpl = {}
    for v in KERNELS.values():
        n = launches[v]
        if not n:
            continue
        rd = agg[(v, "TCC_EA0_RDREQ_sum")] * 128
        w64 = agg[(v, "TCC_EA0_WRREQ_64B_sum")]
        wr = w64 * 64 + (agg[(v, "TCC_EA0_WRREQ_sum")] - w64) * 32
        bpl[v] = int((rd + wr) / n)
    out = os.path.join(root, "profiles", f"hbm_traffic_{scene}_{height}p_{spp}spp.json")
    json.dump({"workload": f"{scene} {height}p {spp}spp",
               "source": f"{os.path.basename(d)}: rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (separate passes of bench.py --steps 1 --warmup 1), reads = requests x 128 B (tools/roofs_traffic.py)",
               "launches_counted": dict(launches), "bytes_per_launch": bpl}, open(out, "w"), indent=1)
    print(out, json.dumps(bpl))


if __name__ == "__main__":
    main()
