#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (one JSON object per kernel name).

    python3 tools/pmc_summary.py <counter_collection.csv, or a directory holding one> [substring filter ...] > summary.json

Counter values are summed over the rows of one dispatch (rocprofv3 writes one row per counter instance) and then
averaged over the dispatches of a kernel; the first dispatch of each kernel is dropped as warm-up when there are
at least three.  Units are the counters' own (SQ_* cycle counters count quad-cycles per the hardware guide,
GRBM_GUI_ACTIVE is summed over the 8 XCDs, FETCH_SIZE / WRITE_SIZE are KiB)."""
import collections
import csv
import glob
import json
import sys


def short(name):
    for drop in ("void ", "(anonymous namespace)::"):
        name = name.replace(drop, "")
    return name.split("(")[0].strip()


def main():
    root = sys.argv[1]
    filters = sys.argv[2:]
    files = [root] if root.endswith(".csv") else glob.glob(root + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + root)
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if filters and not any(s in k for s in filters):
                continue
            per[k][int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[k] = {"grid": r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size"), "vgpr": r.get("VGPR_Count"),
                       "lds": r.get("LDS_Block_Size")}
    out = {}
    for k, disp in per.items():
        ids = sorted(disp)
        if len(ids) >= 3:
            ids = ids[1:]
        agg = collections.defaultdict(float)
        for i in ids:
            for c, v in disp[i].items():
                agg[c] += v
        out[k] = {"dispatches": len(ids), **meta[k], "counters": {c: round(v / len(ids), 2) for c, v in sorted(agg.items())}}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
