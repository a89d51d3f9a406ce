#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE counter passes of tools/roofline_passes.sh.

    python tools/pmc_traffic.py <dir with pmc_fetch/ pmc_write/ trace/ kernels.json> <config@scale>

Per kernel class: mean FETCH_SIZE and WRITE_SIZE per dispatch (rocprofv3 reports them in KiB), the HBM bytes
derived from them, the algorithmic bytes and the mean duration from the kernel trace.  On gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes: a 1 GiB device-to-device copy launched in the same passes
(``--calibrate``) measures that factor, and the read side is multiplied by it.
"""
import collections
import csv
import glob
import json
import sys

CLASSES = (("spmm_csr_kernel", "spmm_csr"), ("spmm_tile_kernel", "spmm_csr"), ("::forward_kernel<", "fused_forward"),
           ("::backward_kernel<", "fused_backward"), ("::backward_pipe_kernel<", "fused_backward"),
           ("fused_forward_kernel", "fused_forward"),
           ("fused_backward_kernel", "fused_backward"), ("fused_reduce_kernel", "fused_reduce"),
           ("sum_blocks_kernel", "sum_blocks"))
COPY_BYTES = 1 << 30


def classify(name):
    for key, cls in CLASSES:
        if key in name:
            return cls
    return None


def mean_counter(path_glob, counter):
    files = glob.glob(path_glob, recursive=True)
    if not files:
        return {}, None
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        key = r["Dispatch_Id"]
        per_dispatch[key] += float(r["Counter_Value"])
        names[key] = r["Kernel_Name"]
    copy = []
    for key, v in per_dispatch.items():
        cls = classify(names[key])
        if cls:
            tot[cls] += v
            cnt[cls] += 1
        elif v * 1024 > 0.4 * COPY_BYTES and "elementwise" in names[key]:
            copy.append(v)
    return {c: tot[c] / cnt[c] for c in tot}, (sum(copy) / len(copy) if copy else None)


def trace_means(d):
    files = glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True)
    out = {}
    if files:
        for r in csv.DictReader(open(files[0])):
            cls = classify(r["Name"])
            if cls and cls not in out:
                out[cls] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])}
    return out


def main():
    d, key = sys.argv[1], sys.argv[2]
    fetch, copy_fetch = mean_counter(d + "/pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE")
    write, copy_write = mean_counter(d + "/pmc_write/**/*counter_collection.csv", "WRITE_SIZE")
    read_factor = (COPY_BYTES / (copy_fetch * 1024)) if copy_fetch else 2.0
    write_factor = (COPY_BYTES / (copy_write * 1024)) if copy_write else 1.0
    kernels = json.load(open(d + "/kernels.json"))["kernels"]
    trace = trace_means(d)
    out = {}
    for cls in sorted(set(fetch) | set(write)):
        f_kib, w_kib = fetch.get(cls, 0.0), write.get(cls, 0.0)
        rec = {"FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
               "hbm_bytes": int(f_kib * 1024 * read_factor + w_kib * 1024 * write_factor)}
        if cls in kernels:
            rec["algorithmic_bytes"] = kernels[cls]["bytes"]
            rec["event_us"] = kernels[cls]["us"]
        if cls in trace:
            rec["trace_avg_us"] = round(trace[cls]["avg_us"], 3)
        out[cls] = rec
    out["_calibration"] = {"copy_bytes": COPY_BYTES, "copy_FETCH_SIZE_KiB": copy_fetch, "copy_WRITE_SIZE_KiB": copy_write,
                           "read_factor": round(read_factor, 4), "write_factor": round(write_factor, 4)}
    print(json.dumps({key: out}, indent=1))


if __name__ == "__main__":
    main()
