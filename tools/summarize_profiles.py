#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the MI355X box) into the committed
summaries under profiles/:  <tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_hbm_traffic.json.

    python tools/summarize_profiles.py r01 [frames_per_launch]

FETCH_SIZE / WRITE_SIZE come from separate --pmc passes (they do not fit one pass) and are averaged per launch and
kernel.  Units: rocprofv3 reports KB (bytes = value * 1024).  gfx950 caveat (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE halves wide 16-B/lane streaming reads; the ORB kernels read 1-4 B per lane, an uncalibrated width, so the
values are reported as measured and the comparison with the algorithmic byte count is indicative."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return {}
    acc = {}
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d = acc.setdefault(k, {})
        key = r.get("Dispatch_Id")
        d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    bench = os.path.join(src, "bench.json")
    if os.path.exists(bench):
        line = [l for l in open(bench).read().splitlines() if l.startswith("{")][-1]
        json.dump(json.loads(line), open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
    fetch = per_kernel(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(src, "write"), "WRITE_SIZE")
    skip = ("__amd_rocclr", "at::native", "Cijk", "ncclDevKernel")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(skip):
            continue
        kernels[k] = {"FETCH_SIZE_KB_per_launch": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB_per_launch": round(write.get(k, 0.0), 1)}
    if kernels:
        doc = {"_about": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --no-ba --no-cpu --steps 3 "
                         "--warmup 1` on MI355X, averaged per launch; %d VGA frames per launch. Units: KB as rocprofv3 reports them "
                         "(bytes = value*1024). gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide 16-B/lane "
                         "streaming reads by 2x; these kernels read 1-4 B per lane, an uncalibrated width, so the values are reported "
                         "as measured." % frames,
               "frames_per_launch": frames, "kernels": kernels}
        json.dump(doc, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
    print("profiles/%s_* written: %d kernels with counters, stats %s, bench %s" % (tag, len(kernels), bool(stats), os.path.exists(bench)))


if __name__ == "__main__":
    main()
