"""Per-kernel averages of the rocprofv3 counter passes under gpurun_out/pmc_*: pmc_summary.py [kernel-substring]"""
import collections, csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:32]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if pat in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
