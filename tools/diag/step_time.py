"""Median per-kernel duration and step time over the steps of a rocprofv3 kernel trace of bench.py: step_time.py <dir>"""
import csv, glob, os, statistics, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0][:28]
starts = [i for i, r in enumerate(rows) if "resize" in r["Kernel_Name"] and (i == 0 or "resize" not in rows[i - 1]["Kernel_Name"])]
per, steps = {}, []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    if not any("hamming" in r["Kernel_Name"] for r in seg): continue
    steps.append((int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3)
    acc = {}
    for r in seg: acc[name(r)] = acc.get(name(r), 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in acc.items(): per.setdefault(k, []).append(v)
print("steps %d  median step %.1f us  min %.1f us" % (len(steps), statistics.median(steps), min(steps)))
print("  ".join("%s %.1f" % (k, statistics.median(v)) for k, v in per.items()))
