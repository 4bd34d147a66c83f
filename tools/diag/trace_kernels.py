"""Prints start-to-start timeline of the last extract call from a rocprofv3 kernel_trace.csv."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the first pipeline kernel
key = sys.argv[2] if len(sys.argv) > 2 else "resize"
last = max(i for i, r in enumerate(rows) if key in r["Kernel_Name"] and (i == 0 or key not in rows[i - 1]["Kernel_Name"]))
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last - 30:last - 14]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s start %8.1f us  dur %7.1f us" % (r["Kernel_Name"][:40], (s - t0) / 1e3, (e - s) / 1e3))
