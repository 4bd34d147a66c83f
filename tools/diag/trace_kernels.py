"""Prints the kernel timeline (start, end, duration) of one step in the middle of the run from a rocprofv3 kernel_trace.csv."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "fast_cells"
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
mid = idx[len(idx) // 2]
nxt = idx[len(idx) // 2 + 1]
per = nxt - mid
lo = mid - 8
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + per + 4]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s start %8.1f  end %8.1f  dur %7.1f us  q%s" % (r["Kernel_Name"].replace("void ", "")[:44], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "")))
