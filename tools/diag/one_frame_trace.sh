export PYTHONPATH=$PWD TMPDIR=/tmp
rm -rf gpurun_out/prof_1f
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_1f -- python3 tools/diag/one_frame_trace.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_1f/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fast_cells" in r["Kernel_Name"]]
i0 = idx[-2]
# back up to the first resize of that call
j = i0
while j > 0 and "resize" in rows[j - 1]["Kernel_Name"] or "blur" in rows[j - 1]["Kernel_Name"]: j -= 1
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:j + 20]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s start %7.1f end %7.1f dur %6.1f us" % (r["Kernel_Name"].replace("void ", "")[:40], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
