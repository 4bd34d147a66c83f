#!/bin/bash
# GPU box: per-kernel stats of the single-window local BA (window-8 and dense): tools/diag/ba_kstats.sh
export PYTHONPATH=$PWD TMPDIR=/tmp
for s in ba_w8 ba_dense; do
  rm -rf gpurun_out/prof_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$s -- python3 tools/diag/$s.py > gpurun_out/prof_$s.log 2>&1
  tail -1 gpurun_out/prof_$s.log
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$s/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]: print("  %-34s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
done
