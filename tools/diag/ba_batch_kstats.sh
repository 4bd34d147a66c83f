#!/bin/bash
# GPU box: per-kernel stats of a batch of 64 window-8 local-BA windows: tools/diag/ba_batch_kstats.sh [lib.so]
export PYTHONPATH=$PWD TMPDIR=/tmp
[ -n "$1" ] && export SLAMIT_LIB=$PWD/$1
rm -rf gpurun_out/prof_bab
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bab -- python3 tools/diag/ba_batch_phases.py 64 > gpurun_out/prof_bab.log 2>&1
tail -1 gpurun_out/prof_bab.log | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_bab/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]: print("  %-34s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
