#!/bin/bash
# GPU box: rocprofv3 kernel-trace of bench.py's extract+match step; prints the stats rows matching a substring.  tools/diag/trace_kernel.sh [substring] [bench args]
export PYTHONPATH=$PWD TMPDIR=/tmp
pat=${1:-hamming}; shift
rm -rf gpurun_out/ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace -- python3 bench.py --no-cpu --no-ba --no-extras --steps 6 --warmup 2 "$@" > /dev/null 2>&1
f=$(ls gpurun_out/ktrace/*/*kernel_stats.csv | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min_us", round(float(r["MinNs"]) / 1e3, 2))
PY
