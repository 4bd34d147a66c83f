# FETCH_SIZE and time of the FAST launch (GPU box): bash tools/diag/fast_fetch.sh
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -rf gpurun_out/ff; mkdir -p gpurun_out/ff
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ff -- python3 bench.py --no-ba --no-cpu --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/ff/**/*counter_collection.csv", recursive=True))
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fast_cells" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("fast_cells FETCH_SIZE per launch: %.1f MB over %d launches" % (sum(v) / len(v) / 1024, len(v)))
PY
