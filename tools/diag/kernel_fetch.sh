# FETCH_SIZE per launch of every kernel of a bench step (GPU box): bash tools/diag/kernel_fetch.sh
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -rf gpurun_out/ff; mkdir -p gpurun_out/ff
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ff -- python3 bench.py --no-ba --no-cpu --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = max(glob.glob("gpurun_out/ff/**/*counter_collection.csv", recursive=True))
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if sum(v) / len(v) > 5000: print("%-30s %8.1f MB per launch (%d)" % (k, sum(v) / len(v) / 1024, len(v)))
PY
