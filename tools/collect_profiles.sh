#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel stats + HBM traffic counters of bench.py.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Summaries land in gpurun_out/; copy what should be judged into profiles/.
set -e
export PYTHONPATH=$PWD TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-ba --no-cpu --steps 10 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --no-ba --no-cpu --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --no-ba --no-cpu --steps 3 --warmup 1 > $OUT/write.log 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
