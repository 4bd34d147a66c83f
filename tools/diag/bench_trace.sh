#!/bin/bash
# Kernel timeline of one bench step (on the GPU box): tools/diag/bench_trace.sh [key]
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --no-cpu --no-ba --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
python3 tools/diag/trace_kernels.py gpurun_out/prof_tl ${1:-resize}
