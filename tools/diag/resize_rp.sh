#!/bin/bash
# Resize stage time per rows-per-lane variant (on the GPU box): tools/diag/resize_rp.sh
export PYTHONPATH=$PWD TMPDIR=/tmp
for rp in 1 2 4; do
  rm -rf gpurun_out/prof_rp$rp
  SLAMIT_RESIZE_RP=$rp rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_rp$rp -- python3 bench.py --no-cpu --no-ba --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
  echo "== RP $rp"; python3 tools/diag/trace_kernels.py gpurun_out/prof_rp$rp resize | grep resize
done
