#!/bin/bash
# GPU box: per-slot phase times of one window-8 window, one dense window and a batch of 64 under two or more builds, alternating
export PYTHONPATH=$PWD
for rep in 1 2; do
  for lib in "$@"; do
    echo "$lib"
    SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_batch_phases.py 1 2>&1 | tail -1 | cut -c1-200
    SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_batch_phases.py 1 dense 2>&1 | tail -1 | cut -c1-200
    SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_batch_phases.py 64 2>&1 | tail -1 | cut -c1-200
  done
done
