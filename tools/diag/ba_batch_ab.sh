#!/bin/bash
# GPU box: per-phase times of a batch of 64 window-8 windows (and one window) under two or more builds: tools/diag/ba_batch_ab.sh libA.so libB.so
export PYTHONPATH=$PWD
for rep in 1 2; do
  for lib in "$@"; do
    echo "$lib"; SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_batch_phases.py 64 2>&1 | tail -1 | cut -c1-260
  done
done
for lib in "$@"; do echo "$lib (1 window, dense batch 16)"; SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_batch_phases.py 1 2>&1 | tail -1 | cut -c1-260; SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_batch_phases.py 16 dense 2>&1 | tail -1 | cut -c1-260; done
