#!/bin/bash
# GPU box: host phases (SLAMIT_BA_TIMING) of single-window solves, window-8 and dense
export PYTHONPATH=$PWD SLAMIT_BA_TIMING=1
python3 tools/diag/ba_w8.py 2>&1 | tail -3
python3 tools/diag/ba_dense.py 2>&1 | tail -3
