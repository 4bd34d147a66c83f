#!/bin/bash
# GPU box: slabs per group of the floating-window Schur product (SLAMIT_BA_SF_CAP): one window-8 window end to end, and the phases of a batch of 64
export PYTHONPATH=$PWD
for cap in 2 3 4 6 8; do echo "single cap $cap: $(SLAMIT_BA_SF_CAP=$cap python3 tools/diag/ba_w8.py 2>&1 | tail -1)"; done
for cap in 4 6 8 16; do echo "batch cap $cap: $(SLAMIT_BA_SF_CAP=$cap python3 tools/diag/ba_batch_phases.py 64 2>&1 | tail -1 | cut -c1-200)"; done
