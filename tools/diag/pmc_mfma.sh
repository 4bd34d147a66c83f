#!/bin/bash
# GPU box: matrix-core + issue counters over bench.py's extract+match step; prints per-kernel averages.  tools/diag/pmc_mfma.sh [kernel-substring]
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_a -- python3 bench.py --no-cpu --no-ba --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_b -- python3 bench.py --no-cpu --no-ba --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
python3 tools/diag/pmc_summary.py "${1:-hamming}"
