#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel stats, HBM traffic counters, SQ issue counters of the FAST
# kernel and MFMA counters of the BA kernels.  Counters go in their own passes, with --kernel-trace only
# (MI355X_MICROARCH.md "rocprofv3 PMC slots"; FETCH_SIZE and WRITE_SIZE do not fit one pass).  The program stands directly
# behind `--`.  Raw output lands in gpurun_out/prof_<tag>/; tools/summarize_profiles.py <tag> turns it into profiles/<tag>_*.
set -e
export PYTHONPATH=$PWD TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-ba --no-cpu --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 10 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace720 -- $B --config 720p --steps 6 > $OUT/trace720.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 3 --warmup 1 > $OUT/write.log 2>&1
# instruction issue of the shipped FAST kernel (verdict r01 item 1): two passes of <= 8 SQ counters, both geometries
for cfg in vga 720p; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_a_$cfg -- $B --config $cfg --steps 3 --warmup 1 > $OUT/sq_a_$cfg.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b_$cfg -- $B --config $cfg --steps 3 --warmup 1 > $OUT/sq_b_$cfg.log 2>&1
done
# local BA: kernel stats and matrix-core counters, single windows and batches (tools/bench_ba.py runs both)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_trace -- python3 tools/bench_ba.py 3 > $OUT/ba_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/ba_mfma -- python3 tools/bench_ba.py 2 > $OUT/ba_mfma.log 2>&1 || true
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --config 720p --no-ba > $OUT/bench_720p.json 2> $OUT/bench_720p.err
python3 bench.py --config pipeline --no-cpu > $OUT/bench_pipeline.json 2> $OUT/bench_pipeline.err
tail -1 $OUT/bench.json | cut -c1-300
