#!/bin/bash
# GPU box: ORB parity tests + kernel timeline of one bench step + the bench line.  tools/diag/quick_bench.sh [pytest -k expr]
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests -x -q -m gpu -k "${1:-orb or bench}" 2>&1 | tail -2 || exit 1
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --no-cpu --no-ba --no-extras --steps 24 --warmup 2 > /dev/null 2>&1
python3 tools/diag/step_time.py gpurun_out/prof_tl
python3 bench.py --no-ba --no-cpu --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['match_ms_per_step'])"
