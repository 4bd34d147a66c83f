#!/bin/bash
# Same-box A/B of two builds (GPU box): tools/diag/ab.sh libA.so libB.so  -- alternates A B A B, prints frames/s and stage times
export PYTHONPATH=$PWD TMPDIR=/tmp
for rep in 1 2; do
  for lib in "$@"; do
    SLAMIT_LIB=$PWD/$lib python3 bench.py --no-ba --no-cpu --no-extras --steps 40 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), d['ms_per_step'], d['stage_ms_per_step'], d['match_ms_per_step'])"
  done
done
