# where the blur forks (SLAMIT_BLUR_SPLIT): bash tools/diag/blur_split_ab.sh   (GPU box)
for rep in 1 2; do
for l in -1 0 1 2 3 7; do
  SLAMIT_BLUR_SPLIT=$l timeout -k 10 200 python3 bench.py --no-ba --no-cpu --no-extras --steps 40 --reps 3 > gpurun_out/o.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/o.json') if l.startswith('{')][-1])
print('split $l', d['ms_per_step'], round(d['value']), d['stage_ms_per_step'])"
done
done
