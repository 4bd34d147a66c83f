#!/bin/bash
# GPU box: rebuild orb_kernels.hip with extra -D flags, relink, print the bench's stage times.  tools/diag/orbk_variant.sh "-DX=1 ..." [bench args]
flags=$1; shift
( cd weiner_slamit_v2_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc $flags -c orb_kernels.hip -o orb_kernels.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamit_hip.so *.o ) || exit 1
echo "== $flags"
PYTHONPATH=$PWD python3 bench.py --no-ba --no-cpu --no-extras --steps 20 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['stage_ms_per_step'])"
