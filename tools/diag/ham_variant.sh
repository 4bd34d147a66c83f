#!/bin/bash
# GPU box: rebuild hamming.hip with extra -D flags, relink, time the matcher kernel.  tools/diag/ham_variant.sh "-DX=1 ..." [bench args]
flags=$1; shift
( cd weiner_slamit_v2_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form $flags -c hamming.hip -o hamming.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamit_hip.so *.o ) || exit 1
echo "== $flags"
tools/diag/trace_kernel.sh hamming "$@"
