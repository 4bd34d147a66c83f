#!/bin/bash
# Diagnostic build: tools/diag/build_diag.sh <name> <file.hip> <-DMACRO ...>  ->  tools/diag/libdiag_<name>.so
# (the product objects for every other source are reused; run python -m weiner_slamit_v2_amd.build first)
set -e
cd "$(dirname "$0")/../.."
name=$1; file=$2; shift 2
C=weiner_slamit_v2_amd/csrc
extra=""
case $file in ba_kernels.hip|pose.hip) extra="-ffp-contract=fast";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Wno-unused-value $extra "$@" -c $C/$file -o /tmp/diag_$name.o
objs=""
for f in $C/*.hip; do b=$(basename $f); [ "$b" = "$file" ] && continue; objs="$objs ${f%.hip}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libdiag_$name.so /tmp/diag_$name.o $objs
echo tools/diag/libdiag_$name.so
