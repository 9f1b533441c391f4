#!/bin/bash
# tools/build_pencil_exp.sh <name> [flags...]: builds dealii-stfem_amd/libstfem_<name>.so = the product
# library with the Q4 fp64 pencil kernels recompiled with the given flags (diagnostic builds:
# -DSTFEM_PENCIL_EXP=<bits>, -DSTFEM_PENCIL_TIMELINE, ...).  Run with STFEM_LIB=<that file>.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../dealii-stfem_amd/csrc"
mkdir -p build_exp/$NAME
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -DSTFEM_QUICK $@"
/opt/rocm/bin/hipcc $FL -DSTFEM_PENCIL_P=4 -c -o build_exp/$NAME/stfem_pencil_p4.o stfem_pencil.hip
OBJS=$(ls *.o | grep -v "^stfem_pencil_p4.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC -o ../libstfem_$NAME.so $OBJS build_exp/$NAME/stfem_pencil_p4.o
