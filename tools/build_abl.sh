#!/bin/bash
# Builds dealii-stfem_amd/libstfem_abl.so: the product library with the Q4 fp64 tile kernels compiled
# with -DSTFEM_ABLATION (timing experiments via STFEM_EXP; results are wrong when it is nonzero).
# Extra flags for the two ablation objects: tools/build_abl.sh -DFOO ...
set -e
cd "$(dirname "$0")/../dealii-stfem_amd/csrc"
mkdir -p build_abl
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast ${ABL--DSTFEM_ABLATION} $@"
/opt/rocm/bin/hipcc $FL -DSTFEM_TILE_P=4 -c -o build_abl/stfem_tile_p4.o stfem_tile.hip &
/opt/rocm/bin/hipcc $FL -c -o build_abl/stfem_tile.o stfem_tile.hip &
wait
OBJS="host_tables.o stfem_kernels.o stfem_kernels_f32.o build_abl/stfem_tile.o stfem_tile_p1.o stfem_tile_p2.o stfem_tile_p3.o build_abl/stfem_tile_p4.o stfem_tile_f32.o stfem_tile_f32_p1.o stfem_tile_f32_p2.o stfem_tile_f32_p3.o stfem_tile_f32_p4.o stfem_capi.o stfem_stokes.o"
/opt/rocm/bin/hipcc -shared -fPIC -o ../libstfem_abl.so $OBJS
