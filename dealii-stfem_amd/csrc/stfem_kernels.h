// Kernel parameter blocks and launch entry points shared by the device translation units and
// stfem_capi.hip.  Everything precision-dependent lives in stfem_kernels_decl.h, instantiated
// for fp64 (namespace stfem::f64, the solver precision) and fp32 (stfem::f32, the precision of
// the reference's multigrid levels, tests/tp_01.cc:780,801-806).
#pragma once
#include <cstdint>

namespace stfem {
constexpr int MAX_BLOCKS = 8; // temporal blocks handled by one launch (larger systems are tiled)
constexpr int EO_N = 20;      // >= eo_size(6) = 18 (FE_Q(5))

} // namespace stfem

#define STFEM_REAL double
#define STFEM_NS f64
#include "stfem_kernels_decl.h"
#undef STFEM_REAL
#undef STFEM_NS
#define STFEM_REAL float
#define STFEM_NS f32
#include "stfem_kernels_decl.h"
#undef STFEM_REAL
#undef STFEM_NS

// the namespace a device translation unit implements (make passes -DSTFEM_F32 for the fp32 objects)
#ifdef STFEM_F32
#define STFEM_PREC f32
#else
#define STFEM_PREC f64
#endif
