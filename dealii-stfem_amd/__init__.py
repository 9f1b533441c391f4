"""dealii-stfem_amd: MI355X-native matrix-free space-time operator apply.

Python is only a thin ctypes veneer over the C-ABI in include/stfem.h (libstfem_hip.so, built
from csrc/ by `make -C dealii-stfem_amd/csrc`).  The classes keep the reference's names
(`MatrixFreeOperator`, `SystemMatrix`, include/operators.h:967-1191, 517-663) so tests and
bench.py read like the reference's callers.  There is NO CPU fallback: if the HIP library is
missing or no GPU is present, construction raises.

Import with  importlib.import_module("dealii-stfem_amd")  (the hyphen is the package name the
build contract prescribes).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstfem_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "stfem.h")

CGP, DG = 0, 1


class StfemError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().stfem_strerror(status).decode()
        hip = lib().stfem_last_hip_error().decode()
        super().__init__(f"{what}: {msg} ({status})" + (f" [{hip}]" if hip else ""))


def build(force=False):
    """Compile libstfem_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


class _MeshDesc(C.Structure):
    _fields_ = [("ncell", C.c_int32 * 3), ("vertices", C.POINTER(C.c_double)),
                ("lower", C.c_double * 3), ("upper", C.c_double * 3),
                ("dirichlet_mask", C.c_int32), ("device", C.c_int32)]


class _SpaceDesc(C.Structure):
    _fields_ = [("degree", C.c_int32), ("n_q_points_1d", C.c_int32), ("n_components", C.c_int32),
                ("precision", C.c_int32)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_lib = None

# every symbol include/stfem.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "stfem_ctx_create": (C.c_int, [C.POINTER(_MeshDesc), C.POINTER(_SpaceDesc), C.POINTER(_vp)]),
    "stfem_ctx_destroy": (None, [_vp]),
    "stfem_n_dofs": (C.c_int64, [_vp]),
    "stfem_n_cells": (C.c_int64, [_vp]),
    "stfem_is_cartesian": (C.c_int, [_vp]),
    "stfem_ctx_precision": (C.c_int, [_vp]),
    "stfem_set_coefficient": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "stfem_vector_create": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "stfem_vector_wrap": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "stfem_vector_rebind": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "stfem_vector_destroy": (None, [_vp]),
    "stfem_vector_n_blocks": (C.c_int, [_vp]),
    "stfem_vector_block": (_vp, [_vp, C.c_int]),
    "stfem_vector_upload": (C.c_int, [_vp, C.POINTER(_dp)]),
    "stfem_vector_download": (C.c_int, [_vp, C.POINTER(_dp)]),
    "stfem_st_vmult": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "stfem_space_vmult": (C.c_int, [_vp, C.c_double, C.c_double, _vp, _vp, _vp]),
    "stfem_diagonal": (C.c_int, [_vp, C.c_double, C.c_double, _vp, _vp]),
    "stfem_tensorproduct_add": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _vp, _vp, _vp]),
    "stfem_dot": (C.c_int, [_vp, _vp, _vp, C.c_int64, _dp, _vp]),
    "stfem_trace_push": (None, [C.c_char_p]),
    "stfem_trace_pop": (None, []),
    "stfem_multi_dot": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _vp, C.c_int64, _dp, _vp]),
    "stfem_multi_axpy": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(_vp), _vp, _vp]),
    "stfem_orthogonalize": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _vp, C.c_int64, _dp, _dp, _dp, _vp]),
    "stfem_diagonal_inverse": (C.c_int, [_vp, C.c_double, C.c_double, _vp, _vp]),
    "stfem_st_diagonal": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, _vp, _vp]),
    "stfem_plane_pack": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "stfem_plane_unpack": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "stfem_n_dofs_1d": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "stfem_comm_get_unique_id": (C.c_int, [_vp]),
    "stfem_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "stfem_comm_destroy": (None, [_vp]),
    "stfem_comm_available": (C.c_int, []),
    "stfem_comm_rccl_count": (C.c_int, [_vp]),
    "stfem_comm_rank": (C.c_int, [_vp]),
    "stfem_comm_size": (C.c_int, [_vp]),
    "stfem_comm_last_error": (C.c_char_p, []),
    "stfem_ghost_update": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "stfem_halo_begin": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "stfem_halo_end": (C.c_int, [_vp, _vp, _vp, _vp]),
    "stfem_halo_begin_split": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "stfem_planes_move": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "stfem_dot_global": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _dp, _vp]),
    "stfem_support_points": (C.c_int, [_vp, _dp]),
    "stfem_quadrature_points": (C.c_int, [_vp, C.c_int, _dp]),
    "stfem_integrate_rhs": (C.c_int, [_vp, C.c_int, _dp, _vp, C.c_int, _vp]),
    "stfem_integrate_difference": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _dp, _dp, _dp, _vp]),
    "stfem_integrate_rhs_product": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, _vp, C.c_int, _vp]),
    "stfem_integrate_difference_product": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _dp, _vp]),
    "stfem_vector_axpby": (C.c_int, [_vp, C.c_double, _vp, C.c_double, _vp, _vp]),
    "stfem_vector_set_zero": (C.c_int, [_vp, _vp, _vp]),
    "stfem_axpby_many": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64), C.c_double, C.POINTER(_vp), C.c_double, C.POINTER(_vp), _vp]),
    "stfem_driver_last_error": (C.c_char_p, []),
    "stfem_gauss_rule": (C.c_int, [C.c_int, _dp, _dp]),
    "stfem_fe_time_points": (C.c_int, [C.c_int, C.c_int, _dp]),
    "stfem_time_prolongation_matrix": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_int32)]),
    "stfem_time_restriction_matrix": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_int32)]),
    "stfem_time_projection_matrix": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_int32)]),
    "stfem_poly_mg_sequence": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stfem_mg_sequence": (C.c_int, [C.c_int] * 5 + [C.c_char] + [C.c_int] * 4 + [C.c_char_p, C.POINTER(C.c_int32)]),
    "stfem_precondition_stmg_types": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "stfem_transfer_create": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "stfem_transfer_create_partitioned": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp)]),
    "stfem_transfer_destroy": (None, [_vp]),
    "stfem_transfer_prolongate": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "stfem_transfer_restrict": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "stfem_transfer_interpolate": (C.c_int, [_vp, _vp, _vp, _vp]),
    "stfem_transfer_last_error": (C.c_char_p, []),
    "stfem_transfer_line_matrices": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "stfem_vector_convert": (C.c_int, [_vp, _vp, _vp]),
    "stfem_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "stfem_stream_destroy": (None, [_vp]),
    "stfem_stream_synchronize": (C.c_int, [_vp]),
    "stfem_graph_begin": (C.c_int, [_vp]),
    "stfem_graph_end": (C.c_int, [_vp, C.POINTER(_vp)]),
    "stfem_graph_launch": (C.c_int, [_vp, _vp]),
    "stfem_graph_destroy": (None, [_vp]),
    "stfem_vanka_create": (C.c_int, [_vp, C.c_int, _dp, _dp, C.POINTER(_vp)]),
    "stfem_vanka_create_partitioned": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, C.POINTER(_vp)]),
    "stfem_vanka_create_partitioned_general": (C.c_int, [_vp, _vp, C.c_int, _dp, _dp, C.c_int, C.POINTER(_vp)]),
    "stfem_vanka_destroy": (None, [_vp]),
    "stfem_vanka_n_classes": (C.c_int, [_vp]),
    "stfem_vanka_plan": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "stfem_vanka_vmult": (C.c_int, [_vp, _vp, _vp, _vp]),
    "stfem_vanka_step": (C.c_int, [_vp, _vp, C.c_double, C.c_int, _vp, _vp]),
    "stfem_vanka_last_error": (C.c_char_p, []),
    "stfem_fe_time_weights": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int, _dp, _dp, _dp, _dp]),
    "stfem_fe_time_weights_wave": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int,
                                             _dp, _dp, _dp, _dp, _dp]),
    "stfem_mesh_vertices": (C.c_int, [C.POINTER(C.c_int32), _dp, _dp, C.c_double, C.c_uint64,
                                      C.c_int32, C.c_int32, _dp]),
    "stfem_coefficient_per_cell": (C.c_int, [C.POINTER(C.c_int32), _dp, C.c_double, C.c_double,
                                             C.c_double, C.c_double, C.POINTER(C.c_int32), _dp,
                                             _dp, _dp]),
    "stfem_stokes_create": (C.c_int, [C.POINTER(_MeshDesc), C.c_int, C.c_double, C.POINTER(_vp)]),
    "stfem_stokes_create_ex": (C.c_int, [C.POINTER(_MeshDesc), C.c_int, C.c_int, C.c_double, C.POINTER(_vp)]),
    "stfem_stokes_destroy": (None, [_vp]),
    "stfem_stokes_n_velocity_dofs": (C.c_int64, [_vp]),
    "stfem_stokes_n_pressure_dofs": (C.c_int64, [_vp]),
    "stfem_stokes_vector_create": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "stfem_stokes_vector_destroy": (None, [_vp, _vp]),
    "stfem_stokes_vector_upload": (C.c_int, [_vp, C.c_int, _vp, _dp]),
    "stfem_stokes_vector_download": (C.c_int, [_vp, C.c_int, _vp, _dp]),
    "stfem_stokes_vmult": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "stfem_stokes_mass_vmult": (C.c_int, [_vp, _vp, _vp, _vp]),
    "stfem_stokes_st_vmult": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, C.POINTER(_vp),
                                        C.POINTER(_vp), _vp]),
    "stfem_stokes_st_vmult_slice_add": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, C.POINTER(_vp),
                                                  _vp, _vp, _vp]),
    "stfem_stokes_last_hip_error": (C.c_char_p, []),
    "stfem_stokes_set_weak_boundaries": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double]),
    "stfem_stokes_n_face_points": (C.c_int64, [_vp]),
    "stfem_stokes_face_points": (C.c_int, [_vp, _dp]),
    "stfem_stokes_nitsche_rhs": (C.c_int, [_vp, _dp, _vp, _vp, _vp]),
    "stfem_stokes_pressure_ctx": (C.c_int, [_vp, C.POINTER(_vp)]),
    "stfem_stokes_pressure_mean_vectors": (C.c_int, [_vp, _dp, _dp, _dp]),
    "stfem_stokes_pressure_quadrature_points": (C.c_int, [_vp, C.c_int, _dp]),
    "stfem_stokes_pressure_difference": (C.c_int, [_vp, C.c_int, _vp, _dp, _dp, _vp]),
    "stfem_stokes_dgp_prolongate": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "stfem_stokes_dgp_restrict": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "stfem_stokes_vanka_create": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int32), _dp, _dp, C.POINTER(_vp)]),
    "stfem_stokes_vanka_destroy": (None, [_vp]),
    "stfem_stokes_vanka_n_classes": (C.c_int, [_vp]),
    "stfem_stokes_vanka_vmult": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "stfem_stokes_vanka_step": (C.c_int, [_vp, C.POINTER(_vp), C.c_double, C.c_int, C.POINTER(_vp), _vp]),
    "stfem_stokes_vanka_last_error": (C.c_char_p, []),
    "stfem_strerror": (C.c_char_p, [C.c_int]),
    "stfem_last_hip_error": (C.c_char_p, []),
    "stfem_last_kernel_name": (C.c_char_p, [_vp]),
}


def lib():
    """Loads the HIP library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C dealii-stfem_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(os.environ.get("STFEM_LIB", LIB_PATH))  # STFEM_LIB: experiment builds only
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _check(status, what):
    if status != 0:
        raise StfemError(status, what)


# --------------------------------------------------------------------------- host helpers


def _time_transfer(fn, *args):
    dims = (C.c_int32 * 2)()
    _check(fn(*args, None, dims), fn.__name__)
    out = np.zeros((dims[0], dims[1]))
    _check(fn(*args, _p(out), dims), fn.__name__)
    return out


def get_time_prolongation_matrix(ttype, r, n_timesteps_at_once=2):
    """fe_time.h:805-849"""
    return _time_transfer(lib().stfem_time_prolongation_matrix, ttype, r, n_timesteps_at_once)


def get_time_restriction_matrix(ttype, r, n_timesteps_at_once=2):
    """fe_time.h:851-898"""
    return _time_transfer(lib().stfem_time_restriction_matrix, ttype, r, n_timesteps_at_once)


def get_time_projection_matrix(ttype, r_src, r_dst, n_timesteps_at_once=1):
    """fe_time.h:749-803"""
    return _time_transfer(lib().stfem_time_projection_matrix, ttype, r_src, r_dst, n_timesteps_at_once)


def get_fe_time_weights(ttype, r, time_step_size=1.0, n_timesteps_at_once=1):
    """fe_time.h:351-409 -> (Alpha, Beta, Gamma, Zeta)."""
    nb = (r if ttype == CGP else r + 1) * n_timesteps_at_once
    A = np.zeros((nb, nb)); B = np.zeros((nb, nb)); G = np.zeros((nb, 1)); Z = np.zeros((nb, 1))
    rc = lib().stfem_fe_time_weights(ttype, r, time_step_size, n_timesteps_at_once,
                                     _p(A), _p(B), _p(G), _p(Z))
    if rc != nb:
        raise StfemError(rc, "stfem_fe_time_weights")
    return A, B, G, Z


def get_fe_time_weights_wave(ttype, r, time_step_size=1.0, n_timesteps_at_once=1):
    """fe_time.h:157-305 -> (Alpha_lhs, Beta_lhs, rhs_uK, rhs_uM, rhs_vM)."""
    nb = (r if ttype == CGP else r + 1) * n_timesteps_at_once
    A = np.zeros((nb, nb)); B = np.zeros((nb, nb))
    v = [np.zeros((nb, 1)) for _ in range(3)]
    rc = lib().stfem_fe_time_weights_wave(ttype, r, time_step_size, n_timesteps_at_once,
                                          _p(A), _p(B), _p(v[0]), _p(v[1]), _p(v[2]))
    if rc != nb:
        raise StfemError(rc, "stfem_fe_time_weights_wave")
    return A, B, v[0], v[1], v[2]


def mesh_vertices(global_ncell, lower=(0, 0, 0), upper=(1, 1, 1), distort=0.0, seed=5489,
                  z_range=None):
    """Structured vertex grid of a z-slab [z0, z1) of cells of the global mesh."""
    gn = (C.c_int32 * 3)(*global_ncell)
    z0, z1 = z_range if z_range else (0, global_ncell[2])
    lo = np.array(lower, dtype=np.float64); up = np.array(upper, dtype=np.float64)
    out = np.zeros(((global_ncell[0] + 1) * (global_ncell[1] + 1) * (z1 - z0 + 1), 3))
    _check(lib().stfem_mesh_vertices(gn, _p(lo), _p(up), distort, seed, z0, z1, _p(out)),
           "stfem_mesh_vertices")
    return out


def coefficient_per_cell(ncell, vertices, c1=1.0, c2=9.0, c3=16.0, distort_coeff=0.0,
                         subdivisions=(1, 1, 1), lower=(0, 0, 0), upper=(1, 1, 1)):
    nc = (C.c_int32 * 3)(*ncell); sub = (C.c_int32 * 3)(*subdivisions)
    lo = np.array(lower, dtype=np.float64); up = np.array(upper, dtype=np.float64)
    v = np.ascontiguousarray(vertices, dtype=np.float64)
    out = np.zeros(int(np.prod(ncell)))
    _check(lib().stfem_coefficient_per_cell(nc, _p(v), c1, c2, c3, distort_coeff, sub, _p(lo),
                                            _p(up), _p(out)), "stfem_coefficient_per_cell")
    return out


# --------------------------------------------------------------------------- device objects


class BlockVector:
    """LinearAlgebra::distributed::BlockVector stand-in: n_blocks device arrays (types.h:19-23)."""

    def __init__(self, ctx, n_blocks=None, device_ptrs=None):
        self.ctx = ctx
        h = _vp()
        if device_ptrs is not None:
            arr = (_vp * len(device_ptrs))(*device_ptrs)
            _check(lib().stfem_vector_wrap(ctx._h, len(device_ptrs), arr, C.byref(h)),
                   "stfem_vector_wrap")
            self.n_blocks = len(device_ptrs)
        else:
            _check(lib().stfem_vector_create(ctx._h, n_blocks, C.byref(h)), "stfem_vector_create")
            self.n_blocks = n_blocks
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.stfem_vector_destroy(self._h)
            self._h = None

    def rebind(self, device_ptrs):
        """Points a view (made with device_ptrs=...) at other device arrays without reallocating."""
        arr = (_vp * len(device_ptrs))(*device_ptrs)
        _check(lib().stfem_vector_rebind(self._h, len(device_ptrs), arr), "stfem_vector_rebind")
        self.n_blocks = len(device_ptrs)
        return self

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        assert host.shape == (self.n_blocks, self.ctx.n_dofs)
        ptrs = (_dp * self.n_blocks)(*[_p(host[b]) for b in range(self.n_blocks)])
        _check(lib().stfem_vector_upload(self._h, ptrs), "stfem_vector_upload")
        return self

    def download(self):
        out = np.zeros((self.n_blocks, self.ctx.n_dofs))
        ptrs = (_dp * self.n_blocks)(*[_p(out[b]) for b in range(self.n_blocks)])
        _check(lib().stfem_vector_download(self._h, ptrs), "stfem_vector_download")
        return out

    def block_ptr(self, b):
        return lib().stfem_vector_block(self._h, b)


class MatrixFreeOperator:
    """Context = MatrixFree + MatrixFreeOperator state of one rank (operators.h:967-1191)."""

    def __init__(self, degree, ncell, vertices=None, lower=(0, 0, 0), upper=(1, 1, 1),
                 dirichlet_mask=63, device=0, mass_matrix_scaling=0.0, laplace_matrix_scaling=0.0,
                 number="double"):
        self.degree = degree
        self.ncell = tuple(int(v) for v in ncell)
        self.mass_matrix_scaling = mass_matrix_scaling
        self.laplace_matrix_scaling = laplace_matrix_scaling
        m = _MeshDesc()
        m.ncell[:] = self.ncell
        self._verts = None
        if vertices is not None:
            self._verts = np.ascontiguousarray(vertices, dtype=np.float64)
            assert self._verts.size == 3 * np.prod([n + 1 for n in self.ncell])
            m.vertices = _p(self._verts)
        m.lower[:] = lower
        m.upper[:] = upper
        m.dirichlet_mask = dirichlet_mask
        m.device = device
        assert number in ("double", "float")  # the operator's Number template argument
        self.number = number
        s = _SpaceDesc(degree, degree + 1, 1, 1 if number == "float" else 0)
        h = _vp()
        _check(lib().stfem_ctx_create(C.byref(m), C.byref(s), C.byref(h)), "stfem_ctx_create")
        self._h = h
        self.n_dofs = lib().stfem_n_dofs(h)
        self.n_cells = lib().stfem_n_cells(h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.stfem_ctx_destroy(self._h)
            self._h = None

    def m(self):
        return self.n_dofs

    @property
    def is_cartesian(self):
        return bool(lib().stfem_is_cartesian(self._h))

    def evaluate_coefficient(self, values, which=1):
        """operators.h:1060-1087.  values: None | (n_cells,) | (n_cells, nq^3)."""
        if values is None:
            _check(lib().stfem_set_coefficient(self._h, which, 0, None), "stfem_set_coefficient")
            return
        v = np.ascontiguousarray(values, dtype=np.float64)
        layout = 1 if v.size == self.n_cells else 2
        assert v.size in (self.n_cells, self.n_cells * (self.degree + 1) ** 3)
        _check(lib().stfem_set_coefficient(self._h, which, layout, _p(v)), "stfem_set_coefficient")

    def initialize_dof_vector(self, n_blocks=1):
        return BlockVector(self, n_blocks)

    def vmult(self, dst, src, stream=None):
        _check(lib().stfem_space_vmult(self._h, self.mass_matrix_scaling,
                                       self.laplace_matrix_scaling, dst._h, src._h, stream),
               "stfem_space_vmult")

    def compute_diagonal(self, mass=None, laplace=None, stream=None):
        d = BlockVector(self, 1)
        _check(lib().stfem_diagonal(self._h,
                                    self.mass_matrix_scaling if mass is None else mass,
                                    self.laplace_matrix_scaling if laplace is None else laplace,
                                    d._h, stream), "stfem_diagonal")
        return d

    def get_matrix_diagonal(self, stream=None):
        """operators.h:1035-1039"""
        return self.compute_diagonal(stream=stream)

    def get_matrix_diagonal_inverse(self, stream=None):
        """operators.h:1041-1045, 1106-1109: 1/d where |d| > sqrt(eps), 1 elsewhere."""
        d = BlockVector(self, 1)
        _check(lib().stfem_diagonal_inverse(self._h, self.mass_matrix_scaling, self.laplace_matrix_scaling,
                                            d._h, stream), "stfem_diagonal_inverse")
        return d

    @property
    def last_kernel_name(self):
        return lib().stfem_last_kernel_name(self._h).decode()


class PreconditionVanka:
    """stmg.h:619-907: cell-patch additive-Schwarz smoother of Alpha (x) K + Beta (x) M on one context."""

    def __init__(self, ctx, Alpha, Beta, neighbour_mask=0, extended=None):
        """neighbour_mask: faces (bits as dirichlet_mask) behind which another rank holds the next cells; extended: on general
        meshes the context of the slab plus one ghost cell layer per such face (stfem_vanka_create_partitioned_general)"""
        self.ctx = ctx
        A = np.ascontiguousarray(Alpha, dtype=np.float64)
        B = np.ascontiguousarray(Beta, dtype=np.float64)
        assert A.shape == B.shape and A.shape[0] == A.shape[1]
        h = _vp()
        if extended is not None:
            _check(lib().stfem_vanka_create_partitioned_general(ctx._h, extended._h, A.shape[0], _p(A), _p(B), neighbour_mask, C.byref(h)),
                   "stfem_vanka_create_partitioned_general")
        else:
            _check(lib().stfem_vanka_create_partitioned(ctx._h, A.shape[0], _p(A), _p(B), neighbour_mask, C.byref(h)),
                   "stfem_vanka_create")
        self._h, self.n_blocks = h, A.shape[0]

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.stfem_vanka_destroy(self._h)
            self._h = None

    @property
    def n_classes(self):
        return lib().stfem_vanka_n_classes(self._h)

    @property
    def plan(self):
        """(row tiles per workgroup, parts per cell block)"""
        out = (C.c_int32 * 2)()
        _check(lib().stfem_vanka_plan(self._h, out), "stfem_vanka_plan")
        return tuple(out)

    def vmult(self, dst, src, stream=None):
        _check(lib().stfem_vanka_vmult(self._h, dst._h, src._h, stream), "stfem_vanka_vmult")

    smooth = vmult  # stmg.h:881-885

    def step(self, dst, omega, accumulate, src, stream=None):
        """dst = (dst if accumulate else 0) + omega * vmult(src): the relaxation step around the smoother (stmg.h:1199-1238)"""
        _check(lib().stfem_vanka_step(self._h, dst._h, omega, int(bool(accumulate)), src._h, stream), "stfem_vanka_step")


class SystemMatrix:
    """operators.h:517-663: A = Alpha (x) K + Beta (x) M on one context."""

    def __init__(self, ctx, Alpha, Beta):
        self.ctx = ctx
        self.Alpha = np.ascontiguousarray(Alpha, dtype=np.float64)
        self.Beta = np.ascontiguousarray(Beta, dtype=np.float64)
        assert self.Alpha.shape == self.Beta.shape and self.Alpha.ndim == 2

    def m(self):
        return self.Alpha.shape[0] * self.ctx.n_dofs

    def initialize_dof_vector(self):
        return BlockVector(self.ctx, self.Alpha.shape[0])

    def _apply(self, dst, src, transpose, add, stream):
        nr, nc = self.Alpha.shape
        _check(lib().stfem_st_vmult(self.ctx._h, nr, nc, _p(self.Alpha), _p(self.Beta),
                                    int(transpose), int(add), dst._h, src._h, stream),
               "stfem_st_vmult")

    def vmult(self, dst, src, stream=None):
        self._apply(dst, src, False, False, stream)

    def Tvmult(self, dst, src, stream=None):
        self._apply(dst, src, True, False, stream)

    def vmult_slice(self, dst, src, stream=None):
        assert self.Alpha.shape[1] == 1
        self._apply(dst, src, False, False, stream)

    def vmult_slice_add(self, dst, src, stream=None):
        assert self.Alpha.shape[1] == 1
        self._apply(dst, src, False, True, stream)

    def _diagonal(self, inverse, stream):
        n = self.Alpha.shape[0]
        assert self.Alpha.shape == (n, n)
        d = BlockVector(self.ctx, n)
        _check(lib().stfem_st_diagonal(self.ctx._h, n, _p(self.Alpha), _p(self.Beta), int(inverse), d._h, stream),
               "stfem_st_diagonal")
        return d

    def get_matrix_diagonal(self, stream=None):
        """operators.h:613-623: block i = Alpha(i,i) diag K + Beta(i,i) diag M."""
        return self._diagonal(False, stream)

    def get_matrix_diagonal_inverse(self, stream=None):
        """operators.h:625-637, as the reference combines it: 1/Alpha(i,i) (diag K)^-1 + 1/Beta(i,i) (diag M)^-1."""
        return self._diagonal(True, stream)


def tensorproduct_add(ctx, c, A, b, stream=None):
    """operators.h:238-250: c_i += A(i,j) b_j, skipping exact zeros."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    _check(lib().stfem_tensorproduct_add(ctx._h, A.shape[0], A.shape[1], _p(A), c._h, b._h,
                                         stream), "stfem_tensorproduct_add")


def dot(ctx, a, b, n_own=0, stream=None):
    out = C.c_double(0.0)
    _check(lib().stfem_dot(ctx._h, a._h, b._h, n_own, C.byref(out), stream), "stfem_dot")
    return out.value


def multi_dot(ctx, vs, w, n_own=0, stream=None):
    """[<v_i, w>]: the inner products of a Gram-Schmidt step in one pass over w per eight vectors (deterministic reductions)"""
    k = len(vs)
    out = np.zeros(k)
    arr = (_vp * k)(*[v._h for v in vs])
    _check(lib().stfem_multi_dot(ctx._h, k, arr, w._h, n_own, _p(out), stream), "stfem_multi_dot")
    return out


def multi_axpy(ctx, coef, xs, y, stream=None):
    """y += sum_i coef_i x_i"""
    k = len(xs)
    c = np.ascontiguousarray(coef, dtype=np.float64)
    assert c.size == k
    arr = (_vp * k)(*[v._h for v in xs])
    _check(lib().stfem_multi_axpy(ctx._h, k, _p(c), arr, y._h, stream), "stfem_multi_axpy")


def orthogonalize(ctx, vs, w, n_own=0, stream=None):
    """one classical Gram-Schmidt pass on the device: returns (h = V^T w, <w, w> after w -= V h, <w, w> before)"""
    k = len(vs)
    h = np.zeros(k)
    n2, b2 = C.c_double(0.0), C.c_double(0.0)
    arr = (_vp * k)(*[v._h for v in vs])
    _check(lib().stfem_orthogonalize(ctx._h, k, arr, w._h, n_own, _p(h), C.byref(b2), C.byref(n2), stream), "stfem_orthogonalize")
    return h, n2.value, b2.value


# ------------------------------------------------------------------ space-time multigrid (8 f-2)

def get_poly_mg_sequence(k_max, k_min, sequence_type="decrease_by_one"):
    """fe_time.cc:40-56; coarsest degree first"""
    kind = {"bisect": 0, "decrease_by_one": 1, "go_to_one": 2}[sequence_type]
    n = C.c_int32(0)
    _check(lib().stfem_poly_mg_sequence(k_max, k_min, kind, None, C.byref(n)), "stfem_poly_mg_sequence")
    out = (C.c_int32 * n.value)()
    _check(lib().stfem_poly_mg_sequence(k_max, k_min, kind, out, C.byref(n)), "stfem_poly_mg_sequence")
    return list(out)


def get_mg_sequence(n_sp_lvl, k_seq, p_seq=(), n_timesteps_at_once=1, n_timesteps_at_once_min=1, lower_lvl="k",
                    coarsening_type="space_and_time", time_before_space=False, use_p_multigrid_space=False,
                    zip_from_back=True):
    """fe_time.cc:58-124 -> string over 't' (tau), 'k', 'h', 'p' (MGType), coarsest transfer first"""
    ct = {"space_or_time": 0, "space_and_time": 1}[coarsening_type]
    args = (n_sp_lvl, len(k_seq), len(p_seq), n_timesteps_at_once, n_timesteps_at_once_min, lower_lvl.encode(), ct,
            int(time_before_space), int(use_p_multigrid_space), int(zip_from_back))
    n = C.c_int32(0)
    _check(lib().stfem_mg_sequence(*args, None, C.byref(n)), "stfem_mg_sequence")
    buf = C.create_string_buffer(n.value + 1)
    _check(lib().stfem_mg_sequence(*args, buf, C.byref(n)), "stfem_mg_sequence")
    return buf.raw[:n.value].decode()


def get_precondition_stmg_types(mg_type_level, coarsening_type="space_and_time", time_before_space=False, smoother=1):
    """fe_time.cc:126-150: smoother id per level (0 = identity, 1 = relaxation, 2 = Chebyshev)"""
    ct = {"space_or_time": 0, "space_and_time": 1}[coarsening_type]
    out = (C.c_int32 * (len(mg_type_level) + 1))()
    _check(lib().stfem_precondition_stmg_types(mg_type_level.encode(), len(mg_type_level), ct, int(time_before_space),
                                               smoother, out), "stfem_precondition_stmg_types")
    return list(out)


class MGTwoLevelTransfer:
    """deal.II MGTwoLevelTransfer between two contexts as MGTwoLevelBlockTransfer uses it (stmg.h:38-110)."""

    def __init__(self, fine, coarse, neighbour_mask=0):
        """neighbour_mask: 16 = a slab below, 32 = a slab above (z-slab partition)"""
        self.fine, self.coarse = fine, coarse
        h = _vp()
        _check(lib().stfem_transfer_create_partitioned(fine._h, coarse._h, neighbour_mask, C.byref(h)), "stfem_transfer_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.stfem_transfer_destroy(self._h)
            self._h = None

    def prolongate_and_add(self, dst, src, stream=None):
        _check(lib().stfem_transfer_prolongate(self._h, dst._h, src._h, 1, stream), "stfem_transfer_prolongate")

    def prolongate(self, dst, src, stream=None):
        _check(lib().stfem_transfer_prolongate(self._h, dst._h, src._h, 0, stream), "stfem_transfer_prolongate")

    def restrict_and_add(self, dst, src, stream=None):
        _check(lib().stfem_transfer_restrict(self._h, dst._h, src._h, 1, stream), "stfem_transfer_restrict")

    def interpolate(self, dst, src, stream=None):
        _check(lib().stfem_transfer_interpolate(self._h, dst._h, src._h, stream), "stfem_transfer_interpolate")


def transfer_line_matrices(ncell_fine, degree_fine, ncell_coarse, degree_coarse):
    """(P [n_f x n_c], I [n_c x n_f]): the 1D factors of a space transfer, without constraints"""
    n_f, n_c = degree_fine * ncell_fine + 1, degree_coarse * ncell_coarse + 1
    P, I = np.zeros((n_f, n_c)), np.zeros((n_c, n_f))
    _check(lib().stfem_transfer_line_matrices(ncell_fine, degree_fine, ncell_coarse, degree_coarse, _p(P), _p(I)),
           "stfem_transfer_line_matrices")
    return P, I


def vector_convert(dst, src, stream=None):
    """dst = src across contexts of different precision (stmg.h:1330-1343)"""
    _check(lib().stfem_vector_convert(dst._h, src._h, stream), "stfem_vector_convert")


# ------------------------------------------------------------------------------- Stokes (8a-14)

def stokes_block_index(n_timedofs, timestep, variable, timedof, variable_major=True):
    """BlockSlice::index (reference include/fe_time.h:956-967), two variables."""
    if variable_major:
        return timestep * (2 * n_timedofs) + variable * n_timedofs + timedof
    return timestep * (2 * n_timedofs) + timedof * 2 + variable


def get_fe_time_weights_stokes(type_, r, time_step_size, n_timesteps_at_once=1):
    """Alpha, Beta, Gamma, Zeta of get_fe_time_weights_stokes (fe_time.h:1242-1285): the scalar matrices
    of get_fe_time_weights scattered into the (variable, time dof) block structure.  The
    pressure-pressure block of Alpha stays empty, Beta only couples velocity with velocity; the
    right-hand-side matrices Gamma, Zeta act on the velocity rows, and for cG Gamma on the pressure
    rows as well (fe_time.h:1275-1282)."""
    A, B, G, Z = get_fe_time_weights(type_, r, time_step_size, n_timesteps_at_once)
    n = A.shape[0]
    nt = n // n_timesteps_at_once
    Alpha = np.zeros((2 * n, 2 * n)); Beta = np.zeros((2 * n, 2 * n))
    Gamma = np.zeros((2 * n, G.shape[1])); Zeta = np.zeros((2 * n, Z.shape[1]))
    idx = lambda v, k: stokes_block_index(nt, k // nt, v, k % nt)  # noqa: E731
    for a in range(n):
        for b in range(n):
            for iv in range(2):
                for jv in range(2):
                    if not (iv == 1 and jv == 1):
                        Alpha[idx(iv, a), idx(jv, b)] = A[a, b]
            Beta[idx(0, a), idx(0, b)] = B[a, b]
        Gamma[idx(0, a), :] = G[a, :]
        Zeta[idx(0, a), :] = Z[a, :]
        if type_ == CGP:
            Gamma[idx(1, a), :] = G[a, :]
    return Alpha, Beta, Gamma, Zeta


class StokesMatrixFreeOperator:
    """StokesMatrixFreeOperator + SystemMatrixStokes of the reference (include/operators.h:1193-1575,
    666-868) for the cell loop, FE_Q(2)^3 x FE_Q(1).  Vectors are device pointers (e.g.
    torch.Tensor.data_ptr()): velocity 3 * n_velocity doubles (component-major), pressure n_pressure."""

    def __init__(self, ncell, vertices=None, lower=(0, 0, 0), upper=(1, 1, 1), dirichlet_mask=63,
                 viscosity=1.0, velocity_degree=2, device=0, weak_boundary_ids=(), outflow_boundary_ids=(),
                 penalty1=20.0, penalty2=10.0, dg_pressure=False):
        """weak_boundary_ids / outflow_boundary_ids: boundary ids 0..5 (face 2 d + s) as in the reference's constructor
        (operators.h:1206-1211); penalty1 / penalty2: its Nitsche penalties (gamma1 = viscosity penalty1, gamma2 = penalty2)."""
        m = _MeshDesc()
        self.ncell = tuple(int(v) for v in ncell)
        m.ncell[:] = self.ncell
        self._verts = None
        if vertices is not None:
            self._verts = np.ascontiguousarray(vertices, dtype=np.float64)
            m.vertices = _p(self._verts)
        m.lower[:] = lower
        m.upper[:] = upper
        m.dirichlet_mask = dirichlet_mask
        m.device = device
        h = _vp()
        # dg_pressure: FE_DGP(1) instead of FE_Q(1) (the reference's dGPressure, tests/tp_03stokes.cc:83-86)
        _check(lib().stfem_stokes_create_ex(C.byref(m), velocity_degree, 1 if dg_pressure else 0, viscosity, C.byref(h)),
               "stfem_stokes_create")
        self._h = h
        self.n_velocity = lib().stfem_stokes_n_velocity_dofs(h)
        self.n_pressure = lib().stfem_stokes_n_pressure_dofs(h)
        self.weak_mask = sum(1 << int(f) for f in set(weak_boundary_ids))
        self.outflow_mask = sum(1 << int(f) for f in set(outflow_boundary_ids))
        if self.weak_mask or self.outflow_mask:
            _check(lib().stfem_stokes_set_weak_boundaries(h, self.weak_mask, self.outflow_mask, penalty1, penalty2),
                   "stfem_stokes_set_weak_boundaries")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.stfem_stokes_destroy(self._h)
            self._h = None

    def face_points(self):
        """quadrature points of the weak faces [point][3], where the Dirichlet function is evaluated (operators.h:1911-1914)"""
        out = np.zeros((lib().stfem_stokes_n_face_points(self._h), 3))
        _check(lib().stfem_stokes_face_points(self._h, _p(out)), "stfem_stokes_face_points")
        return out

    def nitsche_rhs(self, g_at_face_points, dst_u, dst_p, stream=None):
        """StokesNitscheMatrixFreeOperator::vmult(dst) (operators.h:1833-1849): dst += boundary functional of the Dirichlet data"""
        g = np.ascontiguousarray(g_at_face_points, dtype=np.float64)
        assert g.shape == (lib().stfem_stokes_n_face_points(self._h), 3)
        _check(lib().stfem_stokes_nitsche_rhs(self._h, _p(g), getattr(dst_u, "ptr", dst_u), getattr(dst_p, "ptr", dst_p), stream),
               "stfem_stokes_nitsche_rhs")

    def initialize_dof_vector(self, variable, host=None):
        """Device vector of `variable` (0 velocity, 1 pressure) as a StokesVector; optionally filled."""
        return StokesVector(self, variable, host)

    def vmult(self, dst_u, dst_p, src_u, src_p, stream=None):
        dst_u, dst_p, src_u, src_p = (getattr(v, "ptr", v) for v in (dst_u, dst_p, src_u, src_p))
        _check(lib().stfem_stokes_vmult(self._h, dst_u, dst_p, src_u, src_p, stream), "stfem_stokes_vmult")

    def mass_vmult(self, dst_u, src_u, stream=None):
        dst_u, src_u = getattr(dst_u, "ptr", dst_u), getattr(src_u, "ptr", src_u)
        _check(lib().stfem_stokes_mass_vmult(self._h, dst_u, src_u, stream), "stfem_stokes_mass_vmult")

    def st_vmult(self, Alpha, Beta, n_timesteps_at_once, n_timedofs, dst_blocks, src_blocks,
                 variable_major=True, stream=None):
        """SystemMatrixStokes::vmult; dst_blocks / src_blocks: device pointers in BlockSlice order."""
        nb = 2 * n_timesteps_at_once * n_timedofs
        A = np.ascontiguousarray(Alpha, dtype=np.float64); B = np.ascontiguousarray(Beta, dtype=np.float64)
        assert A.shape == (nb, nb) and B.shape == (nb, nb) and len(dst_blocks) == nb and len(src_blocks) == nb
        d = (_vp * nb)(*[getattr(v, "ptr", v) for v in dst_blocks])
        s_ = (_vp * nb)(*[getattr(v, "ptr", v) for v in src_blocks])
        _check(lib().stfem_stokes_st_vmult(self._h, n_timesteps_at_once, n_timedofs, int(variable_major),
                                           _p(A), _p(B), d, s_, stream), "stfem_stokes_st_vmult")

    def st_Tvmult(self, Alpha, Beta, n_timesteps_at_once, n_timedofs, dst_blocks, src_blocks, variable_major=True, stream=None):
        """SystemMatrixStokes::Tvmult AS THE REFERENCE HAS IT (operators.h:708-745): its scatter overload (operators.h:111-123)
        takes j = index(it, v, id), i = index(jt, v, jd), so the result of source time dof (it, id) only goes to the destination
        blocks of the same time dof, weighted with the entries of row j summed over the time dofs - not a transpose.  Expressed
        with the effective matrices of that rule and run as one st_vmult."""
        nt, ns = n_timedofs, n_timesteps_at_once
        nb = 2 * nt * ns
        A = np.asarray(Alpha, dtype=np.float64); B = np.asarray(Beta, dtype=np.float64)
        eps10 = 10 * np.finfo(np.float64).eps
        Ae, Be = np.zeros((nb, nb)), np.zeros((nb, nb))
        idx = lambda it, v, d: stokes_block_index(nt, it, v, d, variable_major)  # noqa: E731
        for it in range(ns):
            for d in range(nt):
                col = idx(it, 0, d)  # the velocity column drives the scatters of st_vmult
                for v in range(2):
                    j = idx(it, v, d)
                    Ae[j, col] = sum(A[j, idx(jt, v, jd)] for jt in range(ns) for jd in range(nt) if abs(A[j, idx(jt, v, jd)]) > eps10)
                j = idx(it, 0, d)
                Be[j, col] = sum(B[j, idx(jt, 0, jd)] for jt in range(ns) for jd in range(nt) if abs(B[j, idx(jt, 0, jd)]) > eps10)
        self.st_vmult(Ae, Be, ns, nt, dst_blocks, src_blocks, variable_major, stream)

    def st_vmult_slice_add(self, Gamma, Zeta, n_timesteps_at_once, n_timedofs, dst_blocks, src_u, src_p,
                           variable_major=True, stream=None):
        """SystemMatrixStokes::vmult_slice_add (n x 1 right-hand-side case); dst is accumulated into."""
        nb = 2 * n_timesteps_at_once * n_timedofs
        g = np.ascontiguousarray(Gamma, dtype=np.float64).reshape(-1)
        z = np.ascontiguousarray(Zeta, dtype=np.float64).reshape(-1)
        assert g.size == nb and z.size == nb and len(dst_blocks) == nb
        d = (_vp * nb)(*[getattr(v, "ptr", v) for v in dst_blocks])
        _check(lib().stfem_stokes_st_vmult_slice_add(self._h, n_timesteps_at_once, n_timedofs, int(variable_major),
                                                     _p(g), _p(z), d, getattr(src_u, "ptr", src_u),
                                                     getattr(src_p, "ptr", src_p), stream),
               "stfem_stokes_st_vmult_slice_add")


class StokesPreconditionVanka:
    """PreconditionVanka over a two-variable BlockSlice (stmg.h:626-738, 832-872, as tests/tp_03stokes.cc:714-726 creates it).
    block_variable[i] = 0 (velocity) / 1 (pressure) for the blocks in BlockSlice order; Alpha, Beta: the matrices of
    get_fe_time_weights_stokes."""

    def __init__(self, op, block_variable, Alpha, Beta):
        self.op = op
        self.nb = len(block_variable)
        bv = (C.c_int32 * self.nb)(*[int(v) for v in block_variable])
        A = np.ascontiguousarray(Alpha, dtype=np.float64); B = np.ascontiguousarray(Beta, dtype=np.float64)
        assert A.shape == (self.nb, self.nb) and B.shape == (self.nb, self.nb)
        h = _vp()
        rc = lib().stfem_stokes_vanka_create(op._h, self.nb, bv, _p(A), _p(B), C.byref(h))
        if rc != 0:
            raise StfemError(rc, "stfem_stokes_vanka_create: " + lib().stfem_stokes_vanka_last_error().decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().stfem_stokes_vanka_destroy(self._h)
            self._h = None

    @property
    def n_classes(self):
        return lib().stfem_stokes_vanka_n_classes(self._h)

    def step(self, dst_blocks, omega, accumulate, src_blocks, stream=None):
        d = (_vp * self.nb)(*[getattr(v, "ptr", v) for v in dst_blocks])
        s_ = (_vp * self.nb)(*[getattr(v, "ptr", v) for v in src_blocks])
        rc = lib().stfem_stokes_vanka_step(self._h, d, omega, int(bool(accumulate)), s_, stream)
        if rc != 0:
            raise StfemError(rc, "stfem_stokes_vanka_step: " + lib().stfem_stokes_vanka_last_error().decode())

    def vmult(self, dst_blocks, src_blocks, stream=None):
        self.step(dst_blocks, 1.0, False, src_blocks, stream)

    smooth = vmult


class StokesVector:
    """One device vector of a StokesMatrixFreeOperator (velocity: 3 * n_velocity, pressure: n_pressure)."""

    def __init__(self, op, variable, host=None):
        self.op, self.variable = op, variable
        self.size = 3 * op.n_velocity if variable == 0 else op.n_pressure
        h = _vp()
        _check(lib().stfem_stokes_vector_create(op._h, variable, C.byref(h)), "stfem_stokes_vector_create")
        self.ptr = h.value
        if host is not None:
            self.upload(host)

    def upload(self, host):
        a = np.ascontiguousarray(host, dtype=np.float64).reshape(-1)
        assert a.size == self.size
        _check(lib().stfem_stokes_vector_upload(self.op._h, self.variable, self.ptr, _p(a)), "stfem_stokes_vector_upload")
        return self

    def download(self):
        out = np.zeros(self.size)
        _check(lib().stfem_stokes_vector_download(self.op._h, self.variable, self.ptr, _p(out)),
               "stfem_stokes_vector_download")
        return out

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None and getattr(self.op, "_h", None):
            _lib.stfem_stokes_vector_destroy(self.op._h, self.ptr)
            self.ptr = None


def axpby_many(ctx, a, xs, b, ys, stream=None):
    """y_i = a x_i + b y_i on device arrays of different lengths in one launch (stfem_axpby_many); xs / ys: objects with .ptr and
    .size (StokesVector) or (pointer, length) pairs"""
    def pl(v):
        return (v.ptr, v.size) if hasattr(v, "ptr") else (int(v[0]), int(v[1]))
    n = len(ys)
    py = [pl(v) for v in ys]
    px = [pl(v) for v in xs] if xs is not None else [(None, ln) for _, ln in py]
    assert len(px) == n and all(a_[1] == b_[1] for a_, b_ in zip(px, py))
    lens = (C.c_int64 * n)(*[ln for _, ln in py])
    X = (_vp * n)(*[p for p, _ in px])
    Y = (_vp * n)(*[p for p, _ in py])
    _check(lib().stfem_axpby_many(ctx._h, n, lens, float(a), X, float(b), Y, stream), "stfem_axpby_many")
