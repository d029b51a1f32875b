"""The binding a maintainer of the reference would add to ``source/optical_flow.py`` to call libvof.so directly
(INTEGRATION.md section 2 prints this file verbatim; tests/test_abi_cpu.py checks the struct layout against the
library and tests/test_gpu_parity.py runs ``solve_stack`` on the reference-produced fixture G1).

``solve_stack`` replaces the per-pair loop OF.py:791-1186 and the epilogue OF.py:1189-1191.
"""
import ctypes as C
import os

import numpy as np

VOF_VERSION = 202                          # include/vof.h
LIB = os.environ.get("VOF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "opticalflow_amd",
                                                "csrc", "libvof.so")


class vof_params(C.Structure):             # include/vof.h: struct vof_params
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
                ("speed_alpha", C.c_double), ("remodelling_alpha", C.c_double), ("delta_x", C.c_double),
                ("delta_t", C.c_double), ("initial_v_x", C.c_double), ("initial_v_y", C.c_double),
                ("initial_remodelling", C.c_double), ("rtol", C.c_double), ("max_iterations", C.c_int32),
                ("nu_pre", C.c_int32), ("nu_post", C.c_int32), ("reference_quirks", C.c_int32),
                ("coarse_precision", C.c_int32), ("vcycle_precision", C.c_int32), ("nu_pre_coarse", C.c_int32),
                ("nu_post_coarse", C.c_int32), ("w_cycle_level", C.c_int32), ("w_cycle_visits", C.c_int32),
                ("krylov_method", C.c_int32), ("gmres_restart", C.c_int32), ("fallback_after", C.c_int32),
                ("warm_start_stride", C.c_int32),
                ("preconditioner", C.c_int32), ("reserved0", C.c_int32)]


class vof_pair_stats(C.Structure):         # include/vof.h: struct vof_pair_stats
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("relative_residual", C.c_double),
                ("L1_functional", C.c_double), ("speed_functional", C.c_double),
                ("remodelling_functional", C.c_double), ("batch_ms", C.c_double), ("batch_pairs", C.c_int32),
                ("reserved", C.c_int32)]


def load(path=LIB):
    lib = C.CDLL(path)
    lib.vof_version.restype = C.c_int
    lib.vof_params_size.restype = C.c_size_t
    lib.vof_default_params.restype = C.c_int
    lib.vof_default_params.argtypes = [C.POINTER(vof_params), C.c_size_t]
    lib.vof_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vof_solve_stack_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(vof_params)] + [C.c_void_p] * 5
    lib.vof_last_error.restype = C.c_char_p
    lib.vof_last_error.argtypes = [C.c_void_p]
    lib.vof_destroy.argtypes = [C.c_void_p]
    # ABI guard: refuse a library built from another include/vof.h instead of corrupting memory
    if lib.vof_version() != VOF_VERSION or lib.vof_params_size() != C.sizeof(vof_params):
        raise RuntimeError("libvof.so version %d / vof_params %d bytes does not match this binding (%d / %d)"
                           % (lib.vof_version(), lib.vof_params_size(), VOF_VERSION, C.sizeof(vof_params)))
    return lib


def solve_stack(movie_to_analyse, delta_x, delta_t, speed_alpha, remodelling_alpha,
                initial_v_x, initial_v_y, initial_remodelling, lib=None):         # replaces OF.py:791-1191
    lib = lib or load()
    T, N_i, N_j = movie_to_analyse.shape
    p = vof_params()
    if lib.vof_default_params(C.byref(p), C.sizeof(p)):                           # rtol 1e-6, max_it 1000 (OF.py:1120)
        raise RuntimeError("vof_params layout mismatch")
    p.speed_alpha, p.remodelling_alpha, p.delta_x, p.delta_t = speed_alpha, remodelling_alpha, delta_x, delta_t
    p.initial_v_x, p.initial_v_y, p.initial_remodelling = initial_v_x, initial_v_y, initial_remodelling
    ctx = C.c_void_p()
    if lib.vof_create(C.byref(ctx), 0, N_i, N_j, min(T - 1, 64), None):           # device 0, <= 64 pairs in flight
        raise RuntimeError(lib.vof_last_error(None).decode())
    movie = np.ascontiguousarray(movie_to_analyse, dtype=np.float64)
    out = [np.empty((T - 1, N_i, N_j)) for _ in range(4)]                         # v_x, v_y, remodelling, speed
    stats = (vof_pair_stats * (T - 1))()
    rc = lib.vof_solve_stack_host(ctx, movie.ctypes.data, T, C.byref(p), *[o.ctypes.data for o in out], stats)
    err = lib.vof_last_error(ctx).decode()
    lib.vof_destroy(ctx)
    if rc:
        raise RuntimeError(err)
    return out, stats   # result['converged'] = bool(stats[T - 2].converged)  (OF.py:1202: last pair)
