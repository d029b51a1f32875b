"""ctypes binding of libvof.so (the C ABI declared in include/vof.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded this module raises, and
every solver call needs a real GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# VOF_LIB: alternative build of the library (A/B experiments); default is the in-tree libvof.so
LIB_PATH = os.environ.get("VOF_LIB") or os.path.join(HERE, "csrc", "libvof.so")

K_NAMES = ["rhs", "apply0", "gs0", "gs", "residual", "restrict", "prolong", "galerkin0", "galerkin",
           "coarse_setup", "coarse_solve", "vector", "reduce", "finalize", "functionals", "coarse_tail"]


class VofParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
                ("speed_alpha", C.c_double), ("remodelling_alpha", C.c_double), ("delta_x", C.c_double),
                ("delta_t", C.c_double), ("initial_v_x", C.c_double), ("initial_v_y", C.c_double),
                ("initial_remodelling", C.c_double), ("rtol", C.c_double), ("max_iterations", C.c_int32),
                ("nu_pre", C.c_int32), ("nu_post", C.c_int32), ("reference_quirks", C.c_int32),
                ("coarse_precision", C.c_int32), ("vcycle_precision", C.c_int32), ("nu_pre_coarse", C.c_int32), ("nu_post_coarse", C.c_int32),
                ("w_cycle_level", C.c_int32), ("w_cycle_visits", C.c_int32), ("krylov_method", C.c_int32),
                ("gmres_restart", C.c_int32), ("fallback_after", C.c_int32), ("warm_start_stride", C.c_int32),
                ("preconditioner", C.c_int32), ("reserved0", C.c_int32)]


class VofPairStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("relative_residual", C.c_double),
                ("L1_functional", C.c_double), ("speed_functional", C.c_double),
                ("remodelling_functional", C.c_double), ("batch_ms", C.c_double), ("batch_pairs", C.c_int32),
                ("reserved", C.c_int32)]


STATS_DTYPE = np.dtype([("iterations", np.int32), ("converged", np.int32), ("relative_residual", np.float64),
                        ("L1_functional", np.float64), ("speed_functional", np.float64),
                        ("remodelling_functional", np.float64), ("batch_ms", np.float64), ("batch_pairs", np.int32),
                        ("reserved", np.int32)])
assert STATS_DTYPE.itemsize == C.sizeof(VofPairStats)

VARIATION_DTYPE = np.dtype([("speed_mean", np.float64), ("speed_variance", np.float64), ("remodelling_mean", np.float64),
                            ("remodelling_variance", np.float64), ("L1_functional", np.float64),
                            ("speed_functional", np.float64), ("remodelling_functional", np.float64),
                            ("max_relative_residual", np.float64), ("converged_last", np.int32),
                            ("converged_all", np.int32), ("max_iterations_used", np.int32), ("reserved", np.int32)])
assert VARIATION_DTYPE.itemsize == 80     # sizeof(vof_variation_stats)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/vof.h declares
SIGNATURES = {
    "vof_version": (C.c_int, []),
    "vof_params_size": (C.c_size_t, []),
    "vof_default_params": (C.c_int, [C.POINTER(VofParams), C.c_size_t]),
    "vof_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "vof_destroy": (None, [_vp]),
    "vof_last_error": (C.c_char_p, [_vp]),
    "vof_workspace_bytes": (C.c_size_t, [_vp]),
    "vof_query_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vof_query_workspace_for": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vof_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "vof_num_levels": (C.c_int, [_vp]),
    "vof_solve_stack_host": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(VofParams), _vp, _vp, _vp, _vp, _vp]),
    "vof_solve_stack_dev": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(VofParams), _vp, _vp, _vp, _vp, _vp]),
    "vof_texture_stack_dev": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_double, C.c_double]),
    "vof_bench_sweeps_dev": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(VofParams), C.c_int]),
    "vof_blur_stack_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "vof_blur_stack_host": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "vof_vary_regularisation_host": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(VofParams), _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp]),
    "vof_field_moments_dev": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vof_subsample_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "vof_profile_enable": (C.c_int, [_vp, C.c_int]),
    "vof_profile_reset": (C.c_int, [_vp]),
    "vof_profile_filter": (C.c_int, [_vp, C.c_int, C.c_int]),
    "vof_profile_get": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "vof_profile_get_units": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vof_profile_get_bytes": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "vof_profile_get_moved": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "vof_kernel_name": (C.c_char_p, [C.c_int]),
    "vof_debug_setup": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(VofParams)]),
    "vof_debug_check_canaries": (C.c_int, [_vp]),
    "vof_debug_level_shape": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vof_debug_rhs": (C.c_int, [_vp, _vp]),
    "vof_debug_apply": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "vof_debug_gs": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int]),
    "vof_debug_sweep": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int]),
    "vof_debug_smooth": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "vof_set_fused_sweeps": (C.c_int, [_vp, C.c_int]),
    "vof_debug_restrict": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "vof_debug_resrestrict_u": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "vof_debug_prolong_add": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "vof_debug_stencil": (C.c_int, [_vp, C.c_int, _vp]),
    "vof_debug_vcycle": (C.c_int, [_vp, _vp, _vp]),
    "vof_debug_coarse_solve": (C.c_int, [_vp, _vp, _vp]),
    "vof_debug_vcycle_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int)]),
}

_lib = None


class VofError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  If
    libvof.so pulled in the system runtime first and torch were imported later, the process would hold two runtimes and
    the second one finds no GPU.  Load torch's copy first (without importing torch) so that libvof.so binds to it and the
    process has ONE runtime whichever is used first.  No torch installed: the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)


def load_library(path: str | None = None):
    """Load libvof.so and declare all prototypes.  Raises if the library is absent (build it with
    ``python -m opticalflow_amd.build`` or ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise VofError(f"native library {p} not found: build it with `python -m opticalflow_amd.build` "
                       "(hipcc, gfx950). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def default_params(**overrides) -> VofParams:
    lib = load_library()
    p = VofParams()
    if lib.vof_default_params(C.byref(p), C.sizeof(p)) != 0:
        raise VofError(f"vof_params layout mismatch: binding {C.sizeof(p)} bytes, library {lib.vof_params_size()} bytes")
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown solver parameter {k!r}")
        setattr(p, k, v)
    return p


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):          # torch tensor (device memory)
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


class Solver:
    """One context = one device; owns the device workspace for (n_i, n_j) images and a batch of
    ``max_pairs_in_flight`` frame pairs."""

    def __init__(self, n_i: int, n_j: int, max_pairs_in_flight: int, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        self.n_i, self.n_j, self.max_pairs = int(n_i), int(n_j), int(max_pairs_in_flight)
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.vof_create(C.byref(h), self.device, self.n_i, self.n_j, self.max_pairs,
                                 C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise VofError("vof_create failed: " + self.lib.vof_last_error(None).decode())
        self.h = h

    # -- lifetime
    def check_canaries(self):
        """VOF_DEBUG_CANARY=1: raises VofError naming the buffer if a kernel wrote outside one (no-op otherwise)."""
        self._check(self.lib.vof_debug_check_canaries(self.h), "vof_debug_check_canaries")

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, None
            rc = self.lib.vof_debug_check_canaries(h)
            msg = self.lib.vof_last_error(h).decode() if rc != 0 else ""
            self.lib.vof_destroy(h)
            if rc != 0:
                raise VofError("device buffer guard regions damaged: " + msg)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise VofError(f"{what} failed ({rc}): " + self.lib.vof_last_error(self.h).decode())

    @property
    def workspace_bytes(self):
        return int(self.lib.vof_workspace_bytes(self.h))

    @property
    def num_levels(self):
        return int(self.lib.vof_num_levels(self.h))

    def level_shape(self, level):
        a, b = C.c_int(), C.c_int()
        self._check(self.lib.vof_debug_level_shape(self.h, level, C.byref(a), C.byref(b)), "level_shape")
        return a.value, b.value

    # -- solves
    def solve_host(self, movie: np.ndarray, params: VofParams, want_speed=True):
        movie = np.ascontiguousarray(movie, dtype=np.float64)
        T = movie.shape[0]
        assert movie.shape[1:] == (self.n_i, self.n_j)
        out = [np.empty((T - 1, self.n_i, self.n_j)) for _ in range(4 if want_speed else 3)]
        stats = np.zeros(T - 1, dtype=STATS_DTYPE)
        rc = self.lib.vof_solve_stack_host(self.h, _ptr(movie), T, C.byref(params), _ptr(out[0]), _ptr(out[1]),
                                           _ptr(out[2]), _ptr(out[3]) if want_speed else None, _ptr(stats))
        self._check(rc, "vof_solve_stack_host")
        return (*out, stats) if want_speed else (*out, None, stats)

    def solve_dev(self, movie, n_frames, params: VofParams, v_x, v_y, remodelling, speed=None, stats=True):
        """All arrays are device memory (torch tensors or raw pointers)."""
        st = np.zeros(n_frames - 1, dtype=STATS_DTYPE) if stats else None
        rc = self.lib.vof_solve_stack_dev(self.h, _ptr(movie), int(n_frames), C.byref(params), _ptr(v_x), _ptr(v_y),
                                          _ptr(remodelling), _ptr(speed), _ptr(st))
        self._check(rc, "vof_solve_stack_dev")
        return st

    def blur_host(self, movie: np.ndarray, weights: np.ndarray):
        """Gaussian blur of every frame on the device (host arrays in and out)."""
        movie = np.ascontiguousarray(movie, dtype=np.float64)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        assert movie.shape[1:] == (self.n_i, self.n_j) and weights.size % 2 == 1
        out = np.empty_like(movie)
        self._check(self.lib.vof_blur_stack_host(self.h, _ptr(movie), _ptr(out), movie.shape[0], _ptr(weights),
                                                 weights.size // 2), "vof_blur_stack_host")
        return out

    def blur_dev(self, movie, out, n_frames, weights: np.ndarray):
        """Gaussian blur of ``n_frames`` device-resident frames (``out`` may alias ``movie``)."""
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        self._check(self.lib.vof_blur_stack_dev(self.h, _ptr(movie), _ptr(out), int(n_frames), _ptr(weights),
                                                weights.size // 2), "vof_blur_stack_dev")

    def vary_regularisation_host(self, movie: np.ndarray, params: VofParams, speed_alphas, remodelling_alphas, weights=None):
        """The whole (speed_alpha, remodelling_alpha) sweep of vary_regularisation on the device; returns a structured
        array (VARIATION_DTYPE) of shape (len(speed_alphas), len(remodelling_alphas))."""
        movie = np.ascontiguousarray(movie, dtype=np.float64)
        assert movie.shape[1:] == (self.n_i, self.n_j)
        sa = np.ascontiguousarray(speed_alphas, dtype=np.float64).ravel()
        ra = np.ascontiguousarray(remodelling_alphas, dtype=np.float64).ravel()
        out = np.zeros((sa.size, ra.size), dtype=VARIATION_DTYPE)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        rc = self.lib.vof_vary_regularisation_host(self.h, _ptr(movie), movie.shape[0], C.byref(params), _ptr(sa), sa.size,
                                                   _ptr(ra), ra.size, _ptr(w), 0 if w is None else w.size // 2, _ptr(out))
        self._check(rc, "vof_vary_regularisation_host")
        return out

    def field_moments_dev(self, field, n):
        """(mean, population variance) of ``n`` device-resident doubles."""
        m, v = C.c_double(), C.c_double()
        self._check(self.lib.vof_field_moments_dev(self.h, _ptr(field), int(n), C.byref(m), C.byref(v)), "vof_field_moments_dev")
        return m.value, v.value

    def subsample_dev(self, field, n_fields, box, offset, out):
        """out[k, a, b] = field[k, a*box + offset, b*box + offset] on device memory."""
        self._check(self.lib.vof_subsample_dev(self.h, _ptr(field), int(n_fields), int(box), int(offset), _ptr(out)),
                    "vof_subsample_dev")

    def texture_stack_dev(self, out, n_frames, mode_params, frame_offsets, period, scale):
        """Synthetic translating texture written to ``out`` (device, (n_frames, n_i, n_j))."""
        mp = np.ascontiguousarray(mode_params, dtype=np.float64)
        fo = np.ascontiguousarray(frame_offsets, dtype=np.float64)
        assert mp.ndim == 2 and mp.shape[0] == 4 and fo.shape == (n_frames, 2)
        self._check(self.lib.vof_texture_stack_dev(self.h, _ptr(out), int(n_frames), _ptr(mp), mp.shape[1], _ptr(fo),
                                                   float(period), float(scale)), "vof_texture_stack_dev")

    def bench_sweeps(self, movie, n_pairs, params, n_sweeps):
        self._check(self.lib.vof_bench_sweeps_dev(self.h, _ptr(movie), n_pairs, C.byref(params), n_sweeps),
                    "vof_bench_sweeps_dev")

    # -- profiler
    def profile_enable(self, on=True):
        self._check(self.lib.vof_profile_enable(self.h, int(bool(on))), "profile_enable")

    def profile_filter(self, kernel=-1, level=-1):
        kid = K_NAMES.index(kernel) if isinstance(kernel, str) else int(kernel)
        self._check(self.lib.vof_profile_filter(self.h, kid, level), "profile_filter")

    def profile_reset(self):
        self._check(self.lib.vof_profile_reset(self.h), "profile_reset")

    def profile_get(self, kernel: int | str, level: int = -1):
        kid = K_NAMES.index(kernel) if isinstance(kernel, str) else int(kernel)
        n, ms = C.c_int64(), C.c_double()
        self._check(self.lib.vof_profile_get(self.h, kid, level, C.byref(n), C.byref(ms)), "profile_get")
        return n.value, ms.value

    def profile_units(self, kernel: int | str, level: int = -1):
        kid = K_NAMES.index(kernel) if isinstance(kernel, str) else int(kernel)
        u = C.c_int64()
        self._check(self.lib.vof_profile_get_units(self.h, kid, level, C.byref(u)), "profile_get_units")
        return u.value

    def profile_bytes(self, kernel: int | str, level: int = -1):
        kid = K_NAMES.index(kernel) if isinstance(kernel, str) else int(kernel)
        u = C.c_double()
        self._check(self.lib.vof_profile_get_bytes(self.h, kid, level, C.byref(u)), "profile_get_bytes")
        return u.value

    def profile_moved(self, kernel: int | str, level: int = -1):
        kid = K_NAMES.index(kernel) if isinstance(kernel, str) else int(kernel)
        u = C.c_double()
        self._check(self.lib.vof_profile_get_moved(self.h, kid, level, C.byref(u)), "profile_get_moved")
        return u.value

    def profile_table(self):
        rows = []
        for kid, name in enumerate(K_NAMES):
            for lvl in range(16):
                n, ms = self.profile_get(kid, lvl)
                if n:
                    rows.append((name, lvl, n, ms))
        return rows

    # -- debug building blocks (host arrays in, host arrays out)
    def debug_setup(self, movie: np.ndarray, params: VofParams):
        movie = np.ascontiguousarray(movie, dtype=np.float64)
        self._dbg_pairs = movie.shape[0] - 1
        self._check(self.lib.vof_debug_setup(self.h, _ptr(movie), self._dbg_pairs, C.byref(params)), "debug_setup")

    def _vec(self, level):
        ni, nj = self.level_shape(level)
        return (self._dbg_pairs, 3, ni, nj)

    def debug_rhs(self):
        b = np.empty(self._vec(0))
        self._check(self.lib.vof_debug_rhs(self.h, _ptr(b)), "debug_rhs")
        return b

    def debug_apply(self, level, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self._vec(level))
        y = np.empty_like(x)
        self._check(self.lib.vof_debug_apply(self.h, level, _ptr(x), _ptr(y)), "debug_apply")
        return y

    def debug_gs(self, level, x, b, colour):
        x = np.array(x, dtype=np.float64, copy=True).reshape(self._vec(level))
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(self._vec(level))
        self._check(self.lib.vof_debug_gs(self.h, level, _ptr(x), _ptr(b), colour), "debug_gs")
        return x

    def debug_sweep(self, level, x, b, reverse=False, from_zero=False):
        x = np.array(x, dtype=np.float64, copy=True).reshape(self._vec(level))
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(self._vec(level))
        self._check(self.lib.vof_debug_sweep(self.h, level, _ptr(x), _ptr(b), int(reverse), int(from_zero)), "debug_sweep")
        return x

    def debug_smooth(self, level, x, b, nu, reverse=False, from_zero=False):
        x = np.array(x, dtype=np.float64, copy=True).reshape(self._vec(level))
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(self._vec(level))
        self._check(self.lib.vof_debug_smooth(self.h, level, _ptr(x), _ptr(b), int(nu), int(reverse), int(from_zero)), "debug_smooth")
        return x

    def set_fused_sweeps(self, on=True):
        self._check(self.lib.vof_set_fused_sweeps(self.h, int(bool(on))), "set_fused_sweeps")

    def debug_restrict(self, level, fine):
        fine = np.ascontiguousarray(fine, dtype=np.float64).reshape(self._vec(level))
        coarse = np.empty(self._vec(level + 1))
        self._check(self.lib.vof_debug_restrict(self.h, level, _ptr(fine), _ptr(coarse)), "debug_restrict")
        return coarse

    def debug_resrestrict_u(self, level, x_new, x_old=None):
        x_new = np.ascontiguousarray(x_new, dtype=np.float64).reshape(self._vec(level))
        if x_old is not None:
            x_old = np.ascontiguousarray(x_old, dtype=np.float64).reshape(self._vec(level))
        coarse = np.empty(self._vec(level + 1))
        self._check(self.lib.vof_debug_resrestrict_u(self.h, level, _ptr(x_new), _ptr(x_old) if x_old is not None else None,
                                                     _ptr(coarse)), "debug_resrestrict_u")
        return coarse

    def debug_prolong_add(self, level, fine, coarse):
        fine = np.array(fine, dtype=np.float64, copy=True).reshape(self._vec(level))
        coarse = np.ascontiguousarray(coarse, dtype=np.float64).reshape(self._vec(level + 1))
        self._check(self.lib.vof_debug_prolong_add(self.h, level, _ptr(fine), _ptr(coarse)), "debug_prolong_add")
        return fine

    def debug_stencil(self, level):
        ni, nj = self.level_shape(level)
        c = np.empty((self._dbg_pairs, 3, 3, 3, 3, ni, nj))
        self._check(self.lib.vof_debug_stencil(self.h, level, _ptr(c)), "debug_stencil")
        return c

    def debug_vcycle(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(self._vec(0))
        e = np.empty_like(r)
        self._check(self.lib.vof_debug_vcycle(self.h, _ptr(r), _ptr(e)), "debug_vcycle")
        return e

    def debug_vcycle_apply(self, r):
        """(y, v, dots, fused): y = M r, v = A y, dots[k] = ((v, r), (v, v)) of pair k."""
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(self._vec(0))
        y, v = np.empty_like(r), np.empty_like(r)
        dots = np.zeros((r.shape[0], 2))
        fused = C.c_int(0)
        self._check(self.lib.vof_debug_vcycle_apply(self.h, _ptr(r), _ptr(y), _ptr(v), _ptr(dots), C.byref(fused)), "debug_vcycle_apply")
        return y, v, dots, bool(fused.value)

    def debug_coarse_solve(self, r):
        last = self.num_levels - 1
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(self._vec(last))
        e = np.empty_like(r)
        self._check(self.lib.vof_debug_coarse_solve(self.h, _ptr(r), _ptr(e)), "debug_coarse_solve")
        return e


def query_workspace(n_i, n_j, pairs, coarse_precision=None, vcycle_precision=None):
    """Device bytes of a context for ``pairs`` frame pairs in flight; the stencil storage depends on the format in use
    (``coarse_precision`` 0..3, default: the library's default format)."""
    if coarse_precision is None and vcycle_precision is None:
        return int(load_library().vof_query_workspace(int(n_i), int(n_j), int(pairs)))
    return int(load_library().vof_query_workspace_for(int(n_i), int(n_j), int(pairs), 3 if coarse_precision is None else int(coarse_precision),
                                                      3 if vcycle_precision is None else int(vcycle_precision)))


def device_memory(device=0):
    f, t = C.c_size_t(), C.c_size_t()
    rc = load_library().vof_device_memory(int(device), C.byref(f), C.byref(t))
    if rc != 0:
        raise VofError("no usable HIP device %d (vof_device_memory rc=%d); this package has no CPU fallback" % (device, rc))
    return f.value, t.value
