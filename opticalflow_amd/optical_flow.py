"""Drop-in for the hot path of the reference module ``source/optical_flow.py``.

``variational_optical_flow`` keeps the reference's signature and result dictionary
(source/optical_flow.py:715-724, 1193-1205) so that ``analysis/analyse_variational_optical_flow.py``
can call it unchanged; the per-pair scipy.sparse assembly + PETSc KSP solve (OF.py:833-1145) is
replaced by the batched HIP solver in ``csrc/`` (BiCGStab + geometric multigrid).
The solve always runs on an MI355X through libvof.so; there is no CPU path in this module.
"""
from __future__ import annotations

import atexit
import contextlib
import threading
import time

import numpy as np

from . import _native

# ---------------------------------------------------------------------------------------------------------
# One native context (device workspace) is kept between calls: creating / destroying tens of GB of device memory costs
# 0.15-1.9 s per call, more than the solve of a whole stack.  It is re-used when the image size and the device match and
# its batch is large enough, replaced otherwise, and released by ``release_device_memory()`` or at interpreter exit.
# A context serves one thread at a time: a second thread calling concurrently gets a private, short-lived context.
# ---------------------------------------------------------------------------------------------------------
_cache_lock = threading.Lock()
_cache = {"key": None, "solver": None}


@contextlib.contextmanager
def _device_context(n_i, n_j, pairs, device, exact=False):
    device = int(device)
    if not _cache_lock.acquire(blocking=False):
        with _native.Solver(n_i, n_j, pairs, device=device) as solver:
            yield solver
        return
    try:
        solver = _cache["solver"]
        # exact: the caller asked for this batch size (max_pairs_in_flight): honour it instead of a larger cached batch
        if (solver is None or not solver.h or _cache["key"] != (n_i, n_j, device) or solver.max_pairs < pairs
                or (exact and solver.max_pairs != pairs)):
            release_device_memory(_locked=True)
            solver = _native.Solver(n_i, n_j, pairs, device=device)
            _cache["key"], _cache["solver"] = (n_i, n_j, device), solver
        try:
            yield solver
        except BaseException:
            release_device_memory(_locked=True)      # do not keep a context whose call failed half-way
            raise
    finally:
        _cache_lock.release()


def release_device_memory(_locked=False):
    """Free the device workspace kept between calls (it is re-created on the next call)."""
    if not _locked:
        with _cache_lock:
            return release_device_memory(_locked=True)
    if _cache["solver"] is not None:
        _cache["solver"].close()
    _cache["key"], _cache["solver"] = None, None


atexit.register(release_device_memory)

__all__ = ["variational_optical_flow", "vary_regularisation", "make_fake_data_frame", "blur_movie",
           "format_elapsed_time", "apply_constant_boundary_condition", "choose_pairs_in_flight",
           "subsample_velocities_for_visualisation", "costum_imshow", "make_velocity_overlay_movie",
           "make_joint_overlay_movie", "release_device_memory"]


def make_fake_data_frame(x_position, y_position, sigma=1.0, width=20.0, include_noise=False, dimension=1000):
    """Synthetic Gaussian-hat frame, same arguments and return value as OF.py:376-423:
    ``frame[i, j] = exp((-(x_i - x0)^2 - (y_j - y0)^2) / sigma^2)`` on ``linspace(0, width, dimension)``;
    returns ``(frame, delta_x)``."""
    x = np.linspace(0, width, dimension)
    y = np.linspace(0, width, dimension)
    frame = np.exp((-(x[:, None] - x_position) ** 2 - (y[None, :] - y_position) ** 2) / sigma ** 2)
    delta_x = x[1] - x[0]
    if include_noise:
        frame = np.abs(frame + np.random.rand(dimension, dimension) * 0.0000001)
    return frame, delta_x


def gaussian_taps(sigma, truncate=4.0):
    """The normalised 1-D taps scipy.ndimage.gaussian_filter uses for ``sigma`` (skimage passes truncate=4.0):
    ``radius = int(truncate * sigma + 0.5)``, ``w = exp(-0.5 x^2 / sigma^2) / sum``."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def blur_movie(movie, smoothing_sigma, device=0, _solver=None):
    """Per-frame Gaussian blur on the GPU, same arguments and result as OF.py:282-306.  The reference calls
    ``skimage.filters.gaussian(frame, sigma, preserve_range=True)``, i.e. ``scipy.ndimage.gaussian_filter(frame,
    sigma, mode='nearest', truncate=4.0)``: two 1-D correlations (axis 0, then axis 1) with clamped edges; the
    HIP kernel keeps scipy's summation order, so the result agrees with the host filter to rounding."""
    movie = np.asarray(movie)
    if movie.ndim != 3:
        raise ValueError("movie must be a 3-D array (frames, x, y)")
    taps = gaussian_taps(smoothing_sigma)
    frames = np.ascontiguousarray(movie, dtype=np.float64)
    if _solver is not None:
        return _solver.blur_host(frames, taps)
    with _device_context(max(4, movie.shape[1]), max(4, movie.shape[2]), 1, device) as solver:
        if (solver.n_i, solver.n_j) != movie.shape[1:]:
            raise ValueError("frames must be at least 4x4")
        return solver.blur_host(frames, taps)


def format_elapsed_time(time_difference):
    """(minutes, seconds, milliseconds) of a ``time.time()`` difference, OF.py:1212-1238."""
    minutes = int(time_difference // 60)
    seconds = int(time_difference % 60)
    milliseconds = int((time_difference - int(time_difference)) * 1000)
    return minutes, seconds, milliseconds


def apply_constant_boundary_condition(image):
    """In-place mirror of the border lines (OF.py:1304-1316): rows first, then columns."""
    image[0, :] = image[2, :]
    image[-1, :] = image[-3, :]
    image[:, 0] = image[:, 2]
    image[:, -1] = image[:, -3]


def choose_pairs_in_flight(n_i, n_j, n_pairs, device=0, memory_fraction=0.6, cap=128, params=None, staging=True):
    """Largest batch of frame pairs whose workspace fits in ``memory_fraction`` of the free HBM (``params``: the solver
    parameters of the call - the stencil storage is sized by their ``coarse_precision``)."""
    free, _total = _native.device_memory(device)
    budget = free * memory_fraction
    fmt = (None, None) if params is None else (int(params.coarse_precision), int(params.vcycle_precision))
    per_pair = _native.query_workspace(n_i, n_j, 1, *fmt) + (9 * n_i * n_j * 8 if staging else 0)  # + host-API staging (two output sets)
    return int(max(1, min(n_pairs, cap, budget // max(per_pair, 1))))


def _float64_copy(a, n_threads=4, wait=True):
    """``a.astype(np.float64)`` (always a copy, OF.py:769); stacks of more than 128 MB are converted by a few threads
    (numpy copies release the GIL; the single-threaded copy of a 2 GB movie takes 0.16 s, a sixth of the whole call).
    ``wait=False``: returns ``(out, join)`` with the copy still running - ``join()`` waits for it."""
    if a.ndim < 1 or a.shape[0] < n_threads or a.size < (1 << 24):
        out = a.astype(np.float64)
        return out if wait else (out, lambda: None)
    out = np.empty(a.shape, dtype=np.float64)
    bounds = np.linspace(0, a.shape[0], n_threads + 1).astype(int)

    def work(i0, i1):
        out[i0:i1] = a[i0:i1]
    threads = [threading.Thread(target=work, args=(bounds[k], bounds[k + 1])) for k in range(n_threads)]
    for t in threads:
        t.start()

    def join():
        for t in threads:
            t.join()
    if wait:
        join()
        return out
    return out, join


def _solver_params(speed_alpha, remodelling_alpha, delta_x, delta_t, initial_v_x, initial_v_y, initial_remodelling,
                   use_direct_solver, rtol, max_iterations, reference_quirks, coarse_precision, vcycle_precision,
                   multigrid_sweeps, w_cycle_level, krylov_method="auto", gmres_restart=None, warm_start_stride=None,
                   preconditioner=None):
    """vof_params from the keyword arguments of ``variational_optical_flow``.  ``krylov_method`` may be a tuple
    ``("auto", fallback_after)``: BiCGStab iterations before GMRES takes over."""
    fallback_after = None
    if isinstance(krylov_method, (tuple, list)):
        krylov_method, fallback_after = krylov_method[0], int(krylov_method[1])
    if rtol is None:
        rtol = 1e-10 if use_direct_solver else 1e-6     # direct branch: the accuracy a sparse LU itself attains on these systems
    params = _native.default_params(
        speed_alpha=float(speed_alpha), remodelling_alpha=float(remodelling_alpha), delta_x=float(delta_x),
        delta_t=float(delta_t), initial_v_x=float(initial_v_x), initial_v_y=float(initial_v_y),
        initial_remodelling=float(initial_remodelling), rtol=float(rtol), max_iterations=int(max_iterations),
        reference_quirks=int(bool(reference_quirks)),
        coarse_precision={"float64": 0, "float32": 1, "bfloat16": 2, "float8": 3}[coarse_precision],
        vcycle_precision={"float64": 0, "float32": 1, "auto": 2, "coarse_float32": 3}[vcycle_precision])
    if multigrid_sweeps is not None:     # (pre, post) on level 0 [, (pre, post) on the coarse levels]
        ms = tuple(int(v) for v in multigrid_sweeps)
        params.nu_pre, params.nu_post = ms[0], ms[1]
        if len(ms) == 4:
            params.nu_pre_coarse, params.nu_post_coarse = ms[2], ms[3]
    if w_cycle_level is not None:        # -1: plain V-cycle; l or (l, visits): level l visits level l+1 several times
        if isinstance(w_cycle_level, (tuple, list)):
            params.w_cycle_level, params.w_cycle_visits = int(w_cycle_level[0]), int(w_cycle_level[1])
        else:
            params.w_cycle_level = int(w_cycle_level)
    params.krylov_method = {"bicgstab": 0, "gmres": 1, "auto": 2}[krylov_method]
    if gmres_restart is not None:
        params.gmres_restart = int(gmres_restart)
    if fallback_after is not None:
        params.fallback_after = fallback_after
    if warm_start_stride is not None:
        params.warm_start_stride = int(warm_start_stride)
    if preconditioner is None:           # the reference's direct branch: the direct preconditioner (falls back if it cannot fit)
        preconditioner = "direct" if use_direct_solver else "auto"
    params.preconditioner = {"multigrid": 0, "direct": 1, "auto": 2}[preconditioner]
    return params


def _direct_unavailable(exc):
    """True if a native call failed only because the direct preconditioner cannot be used here (too large for the free
    device memory, or no rocSOLVER): ``use_direct_solver=True`` then falls back to the multigrid path at rtol 1e-10."""
    msg = str(exc)
    return "direct preconditioner does not fit" in msg or "cannot load rocSOLVER" in msg or "rocSOLVER / rocBLAS symbols" in msg


def variational_optical_flow(movie,
                             delta_x=1.0,
                             delta_t=1.0,
                             speed_alpha=1.0,
                             remodelling_alpha=1000.0,
                             smoothing_sigma=None,
                             initial_v_x=0.0,
                             initial_v_y=0.0,
                             initial_remodelling=0.0,
                             use_direct_solver=False,
                             *,
                             rtol=None,
                             max_iterations=1000,
                             reference_quirks=True,
                             device=0,
                             max_pairs_in_flight=None,
                             coarse_precision="float8",
                             vcycle_precision="coarse_float32",
                             multigrid_sweeps=None,
                             w_cycle_level=None,
                             krylov_method="auto",
                             gmres_restart=None,
                             warm_start_stride=None,
                             preconditioner=None,
                             verbose=False,
                             return_stats=False,
                             output="numpy",
                             _solver=None):
    """Variational optical flow with remodelling on an image stack, on one MI355X.

    Positional/keyword arguments up to ``use_direct_solver`` have the reference's meaning
    (OF.py:725-762).  ``movie`` is ``(T, N_i, N_j)``, any real dtype; the result ``k`` is the flow
    from frame ``k`` to ``k+1``.  Returns the reference's result dict (OF.py:1193-1205): ``v_x``,
    ``v_y``, ``speed``, ``remodelling`` (float64 ``(T-1, N_i, N_j)``), ``original_data``,
    ``blurred_data``, ``delta_x``, ``delta_t``, ``converged`` (flag of the LAST pair, as in the
    reference), ``L1_functional``, ``remodelling_functional``, ``speed_functional``.

    Differences, all opt-in or invisible at the reference's tolerance:
      * all pairs are solved concurrently instead of warm-starting pair k from pair k-1 (OF.py:803-806): from the same
        constant initial guess, or (device-resident mode, large stacks) every 3rd pair first and the others from their
        solved neighbour; the converged answer is the same to solver tolerance;
      * ``use_direct_solver=True`` (SuperLU in the reference, OF.py:1146-1147) selects the direct preconditioner - a
        block-tridiagonal LU of the system by image rows on the GPU - inside the same Krylov iteration, converged to
        rtol=1e-10 (one or two iterations); if its buffers (``n_i (3 n_j)^2`` doubles per pair) do not fit, the multigrid
        path is run to rtol=1e-10 instead;
      * keyword-only extras: ``rtol`` (default 1e-6 = OF.py:1120), ``max_iterations`` (1000),
        ``reference_quirks`` (True keeps OF.py:698-699 'dy'=='dx' and the OF.py:1205
        ``speed_functional`` assignment), ``device``, ``max_pairs_in_flight``, ``coarse_precision`` (storage of the Galerkin
        stencils of the multigrid preconditioner: "float8" (default: 8-bit float off-diagonal blocks whose rounding errors
        are folded into a float32 diagonal block, so block row sums are exact - 1 % more iterations than "float32" on a third
        of the bytes), "bfloat16" (the same with 16-bit off-diagonal blocks: iteration counts of "float32"), "float32",
        "float64") / ``vcycle_precision`` (storage of the V-cycle vectors: "coarse_float32" (default: float64 on
        level 0, float32 on the coarser levels), "float64", "float32" or "auto" = float32; the float32 modes return to
        float64 for pairs that need more than 8 iterations; all arithmetic, the Krylov iteration, the stopping rule and the
        result are float64 either way), ``multigrid_sweeps``
        (block-GS sweeps per V-cycle: ``(pre, post)`` on level 0 and optionally ``(pre, post)`` on the coarse levels),
        ``w_cycle_level`` (-1: V-cycle; ``l``: level ``l`` visits level ``l+1`` twice per cycle),
        ``krylov_method`` ("bicgstab": the reference's KSP type, OF.py:1081; "gmres": restarted GMRES with the same
        preconditioner and stopping rule; "auto" (default): BiCGStab, and GMRES(``gmres_restart``, default 100) for the
        pairs that have not converged after 25 iterations - the grad-div dominated regimes),
        ``preconditioner`` ("auto" (default): the multigrid cycle, and pairs it leaves unconverged - the grad-div dominated
        regimes, e.g. 8-bit data with ``speed_alpha`` below ~1e5 - are solved once more with the direct preconditioner when
        that fits; "multigrid"; "direct"),
        ``warm_start_stride`` (both modes, stacks whose first phase fills the chip: every n-th pair of a batch is solved
        first, the others start from their solved neighbour, cf. OF.py:803-806; default 3, 0 = every pair from the
        constant initial fields),
        ``verbose``, ``return_stats`` (adds ``result['stats']``: per-pair iterations / residual /
        converged / functionals), ``output`` ("numpy": host arrays as in the reference; "torch": ``movie`` may be a
        torch tensor already on the device and every array of the result stays on the device as a float64 torch
        tensor - blur, solve and epilogue without PCIe traffic, see ``subsample_velocities_for_visualisation``).
    """
    if output == "torch":
        return _variational_optical_flow_device(movie, smoothing_sigma, device, max_pairs_in_flight, verbose, return_stats,
                                                reference_quirks, _solver_params(
                                                    speed_alpha, remodelling_alpha, delta_x, delta_t, initial_v_x, initial_v_y,
                                                    initial_remodelling, use_direct_solver, rtol, max_iterations,
                                                    reference_quirks, coarse_precision, vcycle_precision, multigrid_sweeps,
                                                    w_cycle_level, krylov_method, gmres_restart, warm_start_stride, preconditioner),
                                                delta_x, delta_t, direct_fallback=bool(use_direct_solver and preconditioner is None))
    if output != "numpy":
        raise ValueError("output must be 'numpy' or 'torch'")
    source = np.asarray(movie)
    if source.ndim != 3:
        raise ValueError("movie must be a 3-D array (frames, x, y)")
    T, N_i, N_j = source.shape
    if T < 2:
        raise ValueError("movie needs at least two frames")
    # OF.py:769: the reference works on (and returns, as 'original_data') a float64 COPY of the movie.  A stack that already is
    # float64 and contiguous is only read by the solver, so the copy the result needs is made WHILE the solve runs (46 ms of a
    # 0.39-s call at 1024 x 1024 x 256); anything else is converted first, as the reference does
    if source.dtype == np.float64 and source.flags.c_contiguous:
        movie, copy_done = _float64_copy(source, wait=False)
        movie_in = source
    else:
        movie = _float64_copy(source)
        copy_done = None
        movie_in = movie
    params = _solver_params(speed_alpha, remodelling_alpha, delta_x, delta_t, initial_v_x, initial_v_y, initial_remodelling,
                            use_direct_solver, rtol, max_iterations, reference_quirks, coarse_precision, vcycle_precision,
                            multigrid_sweeps, w_cycle_level, krylov_method, gmres_restart, warm_start_stride, preconditioner)
    exact = max_pairs_in_flight is not None
    if max_pairs_in_flight is None and _solver is None:
        max_pairs_in_flight = choose_pairs_in_flight(N_i, N_j, T - 1, device, params=params)
    t0 = time.time()
    # a caller-owned context (_solver) or the module's cached one; blur and solve share it
    try:
        with (contextlib.nullcontext(_solver) if _solver is not None
              else _device_context(N_i, N_j, max_pairs_in_flight, device, exact)) as solver:
            if smoothing_sigma is not None:                                 # OF.py:770-773
                movie_to_analyse = blur_movie(movie_in, smoothing_sigma=smoothing_sigma, device=device, _solver=solver)
            else:
                movie_to_analyse = movie_in
            try:
                v_x, v_y, remodelling, speed, stats = solver.solve_host(np.ascontiguousarray(movie_to_analyse), params)
            except _native.VofError as exc:
                if not (use_direct_solver and preconditioner is None and _direct_unavailable(exc)):
                    raise
                params.preconditioner = 2        # too large for the direct preconditioner: multigrid to the same tight tolerance
                v_x, v_y, remodelling, speed, stats = solver.solve_host(np.ascontiguousarray(movie_to_analyse), params)
    finally:
        if copy_done is not None:
            copy_done()
    if smoothing_sigma is None:
        movie_to_analyse = movie             # (the reference's 'blurred_data' is its float64 copy when nothing is blurred)
    if verbose:
        m, s, ms = format_elapsed_time(time.time() - t0)
        print(f"Elapsed time for solve: {m} minutes, {s} seconds, {ms} milliseconds")
        for k in range(T - 1):
            print(f"pair {k + 1}: iterations {stats['iterations'][k]}, relative residual "
                  f"{stats['relative_residual'][k]:.3e}, converged {bool(stats['converged'][k])}")
        if not stats["converged"].all():
            print("the solver has not actually converged, the result will be incorrect or inaccurate")

    result = dict()
    result["v_x"] = v_x
    result["v_y"] = v_y
    result["speed"] = speed
    result["remodelling"] = remodelling
    result["original_data"] = movie
    result["delta_x"] = delta_x
    result["delta_t"] = delta_t
    result["blurred_data"] = movie_to_analyse
    result["converged"] = bool(stats["converged"][-1])                  # OF.py:1202: last pair only
    result["L1_functional"] = float(np.sum(stats["L1_functional"]))
    result["remodelling_functional"] = float(np.sum(stats["remodelling_functional"]))
    # OF.py:1205 stores the remodelling sum under 'speed_functional'
    result["speed_functional"] = (result["remodelling_functional"] if reference_quirks
                                  else float(np.sum(stats["speed_functional"])))
    if return_stats:
        result["stats"] = stats
    return result


def _variational_optical_flow_device(movie, smoothing_sigma, device, max_pairs_in_flight, verbose, return_stats,
                                     reference_quirks, params, delta_x, delta_t, direct_fallback=False):
    """``output="torch"`` branch of ``variational_optical_flow``: torch only allocates the device arrays."""
    import torch
    dev = torch.device("cuda", int(device))
    movie = torch.as_tensor(movie).to(device=dev, dtype=torch.float64).contiguous()      # OF.py:769
    if movie.ndim != 3:
        raise ValueError("movie must be a 3-D array (frames, x, y)")
    T, N_i, N_j = movie.shape
    if T < 2:
        raise ValueError("movie needs at least two frames")
    exact = max_pairs_in_flight is not None
    if max_pairs_in_flight is None:
        # device-resident call: no staging buffers, and what is free now is free of the caller's tensors already
        max_pairs_in_flight = choose_pairs_in_flight(N_i, N_j, T - 1, int(device), memory_fraction=0.8, params=params, staging=False)
    out = [torch.empty((T - 1, N_i, N_j), dtype=torch.float64, device=dev) for _ in range(4)]
    torch.cuda.synchronize(dev)          # the library launches on its own stream
    with _device_context(N_i, N_j, max_pairs_in_flight, device, exact) as solver:
        if smoothing_sigma is not None:                                                  # OF.py:770-773
            taps = gaussian_taps(smoothing_sigma)
            movie_to_analyse = torch.empty_like(movie)
            solver.blur_dev(movie, movie_to_analyse, T, taps)
        else:
            movie_to_analyse = movie
        try:
            stats = solver.solve_dev(movie_to_analyse, T, params, out[0], out[1], out[2], out[3])
        except _native.VofError as exc:
            # same rule as the host path: only use_direct_solver=True (not an explicit preconditioner="direct") may fall back
            if not (direct_fallback and _direct_unavailable(exc)):
                raise
            params.preconditioner = 2
            stats = solver.solve_dev(movie_to_analyse, T, params, out[0], out[1], out[2], out[3])
    if verbose:
        print(f"iterations {stats['iterations'].tolist()}, converged {stats['converged'].astype(bool).tolist()}")
    result = dict(v_x=out[0], v_y=out[1], speed=out[3], remodelling=out[2], original_data=movie, delta_x=delta_x,
                  delta_t=delta_t, blurred_data=movie_to_analyse, converged=bool(stats["converged"][-1]),
                  L1_functional=float(np.sum(stats["L1_functional"])),
                  remodelling_functional=float(np.sum(stats["remodelling_functional"])))
    result["speed_functional"] = (result["remodelling_functional"] if reference_quirks          # OF.py:1205
                                  else float(np.sum(stats["speed_functional"])))
    if return_stats:
        result["stats"] = stats
    return result


def vary_regularisation(movie,
                        speed_alpha_values=np.arange(500, 2000, 500),
                        remodelling_alpha_values=np.arange(500, 2000, 500),
                        filename=None,
                        **kwargs):
    """Vary both regularisation parameters and keep the summary statistics for heat-maps; same arguments,
    result dictionary and optional ``np.save`` as the reference (OF.py:1918-1998).  ``kwargs`` are the keyword
    arguments of ``variational_optical_flow`` (OF.py:1974-1977).

    Every ``(speed_alpha, remodelling_alpha)`` combination is an independent solve of the same movie.  The whole sweep
    is one native call (``vof_vary_regularisation_host``): the movie is uploaded and blurred once, each combination
    batches all frame pairs, and ``np.mean`` / ``np.var`` of the speed and remodelling stacks (OF.py:1978-1981) are
    two-pass reductions on the device, so only the summary scalars cross PCIe.

    Returns a dict with ``speed_alpha_values``, ``remodelling_alpha_values``, ``speed_means``,
    ``speed_variances``, ``remodelling_means``, ``remodelling_variances``, ``converged`` and ``functional``
    (``L1_functional + speed_functional + remodelling_functional``, OF.py:1983), each of shape
    ``(len(speed_alpha_values), len(remodelling_alpha_values))``.  Extra keyword: ``return_stats=True`` adds
    ``result['stats']`` (the native per-combination records: worst residual / iteration count, all-pairs flag).
    """
    movie = np.asarray(movie)
    if movie.ndim != 3:
        raise ValueError("movie must be a 3-D array (frames, x, y)")
    T, N_i, N_j = movie.shape
    if T < 2:
        raise ValueError("movie needs at least two frames")
    kw = dict(delta_x=1.0, delta_t=1.0, smoothing_sigma=None, initial_v_x=0.0, initial_v_y=0.0, initial_remodelling=0.0,
              use_direct_solver=False, rtol=None, max_iterations=1000, reference_quirks=True, device=0,
              max_pairs_in_flight=None, coarse_precision="float8", vcycle_precision="coarse_float32", multigrid_sweeps=None,
              w_cycle_level=None, krylov_method="auto", gmres_restart=None, warm_start_stride=None, preconditioner=None,
              verbose=False, return_stats=False)
    for k in kwargs:
        if k not in kw:
            raise TypeError(f"variational_optical_flow() got an unexpected keyword argument {k!r}")
    kw.update(kwargs)
    params = _solver_params(1.0, 1.0, kw["delta_x"], kw["delta_t"], kw["initial_v_x"], kw["initial_v_y"],
                            kw["initial_remodelling"], kw["use_direct_solver"], kw["rtol"], kw["max_iterations"],
                            kw["reference_quirks"], kw["coarse_precision"], kw["vcycle_precision"],
                            kw["multigrid_sweeps"], kw["w_cycle_level"], kw["krylov_method"], kw["gmres_restart"],
                            kw["warm_start_stride"], kw["preconditioner"])   # warm_start_stride: accepted for signature parity
    taps = None if kw["smoothing_sigma"] is None else gaussian_taps(kw["smoothing_sigma"])
    # short movies: several combinations share one batch as "virtual pairs" (see vof_vary_regularisation_host)
    n_comb = max(1, len(speed_alpha_values) * len(remodelling_alpha_values))
    pairs = kw["max_pairs_in_flight"] or choose_pairs_in_flight(N_i, N_j, (T - 1) * n_comb, kw["device"], params=params)
    with _device_context(N_i, N_j, pairs, kw["device"], kw["max_pairs_in_flight"] is not None) as solver:
        try:
            rec = solver.vary_regularisation_host(movie.astype(np.float64), params, speed_alpha_values,
                                                  remodelling_alpha_values, taps)
        except _native.VofError as exc:
            if not (kw["use_direct_solver"] and kw["preconditioner"] is None and _direct_unavailable(exc)):
                raise
            params.preconditioner = 2
            rec = solver.vary_regularisation_host(movie.astype(np.float64), params, speed_alpha_values,
                                                  remodelling_alpha_values, taps)
    if kw["verbose"]:
        for i, a in enumerate(speed_alpha_values):
            for j, b in enumerate(remodelling_alpha_values):
                print(f"speed_alpha {a}, remodelling_alpha {b}: max iterations {rec['max_iterations_used'][i, j]}, "
                      f"max relative residual {rec['max_relative_residual'][i, j]:.3e}, "
                      f"all converged {bool(rec['converged_all'][i, j])}")
    # OF.py:1205 stores the remodelling sum under 'speed_functional'; OF.py:1983 adds the three dict entries
    speed_functional = rec["remodelling_functional"] if kw["reference_quirks"] else rec["speed_functional"]
    result_dict = {}
    result_dict["speed_alpha_values"] = speed_alpha_values
    result_dict["remodelling_alpha_values"] = remodelling_alpha_values
    result_dict["speed_means"] = rec["speed_mean"].copy()
    result_dict["speed_variances"] = rec["speed_variance"].copy()
    result_dict["remodelling_means"] = rec["remodelling_mean"].copy()
    result_dict["remodelling_variances"] = rec["remodelling_variance"].copy()
    result_dict["converged"] = rec["converged_last"].astype(bool)       # OF.py:1982: flag of the last pair
    result_dict["functional"] = rec["L1_functional"] + speed_functional + rec["remodelling_functional"]
    if kw["return_stats"]:
        result_dict["stats"] = rec
    if filename is not None:
        np.save(filename, result_dict)
    return result_dict


# ---------------------------------------------------------------------------------------------------------
# Result consumers: the steps the reference's scripts run right after the solve.
# ---------------------------------------------------------------------------------------------------------
def subsample_velocities_for_visualisation(flow_result, iteration=None, arrow_boxsize=5):
    """Arrow positions and velocities for ``plt.quiver``, same arguments and return values as OF.py:1574-1646:
    one sample per ``arrow_boxsize`` x ``arrow_boxsize`` box, taken at pixel ``box_index * arrow_boxsize +
    round(arrow_boxsize / 2)`` (Python's ``round``, i.e. half-to-even, as in the reference); positions are in
    ``delta_x`` units.  Returns ``(x_positions, y_positions, v_x, v_y)`` with the velocities of shape
    ``(T-1, N_i // arrow_boxsize, N_j // arrow_boxsize)``.

    A device-resident result (``output="torch"``) is sampled by a HIP gather kernel and only the
    ``1 / arrow_boxsize^2`` samples cross PCIe.  ``iteration`` selects ``'v_x_steps'`` / ``'v_y_steps'`` entries
    (OF.py:1621-1625), which only the reference's iterative legacy solvers produce."""
    box = int(arrow_boxsize)
    if box < 1:
        raise ValueError("arrow_boxsize must be >= 1")
    offset = round(arrow_boxsize / 2)
    if iteration is not None:
        fields = [flow_result["v_x_steps"][:, iteration], flow_result["v_y_steps"][:, iteration]]
    else:
        fields = [flow_result["v_x"], flow_result["v_y"]]
    n_pairs = flow_result["original_data"].shape[0] - 1
    n_x, n_y = fields[0].shape[1], fields[1].shape[2]
    nbx, nby = n_x // box, n_y // box
    if hasattr(fields[0], "data_ptr") and fields[0].is_cuda:
        import torch
        sub = []
        with _device_context(n_x, n_y, 1, fields[0].device.index) as solver:
            for f in fields:
                out = torch.empty((n_pairs, nbx, nby), dtype=torch.float64, device=f.device)
                torch.cuda.synchronize(f.device)
                solver.subsample_dev(f[:n_pairs].contiguous(), n_pairs, box, offset, out)
                sub.append(out.cpu().numpy())
    else:
        sub = [np.array(np.asarray(f)[:n_pairs, offset:offset + (nbx - 1) * box + 1:box,
                                      offset:offset + (nby - 1) * box + 1:box], dtype=np.float64)
               if nbx and nby else np.zeros((n_pairs, nbx, nby)) for f in fields]
    delta_x = flow_result["delta_x"]
    x_positions = (np.arange(nbx) * box + offset).astype(float) / n_x * (n_x * delta_x)
    y_positions = (np.arange(nby) * box + offset).astype(float) / n_y * (n_y * delta_x)
    return x_positions, y_positions, sub[0], sub[1]


def costum_imshow(image, delta_x, cmap="gray_r", autoscale=False, v_min=0.0, v_max=255.0, unit=r"$\mathrm{\mu}$m"):
    """Show an image without anti-aliasing, axes in physical units (same name, arguments and look as OF.py:1531-1572;
    the figure / axes are created by the caller)."""
    import matplotlib.pyplot as plt
    limits = dict(vmin=None, vmax=None) if autoscale else dict(vmin=v_min, vmax=v_max)
    extent = [0, image.shape[1] * delta_x, image.shape[0] * delta_x, 0]
    plt.imshow(image, cmap=cmap, extent=extent, interpolation=None, **limits)
    plt.xlabel("y-position [" + unit + "]")
    plt.ylabel("x-position [" + unit + "]")


def _quiver(x_positions, y_positions, v_x, v_y, arrow_color, arrow_scale, arrow_width):
    # image rows run downwards: plot (y, x) with the x-velocity flipped (OF.py:1695)
    import matplotlib.pyplot as plt
    plt.quiver(y_positions, x_positions, v_y, -v_x, color=arrow_color, headwidth=5, scale=1.0 / arrow_scale,
               width=arrow_width)


def make_velocity_overlay_movie(flow_result, filename, arrow_boxsize=5, arrow_scale=1.0, cmap="gray_r", autoscale=False,
                                arrow_color="magenta", arrow_width=None, v_min=0.0, v_max=255.0, dpi=600):
    """Movie of the data with the flow arrows on top; arguments as OF.py:1649-1700 (``dpi`` is an extra)."""
    import matplotlib.pyplot as plt
    from matplotlib.animation import FuncAnimation
    movie = flow_result["original_data"]
    xs, ys, v_x, v_y = subsample_velocities_for_visualisation(flow_result, arrow_boxsize=arrow_boxsize)
    fig = plt.figure(figsize=(2.5, 2.5))

    def animate(i):
        plt.cla()
        costum_imshow(movie[i + 1], delta_x=flow_result["delta_x"], cmap=cmap, autoscale=autoscale, v_min=v_min, v_max=v_max)
        _quiver(xs, ys, v_x[i], v_y[i], arrow_color, arrow_scale, arrow_width)
        if i < 1:
            plt.tight_layout()
    FuncAnimation(fig, animate, frames=movie.shape[0] - 1).save(filename, dpi=dpi)
    plt.close(fig)


def make_joint_overlay_movie(flow_result, filename, arrow_boxsize=5, arrow_scale=1.0, arrow_width=None, cmap="gray_r",
                             autoscale=False, arrow_color="magenta", v_min=0.0, v_max=255.0, dpi=600):
    """Six-panel movie: data and blurred data with arrows, speed, net remodelling, v_x, v_y; arguments as
    OF.py:1825-1916 (``dpi`` is an extra)."""
    import matplotlib.pyplot as plt
    import matplotlib.ticker
    from matplotlib.animation import FuncAnimation
    xs, ys, v_x, v_y = subsample_velocities_for_visualisation(flow_result, arrow_boxsize=arrow_boxsize)
    dx = flow_result["delta_x"]
    data_panels = [(231, "original_data", "Original data"), (232, "blurred_data", "Blurred")]
    field_panels = [(233, "speed", "viridis", r"Motion speed [$\mathrm{\mu m}$/s]", False),
                    (234, "remodelling", "plasma", "Net remodelling", False),
                    (235, "v_x", "plasma", r"x velocity [$\mathrm{\mu m}$/s]", True),
                    (236, "v_y", "plasma", r"y velocity [$\mathrm{\mu m}$/s]", False)]
    ranges = {key: (np.min(flow_result[key]), np.max(flow_result[key])) for _, key, _, _, _ in field_panels}
    fig = plt.figure(figsize=(6.5, 4.5), constrained_layout=True)

    def animate(i):
        plt.clf()
        for pos, key, title in data_panels:
            plt.subplot(pos)
            costum_imshow(flow_result[key][i], delta_x=dx, cmap=cmap, autoscale=autoscale, v_min=v_min, v_max=v_max)
            _quiver(xs, ys, v_x[i], v_y[i], arrow_color, arrow_scale, arrow_width)
            plt.title(title)
        for pos, key, field_cmap, title, keep_ylabel in field_panels:
            plt.subplot(pos)
            costum_imshow(flow_result[key][i], delta_x=dx, autoscale=True, cmap=field_cmap)
            if not keep_ylabel:
                plt.ylabel("")
            colorbar = plt.colorbar(shrink=0.6)
            plt.clim(*ranges[key])
            colorbar.formatter = matplotlib.ticker.StrMethodFormatter("{x:.2f}")
            plt.title(title)
    FuncAnimation(fig, animate, frames=flow_result["original_data"].shape[0] - 1).save(filename, dpi=dpi)
    plt.close(fig)
