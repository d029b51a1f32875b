"""Drop-in for the hot path of the reference module ``source/optical_flow.py``.

``variational_optical_flow`` keeps the reference's signature and result dictionary
(source/optical_flow.py:715-724, 1193-1205) so that ``analysis/analyse_variational_optical_flow.py``
can call it unchanged; the per-pair scipy.sparse assembly + PETSc KSP solve (OF.py:833-1145) is
replaced by the batched HIP solver in ``csrc/`` (BiCGStab + geometric multigrid).
The solve always runs on an MI355X through libvof.so; there is no CPU path in this module.
"""
from __future__ import annotations

import time

import numpy as np

from . import _native

__all__ = ["variational_optical_flow", "vary_regularisation", "make_fake_data_frame", "blur_movie",
           "format_elapsed_time", "apply_constant_boundary_condition", "choose_pairs_in_flight"]


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
    with _native.Solver(max(4, movie.shape[1]), max(4, movie.shape[2]), 1, device=device) as solver:
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


def choose_pairs_in_flight(n_i, n_j, n_pairs, device=0, memory_fraction=0.6, cap=96):
    """Largest batch of frame pairs whose workspace fits in ``memory_fraction`` of the free HBM."""
    free, _total = _native.device_memory(device)
    budget = free * memory_fraction
    per_pair = _native.query_workspace(n_i, n_j, 1) + 5 * n_i * n_j * 8  # + host-API staging
    return int(max(1, min(n_pairs, cap, budget // max(per_pair, 1))))


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
                             coarse_precision="float32",
                             vcycle_precision="float64",
                             multigrid_sweeps=None,
                             w_cycle_level=None,
                             verbose=False,
                             return_stats=False,
                             _solver=None):
    """Variational optical flow with remodelling on an image stack, on one MI355X.

    Positional/keyword arguments up to ``use_direct_solver`` have the reference's meaning
    (OF.py:725-762).  ``movie`` is ``(T, N_i, N_j)``, any real dtype; the result ``k`` is the flow
    from frame ``k`` to ``k+1``.  Returns the reference's result dict (OF.py:1193-1205): ``v_x``,
    ``v_y``, ``speed``, ``remodelling`` (float64 ``(T-1, N_i, N_j)``), ``original_data``,
    ``blurred_data``, ``delta_x``, ``delta_t``, ``converged`` (flag of the LAST pair, as in the
    reference), ``L1_functional``, ``remodelling_functional``, ``speed_functional``.

    Differences, all opt-in or invisible at the reference's tolerance:
      * all pairs are solved concurrently from the same constant initial guess instead of
        warm-starting pair k from pair k-1 (OF.py:803-806); the converged answer is the same to
        solver tolerance;
      * ``use_direct_solver=True`` (SuperLU in the reference, OF.py:1146-1147) is honoured as "solve
        to rtol=1e-11" on the GPU;
      * keyword-only extras: ``rtol`` (default 1e-6 = OF.py:1120), ``max_iterations`` (1000),
        ``reference_quirks`` (True keeps OF.py:698-699 'dy'=='dx' and the OF.py:1205
        ``speed_functional`` assignment), ``device``, ``max_pairs_in_flight``, ``coarse_precision`` /
        ``vcycle_precision`` (storage precision inside the multigrid preconditioner only: "float64", "float32" or
        "auto" = float32 for the first 8 iterations, float64 for stragglers; all arithmetic, the Krylov iteration, the
        stopping rule and the result are float64 either way), ``multigrid_sweeps``
        (block-GS sweeps per V-cycle: ``(pre, post)`` on level 0 and optionally ``(pre, post)`` on the coarse levels),
        ``w_cycle_level`` (-1: V-cycle; ``l``: level ``l`` visits level ``l+1`` twice per cycle),
        ``verbose``, ``return_stats`` (adds ``result['stats']``: per-pair iterations / residual /
        converged / functionals).
    """
    movie = np.asarray(movie).astype(np.float64)                       # OF.py:769
    if movie.ndim != 3:
        raise ValueError("movie must be a 3-D array (frames, x, y)")
    if smoothing_sigma is not None:                                     # OF.py:770-773
        movie_to_analyse = blur_movie(movie, smoothing_sigma=smoothing_sigma, device=device, _solver=_solver)
    else:
        movie_to_analyse = movie
    T, N_i, N_j = movie.shape
    if T < 2:
        raise ValueError("movie needs at least two frames")
    if rtol is None:
        rtol = 1e-11 if use_direct_solver else 1e-6
    params = _native.default_params(
        speed_alpha=float(speed_alpha), remodelling_alpha=float(remodelling_alpha), delta_x=float(delta_x),
        delta_t=float(delta_t), initial_v_x=float(initial_v_x), initial_v_y=float(initial_v_y),
        initial_remodelling=float(initial_remodelling), rtol=float(rtol), max_iterations=int(max_iterations),
        reference_quirks=int(bool(reference_quirks)),
        coarse_precision={"float64": 0, "float32": 1}[coarse_precision],
        vcycle_precision={"float64": 0, "float32": 1, "auto": 2}[vcycle_precision])
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
    if max_pairs_in_flight is None and _solver is None:
        max_pairs_in_flight = choose_pairs_in_flight(N_i, N_j, T - 1, device)
    t0 = time.time()
    if _solver is not None:      # a caller-owned context (vary_regularisation re-uses one workspace for all solves)
        v_x, v_y, remodelling, speed, stats = _solver.solve_host(np.ascontiguousarray(movie_to_analyse), params)
    else:
        with _native.Solver(N_i, N_j, max_pairs_in_flight, device=device) as solver:
            v_x, v_y, remodelling, speed, stats = solver.solve_host(np.ascontiguousarray(movie_to_analyse), params)
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


def vary_regularisation(movie,
                        speed_alpha_values=np.arange(500, 2000, 500),
                        remodelling_alpha_values=np.arange(500, 2000, 500),
                        filename=None,
                        **kwargs):
    """Vary both regularisation parameters and keep the summary statistics for heat-maps; same arguments,
    result dictionary and optional ``np.save`` as the reference (OF.py:1918-1998).  Every
    ``(speed_alpha, remodelling_alpha)`` combination is an independent solve of the same movie; one device
    workspace is created once and re-used for all of them, each solve batches all frame pairs.

    Returns a dict with ``speed_alpha_values``, ``remodelling_alpha_values``, ``speed_means``,
    ``speed_variances``, ``remodelling_means``, ``remodelling_variances``, ``converged`` and ``functional``
    (``L1_functional + speed_functional + remodelling_functional``, OF.py:1983), each of shape
    ``(len(speed_alpha_values), len(remodelling_alpha_values))``.
    """
    movie = np.asarray(movie)
    if movie.ndim != 3:
        raise ValueError("movie must be a 3-D array (frames, x, y)")
    shape = (len(speed_alpha_values), len(remodelling_alpha_values))
    speed_means = np.zeros(shape)
    speed_variances = np.zeros_like(speed_means)
    remodelling_means = np.zeros_like(speed_means)
    remodelling_variances = np.zeros_like(speed_means)
    converged = np.zeros_like(speed_means, dtype=bool)
    total_variations = np.zeros_like(speed_means)
    T, N_i, N_j = movie.shape
    device = kwargs.get("device", 0)
    pairs = kwargs.pop("max_pairs_in_flight", None) or choose_pairs_in_flight(N_i, N_j, T - 1, device)
    with _native.Solver(N_i, N_j, pairs, device=device) as solver:
        for i, speed_alpha in enumerate(speed_alpha_values):
            for j, remodelling_alpha in enumerate(remodelling_alpha_values):
                result = variational_optical_flow(movie, speed_alpha=speed_alpha, remodelling_alpha=remodelling_alpha,
                                                  _solver=solver, **kwargs)
                speed_means[i, j] = np.mean(result["speed"])
                speed_variances[i, j] = np.var(result["speed"])
                remodelling_means[i, j] = np.mean(result["remodelling"])
                remodelling_variances[i, j] = np.var(result["remodelling"])
                converged[i, j] = result["converged"]
                total_variations[i, j] = (result["L1_functional"] + result["speed_functional"]
                                          + result["remodelling_functional"])
    result_dict = {}
    result_dict["speed_alpha_values"] = speed_alpha_values
    result_dict["remodelling_alpha_values"] = remodelling_alpha_values
    result_dict["speed_means"] = speed_means
    result_dict["speed_variances"] = speed_variances
    result_dict["remodelling_means"] = remodelling_means
    result_dict["remodelling_variances"] = remodelling_variances
    result_dict["converged"] = converged
    result_dict["functional"] = total_variations
    if filename is not None:
        np.save(filename, result_dict)
    return result_dict
