"""Seeded sweep at large image sizes (400-1150 pixels a side) over solver options: stopping rule met (library's independent
residual) and fields equal to a tight BiCGStab solve.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    rng = np.random.default_rng(7000 + case)
    n_i, n_j, T = int(rng.integers(400, 1150)), int(rng.integers(400, 1150)), int(rng.integers(2, 7))
    movie = texture_stack_numpy(max(n_i, n_j), T, seed=200 + case)[:, :n_i, :n_j]
    alpha, beta = float(10 ** rng.uniform(-0.2, 1.0)), float(10 ** rng.uniform(2.0, 4.0))
    opts = dict(krylov_method=["auto", "gmres"][case % 2], vcycle_precision=["coarse_float32", "float64", "float32", "auto"][case % 4],
                w_cycle_level=[None, -1, (1, 2), 2][case % 4], max_pairs_in_flight=[None, 2, 3][case % 3])
    t0 = time.time()
    res = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, rtol=1e-7, return_stats=True, **opts)
    dt = time.time() - t0
    st = res["stats"]
    # reference for the fields: a second solve to a much tighter tolerance with the other Krylov method
    ref = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, rtol=1e-10, krylov_method="bicgstab")
    err = max(np.linalg.norm(res[k] - ref[k]) / np.linalg.norm(ref[k]) for k in ("v_x", "v_y"))
    ok = st["converged"].all() and st["relative_residual"].max() <= 1.6e-7 and err < 1e-4
    bad += not ok
    print(f"case {case}: {n_i}x{n_j}x{T} alpha {alpha:.2g} beta {beta:.2g} {opts} -> iterations {st['iterations'].tolist()} "
          f"relres {st['relative_residual'].max():.1e} err vs tight {err:.1e} {'OK' if ok else 'FAIL'} ({dt:.2f}s)", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
