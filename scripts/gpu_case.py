"""Re-run one case of tests/test_gpu_parity.py::test_seeded_random_configurations_against_oracle with several solver options."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
case = int(sys.argv[1])
rng = np.random.default_rng(1000 + case)
n_i, n_j = int(rng.integers(4, 72)), int(rng.integers(4, 72))
T = int(rng.integers(2, 5))
scale = [1.0, 1.0, 255.0][case % 3]
movie = texture_stack_numpy(max(n_i, n_j, 16), T, seed=case)[:, :n_i, :n_j] * scale
if case % 4 == 0:
    movie = movie + 0.02 * scale * rng.random(movie.shape)
alpha = float(10 ** rng.uniform(-0.5, 2.0)) * scale ** 2
beta = float(10 ** rng.uniform(0.0, 4.0))
kw = dict(speed_alpha=alpha, remodelling_alpha=beta, delta_x=float(rng.uniform(0.2, 2.0)), delta_t=float(rng.uniform(0.5, 2.0)),
          initial_v_x=float(rng.uniform(-0.5, 0.5)), initial_v_y=float(rng.uniform(-0.5, 0.5)),
          initial_remodelling=float(rng.uniform(-0.1, 0.1)), reference_quirks=bool(case % 5 != 0))
print(f"case {case}: {n_i}x{n_j}x{T} scale {scale} alpha {alpha:.3g} beta {beta:.3g} {kw}")
base = dict(coarse_precision=["float32", "float64"][case % 2], vcycle_precision=["float64", "float32", "auto"][(case // 2) % 3],
            w_cycle_level=[None, -1, 0, (1, 2)][case % 4], multigrid_sweeps=[None, (1, 1), (2, 1, 2, 2), (3, 3)][(case // 3) % 4],
            max_pairs_in_flight=[None, 1, 2][case % 3])
print("options of the case:", base)
for label, extra in (("gmres", dict(krylov_method="gmres")), ("gmres restart 128", dict(krylov_method="gmres", gmres_restart=128)),
                     ("bicgstab", dict(krylov_method="bicgstab")), ("auto", dict(krylov_method="auto")),
                     ("gmres, all float64", dict(krylov_method="gmres", coarse_precision="float64", vcycle_precision="float64")),
                     ("gmres, default cycle", dict(krylov_method="gmres", multigrid_sweeps=None)),
                     ("gmres, one batch", dict(krylov_method="gmres", max_pairs_in_flight=None)),
                     ("gmres, rtol 1e-9", dict(krylov_method="gmres", rtol=1e-9))):
    o = dict(base); o.update(extra)
    rtol = o.pop("rtol", 1e-10)
    r = of.variational_optical_flow(movie, rtol=rtol, return_stats=True, max_iterations=400, **kw, **o)
    st = r["stats"]
    print(f"  {label}: iterations {st['iterations'].tolist()} converged {st['converged'].tolist()} relres {['%.2e' % v for v in st['relative_residual']]}")
