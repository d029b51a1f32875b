"""Time vary_regularisation: native device sweep vs. the reference's structure (one host-level solve per combination,
numpy statistics on the host).  GPU box only."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy

n, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 17)
km = sys.argv[3] if len(sys.argv) > 3 else "auto"
if ":" in km: km = (km.split(":")[0], int(km.split(":")[1]))
movie = texture_stack_numpy(n, T, seed=1)
sa, ra = np.array([0.5, 1.0, 2.0]), np.array([3e3, 1e4, 3e4])
of.vary_regularisation(movie[:3], sa[:1], ra[:1], krylov_method=km)          # warm-up
t0 = time.time(); r = of.vary_regularisation(movie, sa, ra, return_stats=True, krylov_method=km); t_native = time.time() - t0
t0 = time.time()
loop = np.zeros((3, 3))
for i, a in enumerate(sa):
    for j, b in enumerate(ra):
        one = of.variational_optical_flow(movie, speed_alpha=a, remodelling_alpha=b, krylov_method=km)
        loop[i, j] = np.var(one["speed"]); np.mean(one["speed"]); np.mean(one["remodelling"]); np.var(one["remodelling"])
t_loop = time.time() - t0
print(f"{n}x{n}x{T} {km}, 9 combinations: native sweep {t_native:.3f} s, per-combination host loop {t_loop:.3f} s, "
      f"ratio {t_loop / t_native:.2f}; max rel diff of speed variance {np.max(np.abs(loop / r['speed_variances'] - 1)):.2e}; "
      f"max iterations {r['stats']['max_iterations_used'].max()}")
