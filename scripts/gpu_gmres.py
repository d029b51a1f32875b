"""Krylov fallback experiment: BiCGStab only vs auto (BiCGStab -> GMRES) vs GMRES only, hard regimes.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy

def run(n, regime, blur, T=3, **kw):
    if regime == "T": scale, a, b = 255.0, 1e4, 1e2
    elif regime == "W": scale, a, b = 255.0, 2e3, 1.0
    else: scale, a, b = 1.0, 1.0, 1e4
    mv = texture_stack_numpy(n, T, seed=1)
    if scale > 1: mv = np.round(mv * scale)
    for method, extra in (("bicgstab", {}), ("auto", {}), ("gmres", {}), ("gmres", dict(gmres_restart=100))):
        t0 = time.time()
        r = of.variational_optical_flow(mv, speed_alpha=a, remodelling_alpha=b, smoothing_sigma=blur, max_iterations=400,
                                        krylov_method=method, return_stats=True, max_pairs_in_flight=T - 1, **extra, **kw)
        st = r["stats"]
        print(f"n={n} {regime} blur={blur} {method}{extra}: iterations {st['iterations'].tolist()} converged "
              f"{st['converged'].tolist()} relres {['%.1e' % v for v in st['relative_residual']]} time {time.time()-t0:.2f}s", flush=True)

for n in (int(a) for a in sys.argv[2:]):
    for blur in (None, 2.0):
        run(n, sys.argv[1], blur)
