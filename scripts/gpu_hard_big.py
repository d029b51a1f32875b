"""Hard regime at benchmark scale: 1024^2 x 129 8-bit stack, alpha 1e4 (no convergence expected): the GMRES fallback must
allocate its (memory-capped) basis, run and report - no crash, bounded time.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
base = np.round(texture_stack_numpy(1024, 9, seed=1) * 255.0)
movie = np.concatenate([base] * 15)[:129].copy()
for kw in (dict(max_iterations=60), dict(max_iterations=60, reference_quirks=False)):
    t0 = time.time()
    r = of.variational_optical_flow(movie, speed_alpha=1e4, remodelling_alpha=1e2, return_stats=True, **kw)
    st = r["stats"]
    print(f"{kw}: {time.time()-t0:.2f} s, iterations min/max {st['iterations'].min()}/{st['iterations'].max()}, converged {int(st['converged'].sum())}/128, "
          f"relres max {st['relative_residual'].max():.2e}", flush=True)
