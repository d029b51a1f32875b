"""Hard regime T (8-bit data, alpha 1e4, beta 1e2, reference quirks on): do more coarse-level sweeps help?  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
for n in (258, 514, 1026):
    mv = np.round(texture_stack_numpy(n, 3, seed=1) * 255.0)
    for blur in (None, 2.0):
        for sweeps, w in (((2, 2, 1, 1), None), ((2, 2, 4, 4), None), ((2, 2, 8, 8), None), ((4, 4, 8, 8), None), ((2, 2, 4, 4), -1), ((2, 2, 8, 8), -1), ((2, 2, 8, 8), (1, 2))):
            t0 = time.time()
            r = of.variational_optical_flow(mv, speed_alpha=1e4, remodelling_alpha=1e2, smoothing_sigma=blur, max_iterations=300,
                                            multigrid_sweeps=sweeps, w_cycle_level=w, return_stats=True)
            st = r["stats"]
            print(f"n={n} blur={blur} sweeps={sweeps} w={w}: iterations {st['iterations'].tolist()} converged {st['converged'].tolist()} "
                  f"relres {['%.1e' % v for v in st['relative_residual']]} time {time.time()-t0:.2f}s", flush=True)
