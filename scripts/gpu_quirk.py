"""Hard regimes with and without the reference's 'dy' == 'dx' quirk (OF.py:698-699).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
for n in (258, 514, 1026):
    mv = np.round(texture_stack_numpy(n, 3, seed=1) * 255.0)
    for regime, a, b in (("T", 1e4, 1e2), ("W", 2e3, 1.0)):
        for blur in (None, 2.0):
            for quirks in (True, False):
                t0 = time.time()
                r = of.variational_optical_flow(mv, speed_alpha=a, remodelling_alpha=b, smoothing_sigma=blur, max_iterations=300,
                                                reference_quirks=quirks, return_stats=True)
                st = r["stats"]
                print(f"n={n} {regime} blur={blur} quirks={quirks}: iterations {st['iterations'].tolist()} converged "
                      f"{st['converged'].tolist()} relres {['%.1e' % v for v in st['relative_residual']]} "
                      f"mean v=({r['v_x'].mean():.3f},{r['v_y'].mean():.3f}) time {time.time()-t0:.2f}s", flush=True)
