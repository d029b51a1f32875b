"""Probe of the direct preconditioner on a small case with stage traces (VOF_TRACE=1) and timings."""
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
for n, scale, al, be in ((50, 1.0, 1.0, 1e4), (66, 255.0, 1e4, 1e2), (130, 255.0, 1e4, 1e2), (258, 255.0, 1e4, 1e2)):
    mv = texture_stack_numpy(n, 3, 5) * scale
    for mode in ("direct", "auto"):
        t0 = time.time()
        r = of.variational_optical_flow(mv, speed_alpha=al, remodelling_alpha=be, preconditioner=mode, rtol=1e-9, max_iterations=60, return_stats=True)
        st = r["stats"]
        print(f"n={n} {mode:7s} {time.time() - t0:7.2f} s  iterations {st['iterations'].tolist()}  relres {st['relative_residual'].max():.2e}  converged {st['converged'].tolist()}", flush=True)
