"""Phase timing of the drop-in host call (1024^2 x 256).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import _native, optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
n, T = 1024, 256
base = texture_stack_numpy(n, 9, seed=1)
movie = np.concatenate([base] * 29)[:T].copy()
p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4)
def lap(label, t0):
    t1 = time.time(); print(f"  {label}: {t1 - t0:.3f} s", flush=True); return t1
for rep in range(4):
    print("rep", rep)
    t = time.time()
    m64 = np.asarray(movie).astype(np.float64); t = lap("astype float64 (2 GB copy)", t)
    B = of.choose_pairs_in_flight(n, n, T - 1, 0); t = lap(f"choose_pairs_in_flight -> {B}", t)
    s = _native.Solver(n, n, B); t = lap("vof_create", t)
    out = s.solve_host(m64, p); t = lap("solve_host (np.empty + C call)", t)
    s.close(); t = lap("vof_destroy", t)
    del out, m64; t = lap("free results", t)
