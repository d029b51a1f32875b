"""Phase trace of the drop-in host call at 1024^2 x 256 (VOF_TRACE_HOST=1 prints the native side's timeline)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["VOF_TRACE_HOST"] = "1"
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
n, T = 1024, 256
base = texture_stack_numpy(n, 17, seed=1)
movie = np.concatenate([base] * 16)[:T].copy()
for rep in range(int(os.environ.get("E2E_CALLS", "3"))):
    t0 = time.time()
    r = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, return_stats=True)
    dt = time.time() - t0
    print(f"call {rep}: {dt:.3f} s = {(T - 1) / dt:.0f} pairs/s, converged {bool(r['stats']['converged'].all())}", flush=True)
    del r
