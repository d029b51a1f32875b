"""Drop-in call (1024^2 x 256, numpy in, numpy out) with the stack solved in one batch against several batches (the
host entry point overlaps the copies of a batch with the solve of the next one).  GPU box only."""
import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
n, T = 1024, 256
base = texture_stack_numpy(n, 9, seed=1)
movie = np.concatenate([base] * 29)[:T].copy()
for B in (None, 128, 96, 64, None):
    for rep in range(2):
        t0 = time.time()
        r = of.variational_optical_flow(movie, remodelling_alpha=1e4, max_pairs_in_flight=B)
        dt = time.time() - t0
        print(f"max_pairs_in_flight {B}: call {rep}: {dt:.3f} s = {(T-1)/dt:.0f} pairs/s, converged {r['converged']}", flush=True)
        del r; gc.collect()
