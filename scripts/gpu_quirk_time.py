"""Timing of the consistent-derivative option (reference_quirks=False) against the default, same stack.  GPU box only."""
import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
base = texture_stack_numpy(1024, 9, seed=1)
movie = np.concatenate([base] * 15)[:129].copy()
for rep in range(2):
    for quirks in (True, False):
        t0 = time.time()
        r = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, reference_quirks=quirks, return_stats=True)
        dt = time.time() - t0
        st = r["stats"]
        print(f"quirks={quirks}: {dt:.2f} s, iterations {st['iterations'].min()}-{st['iterations'].max()}, converged {bool(st['converged'].all())}", flush=True)
        del r; gc.collect()
