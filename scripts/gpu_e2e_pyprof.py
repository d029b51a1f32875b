"""Where does the Python side of the drop-in host call spend its time? (cProfile of the third call at 1024^2 x 256)"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
n, T = 1024, 256
base = texture_stack_numpy(n, 17, seed=1)
movie = np.concatenate([base] * 16)[:T].copy()
for rep in range(2):
    r = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, return_stats=True); del r
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
r = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, return_stats=True)
pr.disable()
print(f"call: {time.time() - t0:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
