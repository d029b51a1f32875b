"""Iteration counts (max over 3 pairs) of the default solver settings across regimes and sizes; used to compare hierarchy
depths (VOF_COARSEST_MAX) and other global switches.  usage: [ENV=..] python scripts/gpu_regimes2.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy

cases = [("N [0,1] a=1 b=1e4", 1.0, 1.0, 1e4), ("8bit a=1e5 b=1e3", 255.0, 1e5, 1e3), ("[0,1] a=10 b=1", 1.0, 10.0, 1.0),
         ("[0,1] a=1 b=1", 1.0, 1.0, 1.0), ("[0,1] a=0.5 b=1e3", 1.0, 0.5, 1e3), ("T 8bit a=1e4 b=1e2", 255.0, 1e4, 1e2),
         ("W 8bit a=2e3 b=1", 255.0, 2e3, 1.0)]
tag = os.environ.get("VOF_COARSEST_MAX", "-")
for n in (50, 66, 130, 258, 514):
    row = []
    for name, scale, al, be in cases:
        mv = texture_stack_numpy(n, 4, 5) * scale
        t = time.time()
        r = of.variational_optical_flow(mv, speed_alpha=al, remodelling_alpha=be, return_stats=True, max_iterations=300)
        st = r["stats"]
        row.append(f"{name.split()[0]}:{st['iterations'].max():3d}{'' if st['converged'].all() else '!'}")
    print(f"cmax={tag} n={n:4d} ", "  ".join(row), flush=True)
    of.release_device_memory()
