"""Iteration counts (sum over 3 pairs / max) across regimes and sizes for the current environment switches."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
cases = [("N", 1.0, 1.0, 1e4), ("8b5", 255.0, 1e5, 1e3), ("a10", 1.0, 10.0, 1.0), ("a1b1", 1.0, 1.0, 1.0), ("a.5", 1.0, 0.5, 1e3),
         ("a.1", 1.0, 0.1, 1e2), ("8b1e6", 255.0, 1e6, 1e6), ("T", 255.0, 1e4, 1e2), ("W", 255.0, 2e3, 1.0)]
tag = os.environ.get("TAG", "-")
only = os.environ.get("REGIMES")            # e.g. "T,W,a.1": a subset of the regimes
if only:
    cases = [c for c in cases if c[0] in only.split(",")]
big_too = os.environ.get("BIG_TOO") == "1"  # run T / W above 600 pixels as well
for n in [int(v) for v in os.environ.get("SIZES", "66,130,258,514,1026").split(",")]:
    row = []
    for name, scale, al, be in cases:
        if n > 600 and name in ("T", "W") and not big_too:
            continue
        mv = texture_stack_numpy(n, 4, 5) * scale
        r = of.variational_optical_flow(mv, speed_alpha=al, remodelling_alpha=be, return_stats=True, max_iterations=200,
                                        coarse_precision=os.environ.get("COARSE", "float8"), preconditioner="multigrid")
        st = r["stats"]
        row.append(f"{name}:{st['iterations'].sum():3d}/{st['iterations'].max():3d}{'' if st['converged'].all() else '!'}")
    print(f"{tag:8s} n={n:4d} ", "  ".join(row), flush=True)
    of.release_device_memory()
