"""Fixed-work timing of the level-0 smoother (vof_bench_sweeps_dev): n sweeps on P pairs, HIP-event time per launch.
usage: python scripts/gpu_sweep_micro.py [pairs] [sweeps] ; VOF_LIB selects an alternative build, VOF_SWEEP0M the kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from opticalflow_amd import _native
from opticalflow_amd.synthetic import texture_stack_torch

P = int(sys.argv[1]) if len(sys.argv) > 1 else 255
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(os.environ.get("VOF_N", "1024"))
s = _native.Solver(n, n, P)
movie = texture_stack_torch(n, P + 1, 1, torch.device("cuda", 0), solver=s)
prm = _native.default_params(remodelling_alpha=1e4, vcycle_precision=int(os.environ.get("VOF_VP", "3")))
s.bench_sweeps(movie, P, prm, NS)
s.profile_enable(True)
for rep in range(3):
    s.profile_reset()
    s.bench_sweeps(movie, P, prm, NS)
    cnt, ms = s.profile_get("gs0", 0)
    by = s.profile_bytes("gs0", 0)
    print(f"{os.environ.get('VOF_LIB', 'default')[-24:]:24s} pairs {P} sweeps {NS}: {cnt} passes, {ms:8.2f} ms total = {1e3 * ms / NS:8.1f} us per sweep, "
          f"{by / ms / 1e6:7.0f} GB/s on the bytes actually needed (the first sweep starts from zero)")
s.close()
