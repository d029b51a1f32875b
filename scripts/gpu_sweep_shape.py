"""Access-pattern probe of the level-0 smoother: the same number of pixels per pair as one strip-wide, very tall image
(every workgroup then streams whole, consecutive rows) against the square image (1-KB row segments 8 KB apart).
usage: python scripts/gpu_sweep_shape.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from opticalflow_amd import _native

P = int(sys.argv[1]) if len(sys.argv) > 1 else 255
for Ni, Nj in ((1024, 1024), (8194, 114), (4098, 226), (2050, 450)):
    s = _native.Solver(Ni, Nj, P)
    g = torch.Generator(device="cuda").manual_seed(1)
    movie = torch.rand((P + 1, Ni, Nj), dtype=torch.float64, device="cuda", generator=g)
    prm = _native.default_params(remodelling_alpha=1e4)
    s.bench_sweeps(movie, P, prm, 4)
    s.profile_enable(True)
    s.profile_reset()
    s.bench_sweeps(movie, P, prm, 4)
    cnt, ms = s.profile_get("gs0", 0)
    mv = s.profile_moved("gs0", 0) if hasattr(s, "profile_moved") else 0.0
    px = (Ni - 2) * (Nj - 2) * P
    print(f"{Ni}x{Nj} pairs {P}: {cnt} passes {ms:8.2f} ms = {1e6 * ms / 4 / px * 1e3:7.3f} ps per pixel-sweep")
    s.close(); del movie
    torch.cuda.empty_cache()
