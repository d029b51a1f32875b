#!/usr/bin/env python3
"""Benchmark of the variational optical-flow hot path on MI355X.

One "step" = one pass of the hot path over the whole synthetic stack of this rank: Galerkin hierarchy
setup + batched BiCGStab/multigrid solve of every frame pair to the reference's stopping rule
(rtol 1e-6, OF.py:1120) + epilogue (mirror fix-up, unit scaling, speed, functionals), inputs and
outputs resident in HBM.  For N > 1 each rank owns a contiguous shard of the stack (weak scaling:
--frames per rank) and the flow fields are re-assembled with one RCCL all-gather per field inside the
timed step.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W] [--size 1024 --frames 256]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def algorithmic_bytes_per_pixel(kernel, coarse_bytes):
    """Algorithmic HBM bytes per level-pixel and launch."""
    s = 8
    table = {
        "gs0": 10 * s / 4.0,                      # one colour of the 4-colour sweep: 80 B / 4
        "gs": (81 * coarse_bytes + 9 * s) / 4.0,  # stored stencil: C(81) + b(3) + x(3) read + x(3) write, / 4
        "apply0": 7 * s,                          # I + x(3) in, y(3) out (residual mode adds b(3))
        "residual": 81 * coarse_bytes + 9 * s,
        "rhs": 5 * s,
        "restrict": (3 + 0.75) * s,
        "prolong": (0.75 + 6) * s,
        "galerkin0": 1 * s + 81 * coarse_bytes / 4.0 / 9.0,
        "galerkin": (81 * coarse_bytes + 81 * coarse_bytes / 4.0) / 9.0,
        "vector": 9 * s,
        "reduce": 6 * s,
        "finalize": 7 * s,
        "functionals": 5 * s,
    }
    return table.get(kernel)


def largest_batch(n_pairs, per_pair_bytes, budget_bytes):
    """Pairs in flight: all of them if they fit the memory budget, else the fewest equal-sized batches that do."""
    cap = max(1, int(budget_bytes // per_pair_bytes))
    n_batches = -(-n_pairs // cap)
    return max(1, -(-n_pairs // n_batches))


def cpu_baseline(size, seed):
    """The reference's CPU algorithm (assembly OF.py:833-1072 + its direct-solver branch OF.py:1146-1147)
    as restated by the oracle, timed on this host on a bounded sample."""
    import numpy as np
    from oracle import vof_oracle as orc
    crop = min(size, 256)
    movie = orc.make_texture_stack(size, 5, seed=seed)[:, :crop, :crop]
    t0 = time.time()
    pairs = 0
    while pairs < 4 and (pairs == 0 or time.time() - t0 < 12.0):      # ~10-30 s of CPU work
        orc.variational_optical_flow(np.ascontiguousarray(movie[pairs:pairs + 2]), speed_alpha=1.0,
                                     remodelling_alpha=1e4)
        pairs += 1
    dt = (time.time() - t0) / pairs
    pix_ratio = (size * size) / float(crop * crop)
    return {"value": 1.0 / (dt * pix_ratio), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{pairs} frame pair(s), {crop}x{crop} crop of the workload's first frames: sparse assembly + "
                      f"SuperLU direct solve (the reference's use_direct_solver branch, OF.py:1146-1147) took "
                      f"{dt:.1f} s per pair on 1 core; scaled linearly by pixel count to {size}x{size} "
                      f"(optimistic for the CPU: fill-in is super-linear)",
            "seconds_per_sample_pair": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=256, help="frames per rank (weak scaling)")
    ap.add_argument("--pairs-in-flight", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-6)
    ap.add_argument("--coarse-precision", default="float32", choices=["float64", "float32"])
    ap.add_argument("--vcycle-precision", default="float64", choices=["float64", "float32", "auto"])
    ap.add_argument("--nu-pre", type=int, default=2)
    ap.add_argument("--nu-post", type=int, default=2)
    ap.add_argument("--nu-pre-coarse", type=int, default=1)
    ap.add_argument("--nu-post-coarse", type=int, default=1)
    ap.add_argument("--w-cycle-level", type=int, default=None, help="-1: V-cycle; l: level l visits level l+1 twice")
    ap.add_argument("--w-cycle-visits", type=int, default=None)
    ap.add_argument("--warm-start-stride", type=int, default=None, help="0/1: every pair starts from the constant initial fields (library default: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the informational mixed-precision variant run")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--gather-chunks", type=int, default=0, help="chunks of the stack whose all-gather overlaps the next solve")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even at world size 1 (test)")
    ap.add_argument("--profile-table", action="store_true", help="print the per-kernel HIP-event table (stderr)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner on the
    # first communicator): keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from opticalflow_amd import _native
    from opticalflow_amd.synthetic import texture_stack_torch
    from opticalflow_amd.distributed import allgather_chunk

    n, T = args.size, args.frames
    P = T - 1
    seed = {512: 0, 1024: 1, 2048: 3}.get(n, 1)
    # this rank's shard of the (world * T)-frame stack; frames are generated on the device
    movie = texture_stack_torch(n, T, seed, dev, first_frame=rank * (T - 1))
    vx = torch.empty((P, n, n), dtype=torch.float64, device=dev)
    vy = torch.empty_like(vx)
    gm = torch.empty_like(vx)
    sp = torch.empty_like(vx)
    gathered = None
    n_chunks = 1
    if use_dist and not args.no_allgather:
        # re-assembled stack, natural order: gathered[f][r * P + k] = field f of pair k of rank r
        gathered = [torch.empty((world * P, n, n), dtype=torch.float64, device=dev) for _ in range(3)]
        # the stack is solved in a few chunks so that the all-gather of chunk i (RCCL stream) overlaps the solve of
        # chunk i+1 (solver stream)
        n_chunks = args.gather_chunks if args.gather_chunks > 0 else next((d for d in (3, 4, 5, 2) if P % d == 0), 1)
        if P % n_chunks:
            n_chunks = 1
    torch.cuda.synchronize()

    per_pair = _native.query_workspace(n, n, 1)
    free, total = _native.device_memory(local_rank)
    B = args.pairs_in_flight or largest_batch(P // n_chunks, per_pair, 0.7 * free)
    params = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, rtol=args.rtol,
                                    coarse_precision={"float64": 0, "float32": 1}[args.coarse_precision],
                                    vcycle_precision={"float64": 0, "float32": 1, "auto": 2}[args.vcycle_precision],
                                    nu_pre=args.nu_pre, nu_post=args.nu_post, nu_pre_coarse=args.nu_pre_coarse,
                                    nu_post_coarse=args.nu_post_coarse)
    if args.w_cycle_level is not None:
        params.w_cycle_level = args.w_cycle_level
    if args.w_cycle_visits is not None:
        params.w_cycle_visits = args.w_cycle_visits
    if args.warm_start_stride is not None:
        params.warm_start_stride = args.warm_start_stride
    solver = _native.Solver(n, n, B, device=local_rank)
    coarse_bytes = 8 if args.coarse_precision == "float64" else 4

    def step():
        if gathered is None:
            return solver.solve_dev(movie, T, params, vx, vy, gm, sp, stats=True)   # syncs the solver's stream
        cl = P // n_chunks
        works, stats_all = [], []
        for i in range(n_chunks):
            a, b_ = i * cl, (i + 1) * cl
            stats_all.append(solver.solve_dev(movie[a:b_ + 1], cl + 1, params, vx[a:b_], vy[a:b_], gm[a:b_], sp[a:b_],
                                              stats=True))          # returns after the solver's stream has drained
            for dst, src in zip(gathered, (vx, vy, gm)):
                works.append(allgather_chunk(dst, src, a, b_, P))
        for w in works:
            w.wait()
        return np.concatenate(stats_all)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up with the full per-kernel profile: finds the dominant kernel class
    solver.profile_enable(True)
    solver.profile_filter(-1, -1)
    stats = None
    for _ in range(max(1, args.warmup)):
        stats = step()
    torch.cuda.synchronize()
    table = solver.profile_table()
    total_ms = sum(r[3] for r in table) or 1.0
    # dominant kernel class = largest share of GPU time among the classes with an algorithmic byte count (the streaming
    # kernels); on tiny stacks the latency-bound set-up kernels (dense inversion, Galerkin products) can be larger
    with_bytes = [r for r in table if solver.profile_bytes(r[0], r[1]) > 0]
    dom = max(with_bytes or table, key=lambda r: r[3])
    if args.profile_table and rank == 0:
        for name, lvl, cnt, ms in sorted(table, key=lambda r: -r[3]):
            gb = solver.profile_bytes(name, lvl) / 1e9
            print(f"  {name:13s} L{lvl:<2d} launches {cnt:7d}  total {ms:10.3f} ms  avg {1e3 * ms / cnt:9.2f} us  "
                  f"{100 * ms / total_ms:5.1f}%  {gb / (ms * 1e-3) if gb else 0:8.0f} GB/s", file=sys.stderr)
    dom_name, dom_level = dom[0], dom[1]
    solver.profile_reset()
    solver.profile_filter(dom_name, dom_level)       # timed region: events only around the dominant kernel

    # ---- timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    barrier()
    dt = time.perf_counter() - t0
    cnt, ms = solver.profile_get(dom_name, dom_level)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    pairs_total = world * P * args.steps
    value = pairs_total / dt
    li, lj = solver.level_shape(dom_level)
    units = solver.profile_units(dom_name, dom_level)      # frame pairs actually processed, summed over launches
    pix_per_launch = (units / cnt) * li * lj if cnt else 0.0
    alg_bytes = solver.profile_bytes(dom_name, dom_level)   # exact: summed by the library per launch
    solver.profile_enable(False)
    bpp = algorithmic_bytes_per_pixel(dom_name, coarse_bytes)
    avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
    bytes_per_launch = (alg_bytes / cnt) if (cnt and alg_bytes > 0) else (bpp * pix_per_launch if bpp else None)
    achieved = (bytes_per_launch / avg_s) / 1e9 if (bytes_per_launch and cnt) else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    # the PMC passes were taken with all pairs of the stack in one batch; with chunked solves (multi-GPU overlap) a
    # launch processes fewer pairs and the per-launch traffic figure does not apply
    if os.path.exists(tpath) and B >= P:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"{dom_name}_L{dom_level}_{n}x{n}x{T}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "frame-pairs/sec", "value": value, "unit": "frame-pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n}x{n}x{T} synthetic translating texture per GPU (seed {seed}), speed_alpha=1, "
                               f"remodelling_alpha=1e4, rtol={args.rtol:g}, all {P} pairs solved to the stopping rule",
                   "pairs_in_flight": B, "levels": solver.num_levels, "coarse_stencils": args.coarse_precision, "vcycle_vectors": args.vcycle_precision,
                   "sweeps": [args.nu_pre, args.nu_post, args.nu_pre_coarse, args.nu_post_coarse],
                   "w_cycle_level": int(params.w_cycle_level), "w_cycle_visits": int(params.w_cycle_visits),
                   "warm_start_stride": int(params.warm_start_stride),
                   "allgather": gathered is not None, "gather_chunks": n_chunks,
                   "iterations_max": int(stats["iterations"].max()), "iterations_mean": float(stats["iterations"].mean()),
                   "relres_max": float(stats["relative_residual"].max()),
                   "converged": bool(stats["converged"].all())},
        "roofline": {"bound": "hbm", "kernel": f"{dom_name}@L{dom_level}", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                     "launches": cnt, "pairs_per_launch": (units / cnt) if cnt else None, "avg_launch_us": 1e6 * avg_s if cnt else None,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "share_of_gpu_time": dom[3] / total_ms},
    }
    if world == 1 and not use_dist and not args.no_variants:
        # informational, not the headline: the same step with other storage precisions inside the preconditioner
        # (headline = library defaults: float32 Galerkin stencils, float64 V-cycle vectors).  Arithmetic, Krylov
        # vectors, stopping rule and results are float64 in all of them.
        def timed(pv):
            solver.solve_dev(movie, T, pv, vx, vy, gm, sp, stats=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stv = solver.solve_dev(movie, T, pv, vx, vy, gm, sp, stats=True)
            torch.cuda.synchronize()
            d = time.perf_counter() - t1
            return {"value": P / d, "unit": "frame-pairs/s", "iterations_mean": float(stv["iterations"].mean()),
                    "relres_max": float(stv["relative_residual"].max()), "converged": bool(stv["converged"].all())}
        common = dict(speed_alpha=1.0, remodelling_alpha=1e4, rtol=args.rtol, nu_pre=args.nu_pre, nu_post=args.nu_post,
                      nu_pre_coarse=args.nu_pre_coarse, nu_post_coarse=args.nu_post_coarse,
                      w_cycle_level=int(params.w_cycle_level), w_cycle_visits=int(params.w_cycle_visits),
                      warm_start_stride=int(params.warm_start_stride))
        cold = dict(common, warm_start_stride=0)
        out["variants"] = {
            "constant_initial_fields_for_every_pair": timed(_native.default_params(vcycle_precision=0, coarse_precision=1, **cold)),
            "all_float64_storage": timed(_native.default_params(vcycle_precision=0, coarse_precision=0, **common)),
            "float32_stencils_auto_vectors": timed(_native.default_params(vcycle_precision=2, coarse_precision=1, **common)),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, seed)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    solver.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
