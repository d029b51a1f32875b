#!/usr/bin/env python3
"""Benchmark of the variational optical-flow hot path on MI355X.

One "step" = one pass of the hot path over the whole synthetic stack of this rank: Galerkin hierarchy
setup + batched BiCGStab/multigrid solve of every frame pair to the reference's stopping rule
(rtol 1e-6, OF.py:1120) + epilogue (mirror fix-up, unit scaling, speed, functionals), inputs and
outputs resident in HBM.  For N > 1 each rank owns a contiguous shard of the stack (weak scaling:
--frames per rank) and the flow fields are re-assembled with one RCCL all-gather per field inside the
timed step.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W] [--size 1024 --frames 256]

`python bench.py --gpus N` (N > 1) without a torch.distributed environment starts the N ranks itself (fresh child
processes through torch.distributed.run, before anything touches the GPU) and relays rank 0's line.
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


def _cpu_pair(args):
    """One frame pair through the oracle's assembly + direct solve (worker of the all-cores baseline)."""
    size, seed, crop, k = args
    import numpy as np
    from oracle import vof_oracle as orc
    movie = orc.make_texture_stack(size, 2, seed=seed, first_frame=k)[:, :crop, :crop]
    t0 = time.time()
    orc.variational_optical_flow(np.ascontiguousarray(movie), speed_alpha=1.0, remodelling_alpha=1e4)
    return time.time() - t0


def cpu_baseline(size, seed):
    """The reference's CPU algorithm (assembly OF.py:833-1072 + its direct-solver branch OF.py:1146-1147) as restated by
    the oracle, timed on this host on a bounded sample: first on ONE core (the reference is a single sequential process,
    OF.py:1098 PETSc.COMM_SELF), then one pair per core on all the cores this process may use."""
    import multiprocessing as mp
    crop = min(size, 256)
    pix_ratio = (size * size) / float(crop * crop)
    t0 = time.time()
    pairs = 0
    while pairs < 2 and (pairs == 0 or time.time() - t0 < 8.0):
        _cpu_pair((size, seed, crop, pairs))
        pairs += 1
    dt = (time.time() - t0) / pairs
    host_cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cores
    cap = int(os.environ.get("VOF_BENCH_CPU_WORKERS", "16"))   # the GPU box gives one GPU's job a 16-core share (its affinity
    workers = max(1, min(usable, cap))                         # mask still lists every core of the host)
    all_cores = None
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            t1 = time.time()
            per = pool.map(_cpu_pair, [(size, seed, crop, k) for k in range(workers)])
            wall = time.time() - t1
        all_cores = {"value": workers / (wall * pix_ratio), "unit": "frame-pairs/s", "cores": workers,
                     "cores_cap": f"{cap} worker processes = the CPU share of a one-GPU job on the box; the affinity mask "
                                  f"({usable} cores) is the whole host's (VOF_BENCH_CPU_WORKERS overrides)",
                     "seconds_wall": wall, "seconds_per_pair_mean": float(sum(per) / len(per)),
                     "sample": f"{workers} pairs of the same {crop}x{crop} crop, one per process"}
    except Exception as exc:      # noqa: BLE001 - the single-core figure stands on its own
        all_cores = {"error": repr(exc)}
    return {"value": 1.0 / (dt * pix_ratio), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "host_cores": host_cores, "usable_cores": usable,
            "sample": f"{pairs} frame pair(s), {crop}x{crop} crop of the workload's first frames: sparse assembly + "
                      f"SuperLU direct solve (the reference's use_direct_solver branch, OF.py:1146-1147; PETSc is not "
                      f"installable on the box) took {dt:.1f} s per pair on 1 core; scaled linearly by pixel count to "
                      f"{size}x{size} (optimistic for the CPU: fill-in is super-linear)",
            "seconds_per_sample_pair": dt, "all_cores": all_cores}


def selftest_launch(real_stdout):
    """CPU check of the launcher path (tests/test_distributed_cpu.py): every rank joins a gloo group, the ranks agree on a
    sum, rank 0 prints one JSON line.  Nothing here touches a GPU."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        os.write(real_stdout, (json.dumps({"selftest": "launch", "n_gpus": world, "sum": float(t.item())}) + "\n").encode())
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per rank (weak scaling).  Default: 256 on one GPU (BASELINE configs[2]: 1024x1024x256), "
                         "129 per rank on several (configs[3]: 1024 pairs of 1024x1024 over 8 GPUs = 128 pairs per rank)")
    ap.add_argument("--pairs-in-flight", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-6)
    ap.add_argument("--coarse-precision", default="float8", choices=["float64", "float32", "bfloat16", "float8"])
    ap.add_argument("--vcycle-precision", default="coarse_float32", choices=["float64", "float32", "auto", "coarse_float32"])
    ap.add_argument("--nu-pre", type=int, default=2)
    ap.add_argument("--nu-post", type=int, default=2)
    ap.add_argument("--nu-pre-coarse", type=int, default=1)
    ap.add_argument("--nu-post-coarse", type=int, default=1)
    ap.add_argument("--w-cycle-level", type=int, default=None, help="-1: V-cycle; l: level l visits level l+1 several times")
    ap.add_argument("--w-cycle-visits", type=int, default=None)
    ap.add_argument("--warm-start-stride", type=int, default=None, help="0/1: every pair starts from the constant initial fields (library default: 3)")
    ap.add_argument("--krylov-method", type=int, default=None, help="0 BiCGStab only, 1 restarted GMRES only, 2 (library default) BiCGStab with GMRES fallback")
    ap.add_argument("--wobble", type=float, default=0.0, help="time-varying flow of the synthetic stack (0: uniform translation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the informational variant runs")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--gather-chunks", type=int, default=0, help="equal chunks of the rank's pairs whose all-gather overlaps the next solve (default: distributed.chunk_plan)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even at world size 1 (test)")
    ap.add_argument("--profile-table", action="store_true", help="print the per-kernel HIP-event table (stderr)")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` started by hand: become the launcher of N fresh rank processes.  Nothing has
    # touched the GPU (torch is not even imported) - the parent only relays rank 0's line and the exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from opticalflow_amd.distributed import launch_ranks
        raise SystemExit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL and gloo print banners on the
    # first communicator): keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.selftest_launch:
        return selftest_launch(real_stdout)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py --gpus N` or "
                         f"under torch.distributed.run with --nproc-per-node N)")
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
    from opticalflow_amd.distributed import allgather_chunk, block_cyclic_range, chunk_plan

    n = args.size
    T = args.frames or (256 if world == 1 else 129)
    P = T - 1
    seed = {512: 0, 1024: 1 if world == 1 else 2, 2048: 3}.get(n, 1)     # C2, C3 / C4, C5
    gather = use_dist and not args.no_allgather
    # Several ranks: the rank's P pairs are cut into chunks dealt block-cyclically over the global stack, so that the
    # all-gather of chunk i lands in place (natural order) while chunk i+1 is solved; one rank: one chunk = the stack.
    sizes = chunk_plan(P, args.gather_chunks) if gather else [P]
    n_chunks = len(sizes)
    cp_arg = {"float64": 0, "float32": 1, "bfloat16": 2, "float8": 3}[args.coarse_precision]
    vp_arg = {"float64": 0, "float32": 1, "auto": 2, "coarse_float32": 3}[args.vcycle_precision]
    # the variant runs below widen the stencil storage of the same context up to float64 (the context re-allocates it)
    per_pair = _native.query_workspace(n, n, 1, cp_arg if (args.no_variants or world > 1 or use_dist) else 0, vp_arg)
    free, total = _native.device_memory(local_rank)
    gathered_bytes = 3 * world * P * n * n * 8 if gather else 0
    B = args.pairs_in_flight or largest_batch(max(sizes), per_pair, 0.7 * free - gathered_bytes)
    solver = _native.Solver(n, n, B, device=local_rank)
    # frames of every chunk (its pairs + one overlap frame), generated on the device by the HIP generator
    movies, outs = [], []
    for i in range(n_chunks):
        g0, g1 = block_cyclic_range(rank, world, sizes, i) if gather else (rank * P, rank * P + P)
        movies.append(texture_stack_torch(n, g1 - g0 + 1, seed, dev, first_frame=g0, solver=solver, wobble=args.wobble))
        outs.append([torch.empty((g1 - g0, n, n), dtype=torch.float64, device=dev) for _ in range(4)])   # vx, vy, gm, speed
    gathered = [torch.empty((world * P, n, n), dtype=torch.float64, device=dev) for _ in range(3)] if gather else None
    torch.cuda.synchronize()

    params = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, rtol=args.rtol,
                                    coarse_precision={"float64": 0, "float32": 1, "bfloat16": 2, "float8": 3}[args.coarse_precision],
                                    vcycle_precision=vp_arg,
                                    nu_pre=args.nu_pre, nu_post=args.nu_post, nu_pre_coarse=args.nu_pre_coarse,
                                    nu_post_coarse=args.nu_post_coarse)
    if args.w_cycle_level is not None:
        params.w_cycle_level = args.w_cycle_level
    if args.w_cycle_visits is not None:
        params.w_cycle_visits = args.w_cycle_visits
    if args.warm_start_stride is not None:
        params.warm_start_stride = args.warm_start_stride
    if args.krylov_method is not None:
        params.krylov_method = args.krylov_method
    coarse_bytes = {"float64": 8.0, "float32": 4.0, "bfloat16": 45 * 4 / 81.0, "float8": 30 * 4 / 81.0}[args.coarse_precision]   # per coefficient

    def step(pv=None, mv=None):
        pv = pv or params
        works, stats_all = [], []
        for i in range(n_chunks):
            m = (mv or movies)[i]
            o = outs[i]
            stats_all.append(solver.solve_dev(m, m.shape[0], pv, o[0], o[1], o[2], o[3], stats=True))
            # solve_dev returns after the solver's stream has drained: the chunk's fields are final
            if gathered is not None:
                for dst, src in zip(gathered, o[:3]):
                    works.append(allgather_chunk(dst, src, sizes, i))       # RCCL's stream, under the next solve
        for w in works:
            w.wait()
        if works:
            # the exchange is part of the step, and the next step's solver (its own stream) must not overwrite the
            # fields RCCL is still reading
            torch.cuda.current_stream().synchronize()
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
    alg_total = sum(solver.profile_bytes(r[0], r[1]) for r in table)
    moved_total = sum(solver.profile_moved(r[0], r[1]) for r in table)
    counted_ms = sum(r[3] for r in with_bytes)
    if args.profile_table and rank == 0:
        for name, lvl, cnt, ms in sorted(table, key=lambda r: -r[3]):
            gb = solver.profile_bytes(name, lvl) / 1e9
            mv = solver.profile_moved(name, lvl) / 1e9
            print(f"  {name:13s} L{lvl:<2d} launches {cnt:7d}  total {ms:10.3f} ms  avg {1e3 * ms / cnt:9.2f} us  "
                  f"{100 * ms / total_ms:5.1f}%  {gb / (ms * 1e-3) if gb else 0:8.0f} GB/s"
                  + (f"  ({mv / (ms * 1e-3):.0f} GB/s moved)" if abs(mv - gb) > 1e-6 * gb else ""), file=sys.stderr)
        print(f"  all kernels: {total_ms:.1f} ms of GPU time per warm-up step set, {alg_total / 1e9:.1f} GB algorithmic "
              f"({alg_total / 1e9 / (total_ms * 1e-3):.0f} GB/s over all kernel time)", file=sys.stderr)
    dom_name, dom_level = dom[0], dom[1]
    warm_steps = max(1, args.warmup)
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
    moved_bytes = solver.profile_moved(dom_name, dom_level)
    solver.profile_enable(False)
    bpp = algorithmic_bytes_per_pixel(dom_name, coarse_bytes)
    avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
    bytes_per_launch = (alg_bytes / cnt) if (cnt and alg_bytes > 0) else (bpp * pix_per_launch if bpp else None)
    achieved = (bytes_per_launch / avg_s) / 1e9 if (bytes_per_launch and cnt) else None
    achieved_moved = (moved_bytes / cnt / avg_s / 1e9) if (cnt and moved_bytes > 0) else achieved
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    # the PMC passes were taken with all pairs of the stack in one batch; with chunked solves (multi-GPU overlap) a
    # launch processes fewer pairs and the per-launch traffic figure does not apply
    if os.path.exists(tpath) and B >= P and n_chunks == 1 and not args.wobble:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"{dom_name}_L{dom_level}_{n}x{n}x{T}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if world == 1:
        workload = (f"{n}x{n}x{T} synthetic translating texture (seed {seed}), speed_alpha=1, remodelling_alpha=1e4, "
                    f"rtol={args.rtol:g}, all {P} pairs solved to the stopping rule")
    else:
        workload = (f"{n}x{n}x{world * P + 1} synthetic translating texture (seed {seed}) over {world} GPUs "
                    f"({P} pairs = {T} frames per GPU" + (", BASELINE configs[3]: 1024x1024x1024 over 8 GPUs" if (n, world, P) == (1024, 8, 128) else "")
                    + f"), speed_alpha=1, remodelling_alpha=1e4, rtol={args.rtol:g}, all pairs solved to the stopping rule, "
                    f"flow fields all-gathered to every GPU inside the step")
    out = {
        "metric": "frame-pairs/sec", "value": value, "unit": "frame-pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu": T, "wobble": args.wobble,
                   "pairs_in_flight": B, "levels": solver.num_levels, "coarse_stencils": args.coarse_precision, "vcycle_vectors": args.vcycle_precision,
                   "sweeps": [args.nu_pre, args.nu_post, args.nu_pre_coarse, args.nu_post_coarse],
                   "w_cycle_level": int(params.w_cycle_level), "w_cycle_visits": int(params.w_cycle_visits),
                   "warm_start_stride": int(params.warm_start_stride),
                   "allgather": gathered is not None, "gather_chunks": n_chunks, "chunk_sizes": sizes,
                   "iterations_max": int(stats["iterations"].max()), "iterations_mean": float(stats["iterations"].mean()),
                   "relres_max": float(stats["relative_residual"].max()),
                   "converged": bool(stats["converged"].all())},
        # `achieved` = the bytes one launch HAS TO MOVE (every stream read / written once per pass) over the
        # average launch time from HIP events on the solver's stream; `frac` = achieved / peak is an HBM fraction.  `traffic` is
        # the PMC figure of the same launch (halo re-reads included) and is to be compared with `bytes_per_launch`.
        # `effective_per_sweep` counts SURVEY.md 8(d)'s 80 B per pixel for EVERY sweep a pass performs (the level-0 pass performs
        # two): a rate of work, not of traffic - it may exceed the peak and is not a roofline fraction.
        "roofline": {"bound": "hbm", "kernel": f"{dom_name}@L{dom_level}", "achieved": achieved_moved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (achieved_moved / HBM_PEAK_GBS) if achieved_moved else None, "traffic": traffic,
                     "traffic_over_moved": (traffic / (moved_bytes / cnt)) if (traffic and cnt and moved_bytes) else None,
                     "launches": cnt, "pairs_per_launch": (units / cnt) if cnt else None, "avg_launch_us": 1e6 * avg_s if cnt else None,
                     "bytes_per_launch": (moved_bytes / cnt) if cnt else None,
                     "effective_per_sweep": {"bytes_per_launch": bytes_per_launch, "rate": achieved, "unit": "GB/s"},
                     "share_of_gpu_time": dom[3] / total_ms,
                     # the whole solve, not only the dominant kernel: bytes every byte-counted launch of the warm-up step(s)
                     # has to move over the step time (set-up kernels without a byte count count as time only)
                     "whole_solve": {"bytes_per_step": moved_total / warm_steps,
                                     "achieved": moved_total / warm_steps / (dt / args.steps) / 1e9,
                                     "frac": moved_total / warm_steps / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                                     "effective_per_sweep": alg_total / warm_steps / (dt / args.steps) / 1e9,
                                     "kernel_time_share_with_byte_count": counted_ms / total_ms}},
    }
    if world == 1 and not use_dist and not args.no_variants:
        # informational, not the headline: the same step with other settings (headline = library defaults: float8 /
        # float32 row-sum-preserving Galerkin stencils, float64 V-cycle vectors, two-phase warm start).  Arithmetic, Krylov vectors, stopping rule
        # and results are float64 in all of them.
        def timed(pv, mv=None):
            step(pv, mv)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stv = step(pv, mv)
            torch.cuda.synchronize()
            d = time.perf_counter() - t1
            return {"value": P / d, "unit": "frame-pairs/s", "iterations_mean": float(stv["iterations"].mean()),
                    "relres_max": float(stv["relative_residual"].max()), "converged": bool(stv["converged"].all())}
        common = dict(speed_alpha=1.0, remodelling_alpha=1e4, rtol=args.rtol, nu_pre=args.nu_pre, nu_post=args.nu_post,
                      nu_pre_coarse=args.nu_pre_coarse, nu_post_coarse=args.nu_post_coarse,
                      w_cycle_level=int(params.w_cycle_level), w_cycle_visits=int(params.w_cycle_visits),
                      warm_start_stride=int(params.warm_start_stride))
        cold = dict(common, warm_start_stride=0)
        cp, vp = int(params.coarse_precision), int(params.vcycle_precision)
        out["variants"] = {
            "constant_initial_fields_for_every_pair": timed(_native.default_params(vcycle_precision=vp, coarse_precision=cp, **cold)),
            "all_float64_storage": timed(_native.default_params(vcycle_precision=0, coarse_precision=0, **common)),
            "float32_stencils": timed(_native.default_params(vcycle_precision=vp, coarse_precision=1, **common)),
            "bfloat16_stencils": timed(_native.default_params(vcycle_precision=vp, coarse_precision=2, **common)),
            "float64_vcycle_vectors_on_every_level": timed(_native.default_params(vcycle_precision=0, coarse_precision=cp, **common)),
            "float32_vcycle_vectors_on_every_level": timed(_native.default_params(vcycle_precision=2, coarse_precision=cp, **common)),
        }
        if not args.wobble:
            # a less warm-start-friendly sibling of the headline input: the same texture with a time-varying flow
            # (frame-to-frame step (0.3, 0.6) x (1 +- 0.3)), default settings
            wob = [texture_stack_torch(n, T, seed, dev, first_frame=0, solver=solver, wobble=0.3)]
            out["variants"]["time_varying_flow_wobble_0.3"] = timed(params, wob)
            out["variants"]["time_varying_flow_wobble_0.3_cold"] = timed(_native.default_params(vcycle_precision=vp, coarse_precision=cp, **cold), wob)
            del wob
        free_now, _ = _native.device_memory(local_rank)
        if n_chunks == 1 and P >= 8 and per_pair * (P + 2) < 0.8 * free_now:   # (beside the main context: 2048^2 x 128 does not fit twice)
            # two contexts, each with one half of the stack, on their own streams and host threads at the same time: the
            # instruction-bound passes of one half run over the bandwidth-bound kernels of the other.
            # Informational: per-launch times overlap in this mode, so the roofline figures above are taken with one context.
            import threading
            cuts = [0, (P + 1) // 2, P]
            subs = [_native.Solver(n, n, cuts[i + 1] - cuts[i], device=local_rank) for i in range(2)]

            def both():
                res = [None, None]

                def work(i):
                    a, b = cuts[i], cuts[i + 1]
                    o = outs[0]
                    res[i] = subs[i].solve_dev(movies[0][a:b + 1], b - a + 1, params, o[0][a:b], o[1][a:b], o[2][a:b], o[3][a:b], stats=True)
                th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                return np.concatenate(res)
            both()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stv = both()
            torch.cuda.synchronize()
            d = time.perf_counter() - t1
            out["variants"]["two_contexts_at_once"] = {
                "value": P / d, "unit": "frame-pairs/s", "iterations_mean": float(stv["iterations"].mean()),
                "relres_max": float(stv["relative_residual"].max()), "converged": bool(stv["converged"].all()),
                "what": "the stack as two halves solved at the same time by two contexts (two streams, two host threads)"}
            for sub in subs:
                sub.close()
    if world == 1 and not use_dist and not args.no_variants:
        # the reference's second consumer of the path: vary_regularisation (OF.py:1918-1998) in the shape of its own sweep,
        # AVOF.py:608-615 - a 20 x 20 logspace(-1, 4) grid of (speed_alpha, remodelling_alpha) on a down-sampled 8-bit
        # pair with smoothing_sigma = 1; one native call, host arrays in, summary tables out
        from opticalflow_amd import optical_flow as of
        from opticalflow_amd.synthetic import texture_stack_numpy
        vm = np.round(255.0 * texture_stack_numpy(128, 3, seed=7)).astype(np.uint8)
        grid = np.logspace(-1, 4, 20)
        vt = []
        for _ in range(2):           # the first call creates the device context
            t1 = time.perf_counter()
            vr = of.vary_regularisation(vm, grid, grid, smoothing_sigma=1.0, use_direct_solver=True, return_stats=True)
            vt.append(time.perf_counter() - t1)
        out["variants"]["vary_regularisation"] = {
            "value": grid.size ** 2 / vt[-1], "unit": "combinations/s", "seconds": vt[-1], "seconds_first_call": vt[0],
            "combinations": int(grid.size ** 2), "converged_all": bool(np.asarray(vr["stats"]["converged_all"]).all()),
            "max_iterations_used": int(np.asarray(vr["stats"]["max_iterations_used"]).max()),
            "what": "vary_regularisation(128x128x3 8-bit texture, logspace(-1, 4, 20)^2, smoothing_sigma=1, use_direct_solver=True) - "
                    "the reference script's own call, AVOF.py:608-615: 400 combinations x 2 frame pairs"}
        of.release_device_memory()
    if world == 1 and not use_dist and not args.no_end_to_end:
        # end-to-end rate of the drop-in call, pageable numpy arrays in and out (the reference's
        # contract), i.e. including the float64 copy, H2D of the movie and D2H of the four result stacks.  Never `value`.
        from opticalflow_amd import optical_flow as of
        movie_host = movies[0].cpu().numpy()
        solver.close()
        solver = None
        torch.cuda.empty_cache()
        e2e = []
        for _ in range(2):          # the first call creates the cached device context
            t1 = time.perf_counter()
            r = of.variational_optical_flow(movie_host, speed_alpha=1.0, remodelling_alpha=1e4, return_stats=True)
            e2e.append(time.perf_counter() - t1)
            ok = bool(r["stats"]["converged"].all())
            del r
        of.release_device_memory()
        out["end_to_end"] = {"value": P / e2e[-1], "unit": "frame-pairs/s", "seconds": e2e[-1], "seconds_first_call": e2e[0],
                             "converged": ok,
                             "what": "opticalflow_amd.optical_flow.variational_optical_flow(movie) with pageable numpy arrays "
                                     "in and out: float64 copy, H2D, solve, D2H of v_x, v_y, speed, remodelling "
                                     f"({movie_host.nbytes / 1e9:.1f} GB up, {4 * P * n * n * 8 / 1e9:.1f} GB down)"}
        del movie_host
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, seed)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if solver is not None:
        solver.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
