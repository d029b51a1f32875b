"""World-size-2 gloo tests (CPU) of the multi-GPU host logic: pair-range sharding with one overlap
frame, the all-gather re-assembly with uneven shards, and the reduction of the functionals.  The
per-shard solver is injected (the CPU oracle) because the product solver needs a GPU."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from opticalflow_amd.distributed import shard_pair_range


def test_shard_ranges_cover_all_pairs_once():
    for P in (1, 2, 7, 63, 255, 1023):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                a, b = shard_pair_range(P, world, r)
                assert 0 <= a <= b <= P
                seen += list(range(a, b))
            assert seen == list(range(P))
            sizes = [shard_pair_range(P, world, r)[1] - shard_pair_range(P, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_frames, q, sigma=None, output="numpy"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vof_oracle as orc
        from opticalflow_amd.distributed import variational_optical_flow_sharded
        movie = orc.make_texture_stack(24, n_frames, seed=5)
        calls = []

        def solve_fn(sub, **kw):
            calls.append(sub.shape[0])
            return orc.variational_optical_flow(sub, **kw)

        kw = dict(speed_alpha=1.0, remodelling_alpha=50.0, delta_x=0.5, delta_t=1.0)
        if sigma is not None:
            kw["smoothing_sigma"] = sigma
        res = variational_optical_flow_sharded(movie, solve_fn=solve_fn, output=output, **kw)
        if output == "torch":
            import torch
            assert all(isinstance(res[k], torch.Tensor) for k in ("v_x", "v_y", "speed", "remodelling", "original_data",
                                                                   "blurred_data"))
            res = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}
        assert (res["blurred_data"] is res["original_data"]) == (sigma is None)      # OF.py:770-773
        q.put((rank, calls, res["v_x"], res["remodelling"], res["L1_functional"], res["speed_functional"],
               res["converged"], res["blurred_data"], res["speed"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,sigma,output", [(4, None, "numpy"), (5, None, "numpy"), (2, None, "numpy"),
                                                   (5, 1.5, "numpy"), (4, 1.0, "torch"), (2, 1.5, "numpy")])
def test_two_rank_gloo_sharded_solve_matches_single_process(n_frames, sigma, output):
    from oracle import vof_oracle as orc
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q, sigma, output)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    movie = orc.make_texture_stack(24, n_frames, seed=5)
    ref = orc.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=50.0, delta_x=0.5, delta_t=1.0,
                                       **({} if sigma is None else {"smoothing_sigma": sigma}))
    P = n_frames - 1
    for rank, calls, vx, gm, L1, sf, conv, blurred, speed in got:
        # the stack the solves ran on (OF.py:1199), re-assembled from the shards' blurred frames
        np.testing.assert_allclose(blurred, ref["blurred_data"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(speed, ref["speed"], rtol=1e-9, atol=1e-12)
        a, b = shard_pair_range(P, world, rank)
        assert calls == ([b - a + 1] if b > a else [])       # shard + one overlap frame
        assert vx.shape == ref["v_x"].shape
        np.testing.assert_allclose(vx, ref["v_x"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gm, ref["remodelling"], rtol=1e-9, atol=1e-12)
        assert L1 == pytest.approx(ref["L1_functional"], rel=1e-9)
        assert sf == pytest.approx(ref["speed_functional"], rel=1e-9)
        assert conv is True


def _provider_worker(rank, world, port, n_frames, q, gather_movie, sigma):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vof_oracle as orc
        from opticalflow_amd.distributed import variational_optical_flow_sharded
        asked = []

        def provider(first, count):                      # this rank's frames only (the generator takes a first_frame offset)
            asked.append((first, count))
            return orc.make_texture_stack(24, count, seed=5, first_frame=first)

        def solve_fn(sub, **kw):
            return orc.variational_optical_flow(sub, **kw)

        kw = dict(speed_alpha=1.0, remodelling_alpha=50.0, delta_x=0.5, delta_t=1.0)
        if sigma is not None:
            kw["smoothing_sigma"] = sigma
        res = variational_optical_flow_sharded(provider, solve_fn=solve_fn, n_frames=n_frames, gather_movie=gather_movie, **kw)
        q.put((rank, asked, res["v_x"], res["original_data"], res["blurred_data"], res.get("frame_range"), res["L1_functional"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,gather_movie,sigma", [(6, False, None), (5, True, None), (2, False, None), (5, True, 1.5), (6, False, 1.5)])
def test_two_rank_gloo_sharded_solve_from_a_frame_provider(n_frames, gather_movie, sigma):
    """The sharded entry point with a per-rank frame provider: every rank asks for exactly the frames of its own pairs (+ the
    overlap frame), nobody holds the whole movie unless gather_movie asks for it, results equal the single-process solve."""
    from oracle import vof_oracle as orc
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_provider_worker, args=(r, world, port, n_frames, q, gather_movie, sigma)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    movie = orc.make_texture_stack(24, n_frames, seed=5)
    ref = orc.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=50.0, delta_x=0.5, delta_t=1.0,
                                       **({} if sigma is None else {"smoothing_sigma": sigma}))
    P = n_frames - 1
    for rank, asked, vx, orig, blurred, frange, L1 in got:
        a, b = shard_pair_range(P, world, rank)
        assert asked == ([(a, b - a + 1)] if b > a else [(min(a, n_frames - 1), 1)])
        np.testing.assert_allclose(vx, ref["v_x"], rtol=1e-9, atol=1e-12)
        assert L1 == pytest.approx(ref["L1_functional"], rel=1e-9)
        if gather_movie:
            assert frange is None
            np.testing.assert_allclose(orig, movie, rtol=0, atol=1e-15)
            np.testing.assert_allclose(blurred, ref["blurred_data"], rtol=0, atol=1e-15)
        else:
            assert frange == ((a, b + 1) if b > a else (a, a))
            np.testing.assert_allclose(orig, movie[a: b + 1] if b > a else movie[:0], rtol=0, atol=1e-15)
            np.testing.assert_allclose(blurred, ref["blurred_data"][a: b + 1] if b > a else movie[:0], rtol=0, atol=1e-15)


def _gather_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        from opticalflow_amd.distributed import allgather_chunk, block_cyclic_range, chunk_plan
        n = 5
        sizes = [3, 2, 1]                       # uneven chunks, as chunk_plan makes them (small last chunk)
        P = sum(sizes)
        gathered = torch.full((world * P, n, n), -1.0, dtype=torch.float64)
        works = []
        for c in range(len(sizes)):
            g0, g1 = block_cyclic_range(rank, world, sizes, c)
            # the value of global pair g is g everywhere: the re-assembled stack must come out in natural order
            local = torch.arange(g0, g1, dtype=torch.float64)[:, None, None].expand(g1 - g0, n, n).contiguous()
            works.append(allgather_chunk(gathered, local, sizes, c))                   # asynchronous, in place
        for w in works:
            w.wait()
        q.put((rank, gathered.numpy(), chunk_plan(128), chunk_plan(255), chunk_plan(40), chunk_plan(9), chunk_plan(9, 3)))
    finally:
        dist.destroy_process_group()


def test_chunked_allgather_reassembles_in_natural_order():
    """bench.py's multi-GPU step: pairs dealt block-cyclically, one all_gather_into_tensor per chunk writing its
    [rank][pair] block of the natural-order stack in place."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P, n = 6, 5
    expect = np.broadcast_to(np.arange(world * P, dtype=np.float64)[:, None, None], (world * P, n, n))
    for rank, g, p128, p255, p40, p9, p9_3 in got:
        np.testing.assert_array_equal(g, expect)
        assert p128 == [52, 52, 24] and p255 == [102, 102, 51] and p40 == [20, 20] and p9 == [9] and p9_3 == [3, 3, 3]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed environment starts two fresh rank processes itself
    (distributed.launch_ranks -> torch.distributed.run on 127.0.0.1) and relays rank 0's single line.  The hidden
    --selftest-launch mode makes the ranks join a gloo group instead of touching a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--selftest-launch"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d == {"selftest": "launch", "n_gpus": 2, "sum": 3.0}
    # a failing child is not hidden: --gpus 2 under a WORLD_SIZE=1 environment is refused
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=root, env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def _sweep_worker(rank, world, port, n_sa, n_ra, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vof_oracle as orc
        from opticalflow_amd.distributed import vary_regularisation_sharded
        movie = orc.make_texture_stack(16, 3, seed=8)
        sa, ra = np.linspace(1.0, 3.0, n_sa), np.logspace(1, 3, n_ra)
        calls = []

        def sweep_fn(mv, a, b, **kw):
            calls.append((len(a), len(b)))
            return orc.vary_regularisation(mv, a, b, **kw)

        res = vary_regularisation_sharded(movie, sa, ra, sweep_fn=sweep_fn, delta_x=0.5, delta_t=1.0)
        q.put((rank, sum(x * y for x, y in calls), {k: res[k] for k in ("speed_means", "remodelling_variances",
                                                                         "functional", "converged")}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_sa,n_ra", [(2, 3), (1, 1), (3, 1)])
def test_two_rank_gloo_sharded_parameter_sweep(n_sa, n_ra):
    """vary_regularisation shards over the (alpha, beta) combinations; one all-reduce of the tables."""
    from oracle import vof_oracle as orc
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, n_sa, n_ra, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    movie = orc.make_texture_stack(16, 3, seed=8)
    ref = orc.vary_regularisation(movie, np.linspace(1.0, 3.0, n_sa), np.logspace(1, 3, n_ra), delta_x=0.5, delta_t=1.0)
    assert sorted(n for _, n, _ in got) == sorted(shard_pair_range(n_sa * n_ra, world, r)[1]
                                                  - shard_pair_range(n_sa * n_ra, world, r)[0] for r in range(world))
    for rank, _, res in got:
        for k in ("speed_means", "remodelling_variances", "functional"):
            np.testing.assert_allclose(res[k], ref[k], rtol=1e-12, err_msg=k)
        assert res["converged"].dtype == bool and res["converged"].all() and res["converged"].shape == (n_sa, n_ra)
