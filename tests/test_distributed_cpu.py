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


def _worker(rank, world, port, n_frames, q):
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

        res = variational_optical_flow_sharded(movie, solve_fn=solve_fn, speed_alpha=1.0, remodelling_alpha=50.0,
                                               delta_x=0.5, delta_t=1.0)
        q.put((rank, calls, res["v_x"], res["remodelling"], res["L1_functional"], res["speed_functional"],
               res["converged"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [4, 5, 2])
def test_two_rank_gloo_sharded_solve_matches_single_process(n_frames):
    from oracle import vof_oracle as orc
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    movie = orc.make_texture_stack(24, n_frames, seed=5)
    ref = orc.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=50.0, delta_x=0.5, delta_t=1.0)
    P = n_frames - 1
    for rank, calls, vx, gm, L1, sf, conv in got:
        a, b = shard_pair_range(P, world, rank)
        assert calls == ([b - a + 1] if b > a else [])       # shard + one overlap frame
        assert vx.shape == ref["v_x"].shape
        np.testing.assert_allclose(vx, ref["v_x"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gm, ref["remodelling"], rtol=1e-9, atol=1e-12)
        assert L1 == pytest.approx(ref["L1_functional"], rel=1e-9)
        assert sf == pytest.approx(ref["speed_functional"], rel=1e-9)
        assert conv is True


def _gather_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        from opticalflow_amd.distributed import allgather_chunk
        P, n = 6, 5
        local = (torch.arange(P * n * n, dtype=torch.float64).reshape(P, n, n) + 1000.0 * rank)
        gathered = torch.full((world * P, n, n), -1.0, dtype=torch.float64)
        works = [allgather_chunk(gathered, local, a, a + 2, P) for a in (0, 2, 4)]     # 3 chunks, asynchronous
        for w in works:
            w.wait()
        q.put((rank, gathered.numpy()))
    finally:
        dist.destroy_process_group()


def test_chunked_allgather_reassembles_in_natural_order():
    """bench.py's multi-GPU step: chunk-wise asynchronous all-gathers into the rank-major output stack."""
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
    expect = np.concatenate([np.arange(P * n * n, dtype=np.float64).reshape(P, n, n) + 1000.0 * r for r in range(world)])
    for rank, g in got:
        np.testing.assert_array_equal(g, expect)


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
