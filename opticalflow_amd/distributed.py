"""Multi-GPU driver: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Frame pairs are independent units (pair k needs only frames k and k+1, OF.py:794-795; the
reference's only cross-pair coupling is the warm-start guess, OF.py:803-806).  Each rank solves a
contiguous range of pairs; the ONLY collective is the all-gather that re-assembles the output stack
(BASELINE config 4).  Uneven shards are padded to the longest shard for the collective.
"""
from __future__ import annotations

import numpy as np


def shard_pair_range(n_pairs: int, world: int, rank: int):
    """Contiguous pair range [start, stop) of ``rank``: the first ``n_pairs % world`` ranks get one
    extra pair.  Ranks beyond ``n_pairs`` get an empty range."""
    base, extra = divmod(n_pairs, world)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return start, stop


def _gather_field(local, counts, group, device):
    """All-gather a (n_local, N_i, N_j) array with per-rank counts; returns (sum counts, N_i, N_j)."""
    import torch
    import torch.distributed as dist
    world = len(counts)
    m = max(counts)
    shape = local.shape[1:]
    buf = torch.zeros((m,) + tuple(shape), dtype=torch.float64, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.as_tensor(local, device=device)
    out = torch.empty((world * m,) + tuple(shape), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = [out[r * m: r * m + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def variational_optical_flow_sharded(movie, solve_fn=None, group=None, device=None, **kwargs):
    """Solve a stack across all ranks of ``group`` and return the full result dict on every rank.

    ``movie`` is the full (T, N_i, N_j) stack (every rank passes the same array; only its own frame
    range is read).  ``solve_fn(sub_movie, **kwargs) -> dict`` defaults to the single-GPU drop-in
    ``optical_flow.variational_optical_flow`` on this rank's device.
    """
    import torch
    import torch.distributed as dist
    if solve_fn is None:
        from .optical_flow import variational_optical_flow as solve_fn
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        kwargs.setdefault("device", device.index)
    if device is None:
        device = torch.device("cpu")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    movie = np.asarray(movie)
    T = movie.shape[0]
    P = T - 1
    counts = [shard_pair_range(P, world, r)[1] - shard_pair_range(P, world, r)[0] for r in range(world)]
    start, stop = shard_pair_range(P, world, rank)
    fields = ("v_x", "v_y", "remodelling")
    scal = np.zeros(4)
    if stop > start:
        sub = solve_fn(movie[start: stop + 1], **kwargs)      # one overlap frame per shard
        local = {k: np.asarray(sub[k]) for k in fields}
        scal[:] = (sub["L1_functional"], sub["remodelling_functional"], sub["speed_functional"],
                   float(bool(sub["converged"])))
    else:
        local = {k: np.zeros((0,) + movie.shape[1:]) for k in fields}
        scal[3] = 1.0
    out = {k: _gather_field(local[k], counts, group, device).cpu().numpy() for k in fields}
    # functionals are sums over pairs (OF.py:1203-1205); 'converged' is the flag of the LAST pair (OF.py:1202)
    s = torch.as_tensor(scal[:3].copy(), device=device)
    dist.all_reduce(s, group=group)
    flags = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(flags, torch.as_tensor(scal[3:4].copy(), device=device), group=group)
    last_rank = max(r for r in range(world) if counts[r] > 0)
    movie64 = movie.astype(np.float64)
    delta_x, delta_t = kwargs.get("delta_x", 1.0), kwargs.get("delta_t", 1.0)
    result = dict(out)
    result["speed"] = np.sqrt(out["v_x"] ** 2 + out["v_y"] ** 2)
    result["original_data"] = movie64
    result["blurred_data"] = movie64
    result["delta_x"] = delta_x
    result["delta_t"] = delta_t
    result["converged"] = bool(flags[last_rank].item())
    result["L1_functional"], result["remodelling_functional"], result["speed_functional"] = (float(v) for v in s.cpu())
    return result


def allgather_chunk(gathered, local, a, b, n_pairs_per_rank, group=None, async_op=True):
    """Start the all-gather of pairs [a, b) of this rank's ``local`` (n_pairs_per_rank, N_i, N_j) tensor into the
    re-assembled stack ``gathered`` ((world * n_pairs_per_rank, N_i, N_j), natural order: rank r's pair k at
    ``r * n_pairs_per_rank + k``).  Returns the work handle (``.wait()``) when ``async_op``.  Used by bench.py to
    overlap the exchange of one chunk with the solve of the next."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    P = n_pairs_per_rank
    outs = [gathered[r * P + a: r * P + b] for r in range(world)]
    return dist.all_gather(outs, local[a:b], group=group, async_op=async_op)


_VARIATION_KEYS = ("speed_means", "speed_variances", "remodelling_means", "remodelling_variances", "functional", "converged")


def vary_regularisation_sharded(movie, speed_alpha_values, remodelling_alpha_values, sweep_fn=None, group=None,
                                device=None, filename=None, **kwargs):
    """``vary_regularisation`` (OF.py:1918-1998) across all ranks of ``group``: the flattened list of
    ``(speed_alpha, remodelling_alpha)`` combinations is cut into contiguous ranges (``shard_pair_range``), each rank
    runs the native sweep on its range, and one all-reduce of the six small summary tables re-assembles the result on
    every rank (each entry is written by exactly one rank).  ``sweep_fn(movie, speed_alphas, remodelling_alphas,
    **kwargs) -> dict`` defaults to the single-GPU ``optical_flow.vary_regularisation`` on this rank's device."""
    import torch
    import torch.distributed as dist
    if sweep_fn is None:
        from .optical_flow import vary_regularisation as sweep_fn
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        kwargs.setdefault("device", device.index)
    if device is None:
        device = torch.device("cpu")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_sa, n_ra = len(speed_alpha_values), len(remodelling_alpha_values)
    start, stop = shard_pair_range(n_sa * n_ra, world, rank)
    tables = np.zeros((len(_VARIATION_KEYS), n_sa, n_ra))
    for i in range(n_sa):                     # this rank's part of row i: columns [j0, j1)
        j0, j1 = max(start, i * n_ra) - i * n_ra, min(stop, (i + 1) * n_ra) - i * n_ra
        if j1 <= j0:
            continue
        part = sweep_fn(movie, np.asarray(speed_alpha_values)[i:i + 1], np.asarray(remodelling_alpha_values)[j0:j1], **kwargs)
        for t, key in enumerate(_VARIATION_KEYS):
            tables[t, i, j0:j1] = np.asarray(part[key], dtype=np.float64)[0]
    buf = torch.as_tensor(tables, device=device)
    dist.all_reduce(buf, group=group)
    tables = buf.cpu().numpy()
    result = {"speed_alpha_values": speed_alpha_values, "remodelling_alpha_values": remodelling_alpha_values}
    for t, key in enumerate(_VARIATION_KEYS):
        result[key] = tables[t] > 0.5 if key == "converged" else tables[t]
    if filename is not None and rank == 0:
        np.save(filename, result)
    return result
