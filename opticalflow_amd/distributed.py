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


def _gather_stack(local, counts, group):
    """All-gather per-rank stacks ``local`` ((counts[rank], ...) torch tensor, CPU for gloo / device for RCCL) into the
    natural-order stack (sum(counts), ...) with ONE ``all_gather_into_tensor`` on a contiguous [rank][pair] buffer.
    Even shards: the buffer IS the result (no copy).  Uneven shards are padded to the longest and compacted."""
    import torch
    import torch.distributed as dist
    world, m = len(counts), max(counts)
    tail = tuple(local.shape[1:])
    if local.shape[0] == m:
        buf = local.contiguous()
    else:
        buf = torch.zeros((m,) + tail, dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
    out = torch.empty((world * m,) + tail, dtype=local.dtype, device=local.device)     # [rank][pair], flattened
    dist.all_gather_into_tensor(out, buf, group=group)
    if min(counts) == m:
        return out
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0)


def variational_optical_flow_sharded(movie, solve_fn=None, group=None, device=None, output="numpy", n_frames=None,
                                     gather_movie=None, **kwargs):
    """Solve a stack across all ranks of ``group`` and return the full result dict on every rank.

    ``movie`` is the full (T, N_i, N_j) stack (every rank passes the same array or device tensor; only its own frame
    range is read) - or a FRAME PROVIDER ``movie(first_frame, count) -> (count, N_i, N_j)`` array / device tensor together
    with ``n_frames=T``: each rank then asks for exactly the frames of its own pairs (generated or loaded on its GPU, as
    bench.py does) and no rank ever holds the whole movie.  ``gather_movie`` (default: True for an
    array, False for a provider): whether ``original_data`` / ``blurred_data`` are the whole movie on every rank
    (all-gathered from the shards when it comes from a provider) or this rank's frames only, in which case
    ``result["frame_range"] = (first, last + 1)`` says which.  Rank r solves the contiguous pair range ``shard_pair_range(T - 1, world, r)`` on its own GPU with
    the device-resident drop-in (``variational_optical_flow(..., output="torch")``), and the flow fields are re-assembled
    straight from the solver's device outputs with one ``all_gather_into_tensor`` per field (RCCL over xGMI; layout
    [rank][pair] = natural order).  ``output="numpy"`` (default, the reference's contract OF.py:1193-1197) copies the
    re-assembled stacks to host arrays; ``output="torch"`` leaves every array on the device.

    ``solve_fn(sub_movie, **kwargs) -> dict`` replaces the per-shard solver (the CPU tests inject a host solver and run the
    same code over gloo).
    """
    import torch
    import torch.distributed as dist
    if output not in ("numpy", "torch"):
        raise ValueError("output must be 'numpy' or 'torch'")
    native = solve_fn is None
    if native:
        from .optical_flow import variational_optical_flow as solve_fn
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        kwargs.setdefault("device", device.index)
        kwargs["output"] = "torch"
    if device is None:
        device = torch.device("cpu")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    provider = movie if callable(movie) else None
    if provider is not None:
        if n_frames is None:
            raise ValueError("a frame provider needs n_frames (the length of the whole stack)")
        T = int(n_frames)
        start, stop = shard_pair_range(T - 1, world, rank)
        # (an empty shard still asks for one frame: the frame shape of the re-assembled stack)
        movie = provider(start, stop - start + 1) if stop > start else provider(min(start, T - 1), 1)
        frame0 = start if stop > start else min(start, T - 1)
    if gather_movie is None:
        gather_movie = provider is None
    is_tensor = hasattr(movie, "data_ptr")
    if not is_tensor:
        movie = np.asarray(movie)
    if provider is None:
        T = movie.shape[0]
        frame0 = 0
    P = T - 1
    frame_shape = tuple(movie.shape[1:])
    counts = [shard_pair_range(P, world, r)[1] - shard_pair_range(P, world, r)[0] for r in range(world)]
    start, stop = shard_pair_range(P, world, rank)
    fields = ("v_x", "v_y", "remodelling")
    blur = kwargs.get("smoothing_sigma") is not None
    scal = np.zeros(4)

    def on_device(a):
        return torch.as_tensor(a).to(device=device, dtype=torch.float64)

    if stop > start:
        sub = solve_fn(movie[start - frame0: stop + 1 - frame0], **kwargs)      # one overlap frame per shard
        local = {k: on_device(sub[k]) for k in fields}
        scal[:] = (sub["L1_functional"], sub["remodelling_functional"], sub["speed_functional"],
                   float(bool(sub["converged"])))
        blurred = on_device(sub["blurred_data"]) if blur else None     # frames start .. stop (count + 1)
    else:
        local = {k: torch.zeros((0,) + frame_shape, dtype=torch.float64, device=device) for k in fields}
        scal[3] = 1.0
        blurred = torch.zeros((1,) + frame_shape, dtype=torch.float64, device=device) if blur else None
    out = {k: _gather_stack(local[k], counts, group) for k in fields}
    # functionals are sums over pairs (OF.py:1203-1205); 'converged' is the flag of the LAST pair (OF.py:1202)
    s = torch.as_tensor(scal[:3].copy(), device=device)
    dist.all_reduce(s, group=group)
    flags = torch.zeros(world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(flags, torch.as_tensor(scal[3:4].copy(), device=device), group=group)
    last_rank = max(r for r in range(world) if counts[r] > 0)

    def whole_stack(frames):
        """Frames of this rank's pairs (+ the final frame on the last non-empty rank) -> the whole stack on every rank."""
        body = _gather_stack(frames[: counts[rank]], counts, group)
        lastf = torch.zeros((world,) + frame_shape, dtype=torch.float64, device=device)
        tail = frames[counts[rank]: counts[rank] + 1] if frames.shape[0] > counts[rank] else torch.zeros((1,) + frame_shape, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(lastf, tail.contiguous(), group=group)
        return torch.cat([body, lastf[last_rank: last_rank + 1]], dim=0)

    own = slice(start - frame0, stop + 1 - frame0) if stop > start else slice(0, 0)
    if provider is None and gather_movie:
        # (the caller holds the whole movie anyway; round 2 made this float64 copy of ALL of it on every rank even when
        # only the shard was wanted - 8 x 8 GiB of host memory at BASELINE config 4)
        movie64 = on_device(movie) if output == "torch" else (movie.cpu().numpy() if is_tensor else movie).astype(np.float64)
    elif gather_movie:
        full = whole_stack(on_device(movie[own]) if stop > start else torch.zeros((0,) + frame_shape, dtype=torch.float64, device=device))
        movie64 = full if output == "torch" else full.cpu().numpy()
    else:
        mine = movie[own]
        movie64 = on_device(mine) if output == "torch" else (mine.cpu().numpy() if is_tensor else np.asarray(mine)).astype(np.float64)
    if blur:
        # the blurred stack the solves ran on (OF.py:1199): every rank contributes the frames of its pairs, the last
        # non-empty rank also the final frame (the blur is per frame, so the overlap frames agree)
        blurred_full = whole_stack(blurred) if gather_movie else (blurred[: counts[rank] + 1] if stop > start else blurred[:0])
    result = dict(out)
    result["speed"] = torch.sqrt(out["v_x"] ** 2 + out["v_y"] ** 2)
    if output == "numpy":
        result = {k: v.cpu().numpy() for k, v in result.items()}
    result["original_data"] = movie64
    if blur:
        result["blurred_data"] = blurred_full if output == "torch" else blurred_full.cpu().numpy()
    else:
        result["blurred_data"] = movie64                      # same object, as in the reference (OF.py:773)
    if not gather_movie:
        result["frame_range"] = (start, stop + 1) if stop > start else (start, start)
    result["delta_x"] = kwargs.get("delta_x", 1.0)
    result["delta_t"] = kwargs.get("delta_t", 1.0)
    result["converged"] = bool(flags[last_rank].item())
    result["L1_functional"], result["remodelling_functional"], result["speed_functional"] = (float(v) for v in s.cpu())
    return result


# ---------------------------------------------------------------------------------------------------------
# Overlapped exchange (bench.py): the rank's work is cut into a few chunks and the all-gather of chunk i runs on RCCL's
# stream while chunk i+1 is solved.  Pairs are dealt BLOCK-CYCLICALLY: chunk i of rank r is the global pair range
#     [world * S_i + r * c_i, world * S_i + (r + 1) * c_i),   c_i = chunk size, S_i = c_0 + ... + c_{i-1},
# so the chunk-i blocks of all ranks are adjacent in the natural-order stack and ONE all_gather_into_tensor per field
# and chunk writes them in place ([rank][pair] layout, contiguous, no temporary and no copy-out).
# ---------------------------------------------------------------------------------------------------------
def chunk_plan(n_pairs_per_rank, n_chunks=0, min_warm=48):
    """Chunk sizes for the overlapped exchange.  Default: three chunks (40 %, 40 %, 20 %) when both large ones keep at
    least ``min_warm`` pairs (the two-phase warm start needs >= 16 Mpixel of first-phase pairs, i.e. 48 pairs of
    1024^2) - the gather of the last chunk cannot hide under a solve, so it is the small one; two halves for medium
    stacks; one chunk otherwise.  ``n_chunks > 0``: that many (nearly) equal chunks."""
    P = int(n_pairs_per_rank)
    if n_chunks > 0:
        n_chunks = min(n_chunks, P)
        base, extra = divmod(P, n_chunks)
        return [base + (1 if i < extra else 0) for i in range(n_chunks)]
    big = -(-2 * P // 5)
    if big >= min_warm and P - 2 * big >= 1:
        return [big, big, P - 2 * big]
    if P >= 32:
        return [P - P // 2, P // 2]
    return [P]


def block_cyclic_range(rank, world, sizes, chunk):
    """Global pair range [g0, g1) of ``chunk`` of ``rank`` (sizes = chunk sizes per rank, see above)."""
    s = sum(sizes[:chunk])
    g0 = world * s + rank * sizes[chunk]
    return g0, g0 + sizes[chunk]


def allgather_chunk(gathered, local_chunk, sizes, chunk, group=None, async_op=True):
    """Start the all-gather of this rank's ``local_chunk`` ((sizes[chunk], N_i, N_j) contiguous tensor) into the
    natural-order stack ``gathered`` ((world * sum(sizes), N_i, N_j)): the chunk's blocks of all ranks are adjacent there
    (block-cyclic distribution), so the collective writes its [rank][pair] output in place.  Returns the work handle."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    g0 = world * sum(sizes[:chunk])
    out = gathered[g0: g0 + world * sizes[chunk]]
    return dist.all_gather_into_tensor(out, local_chunk, group=group, async_op=async_op)


def launch_ranks(script, argv, n_ranks, timeout=None):
    """Run ``script argv`` as ``n_ranks`` ranks of one node (``python -m torch.distributed.run``, rendezvous on
    127.0.0.1 at a free port) from a parent that has not touched the GPU: fresh child processes, the parent only relays
    their stdout (rank 0 prints the result line) and returns the launcher's exit code."""
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    import os
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=timeout)
    sys.stdout.write(proc.stdout.decode(errors="replace"))
    sys.stdout.flush()
    return proc.returncode


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
