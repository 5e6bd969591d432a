"""Batch sharding across GPUs: whole clouds are independent units -- the encoder state never
crosses clouds and the adaptive-int modes are decided per encode() call (src/v5_codec.cpp:904) -- so a batch is
split by cloud, every rank runs the single-GPU path on its share, and the only exchange is an all-gather of the
per-cloud encoded sizes (so that any rank can lay the batch out / write an index). No payload crosses xGMI.

One large cloud (second row) is split by whole 32768-point chunks instead: every state of
the stage-1 codec resets per chunk (src/v5_codec.cpp:910-915, src/v4_codec.cpp:69) except the adaptive-int modes,
which the cloud's first <= 4096 points decide once (src/v5_codec.cpp:934-949). The rank that holds the head of the
cloud probes the modes, one broadcast of those few bytes and one all-gather of the per-rank byte counts are the
only exchange; each rank's framed chunks are byte-identical to its range of the single-GPU stream.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on the GPU box, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_clouds(n_clouds: int, world_size: int, rank: int) -> List[int]:
    """Round-robin ownership: cloud k belongs to rank k % world_size (keeps ragged batches balanced)."""
    return list(range(rank, n_clouds, world_size))


def batch_layout(sizes: Sequence[int]) -> Tuple[np.ndarray, int]:
    """Exclusive scan of the encoded sizes: byte offset of every cloud in the concatenated batch, and the total."""
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    off[1:] = np.cumsum(np.asarray(sizes, dtype=np.int64))
    return off[:-1], int(off[-1])


def exchange_sizes(local_sizes: Sequence[int], n_clouds: int, rank: int, world_size: int, device=None) -> np.ndarray:
    """All-gather the encoded size of every cloud. Rank r contributes the sizes of shard_clouds(n_clouds, W, r)."""
    import torch
    import torch.distributed as dist

    per_rank = (n_clouds + world_size - 1) // world_size
    mine = torch.full((per_rank,), -1, dtype=torch.int64, device=device)
    if len(local_sizes):
        mine[: len(local_sizes)] = torch.as_tensor(list(local_sizes), dtype=torch.int64, device=device)
    if world_size == 1:
        gathered = [mine]
    else:
        gathered = [torch.empty_like(mine) for _ in range(world_size)]
        dist.all_gather(gathered, mine)
    sizes = np.full(n_clouds, -1, dtype=np.int64)
    for r in range(world_size):
        owned = shard_clouds(n_clouds, world_size, r)
        vals = gathered[r].cpu().numpy()
        sizes[owned] = vals[: len(owned)]
    if (sizes < 0).any():
        raise RuntimeError("size exchange incomplete")
    return sizes


def encode_sharded(clouds: Sequence[np.ndarray], encode_fn, rank: int, world_size: int, device=None):
    """Encode this rank's share of `clouds` with encode_fn(list_of_clouds) -> list_of_streams and return
    (owned indices, their streams, sizes of ALL clouds, offsets of ALL clouds in batch order, total bytes)."""
    owned = shard_clouds(len(clouds), world_size, rank)
    streams = encode_fn([clouds[k] for k in owned]) if owned else []
    sizes = exchange_sizes([len(s) for s in streams], len(clouds), rank, world_size, device)
    offsets, total = batch_layout(sizes)
    return owned, streams, sizes, offsets, total


# ---- one large cloud, split by chunk ranges -----------------------------------------------------------------

POINTS_PER_CHUNK = 32768  # kPointsPerChunk, src/codec_common.hpp:28
PROBE_POINTS = 4096       # mode decision window, src/v5_codec.cpp:934-949


def shard_chunks(n_points: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous range of whole chunks per rank: returns (first_point, n_points) of this rank's share."""
    n_chunks = (n_points + POINTS_PER_CHUNK - 1) // POINTS_PER_CHUNK
    per = (n_chunks + world_size - 1) // world_size
    c0 = min(n_chunks, rank * per)
    c1 = min(n_chunks, c0 + per)
    p0 = c0 * POINTS_PER_CHUNK
    p1 = min(n_points, c1 * POINTS_PER_CHUNK)
    return p0, max(0, p1 - p0)


def broadcast_modes(modes, n_modes: int, src: int, rank: int, world_size: int, device=None) -> np.ndarray:
    """The one-to-all exchange of the committed adaptive-int modes (n_modes bytes; 0 -> nothing to send)."""
    import torch
    import torch.distributed as dist

    if n_modes == 0:
        return np.zeros(0, dtype=np.uint8)
    t = torch.zeros(n_modes, dtype=torch.uint8, device=device)
    if rank == src:
        t.copy_(torch.as_tensor(np.asarray(modes, dtype=np.uint8)[:n_modes]))
    if world_size > 1:
        dist.broadcast(t, src=src)
    return t.cpu().numpy()


def exchange_part_sizes(local_size: int, rank: int, world_size: int, device=None) -> np.ndarray:
    """All-gather of the bytes every rank produced -> sizes[world_size] (exclusive scan = each rank's offset)."""
    import torch
    import torch.distributed as dist

    mine = torch.tensor([int(local_size)], dtype=torch.int64, device=device)
    if world_size == 1:
        return mine.cpu().numpy()
    gathered = [torch.empty_like(mine) for _ in range(world_size)]
    dist.all_gather(gathered, mine)
    return np.array([int(g.item()) for g in gathered], dtype=np.int64)


def encode_cloud_sharded(cloud: np.ndarray, point_step: int, n_modes: int, probe_fn, encode_part_fn, rank: int,
                         world_size: int, device=None):
    """Stage 1 of ONE cloud over `world_size` ranks.

    probe_fn(head_points_u8) -> modes           run on rank 0 over the cloud's first min(n, 4096) points
    encode_part_fn(part_points_u8, modes) -> framed chunks of that range (modes empty when the schema has none)

    Returns (this rank's bytes, its byte offset in the cloud's stream, total stream bytes, modes). In a real
    deployment each rank only holds its own range (plus, on rank 0, the head); `cloud` is indexed by range here.
    """
    data = np.ascontiguousarray(cloud).view(np.uint8).reshape(-1)
    n_points = data.size // point_step
    modes = None
    if n_modes and rank == 0:
        head = min(n_points, PROBE_POINTS)
        modes = probe_fn(data[: head * point_step])
    modes = broadcast_modes(modes, n_modes, 0, rank, world_size, device)
    p0, cnt = shard_chunks(n_points, world_size, rank)
    part = encode_part_fn(data[p0 * point_step:(p0 + cnt) * point_step], modes) if cnt else np.zeros(0, np.uint8)
    sizes = exchange_part_sizes(len(part), rank, world_size, device)
    offsets, total = batch_layout(sizes)
    return part, int(offsets[rank]), total, modes
