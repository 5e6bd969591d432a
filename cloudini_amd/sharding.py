"""Batch sharding across GPUs: whole clouds are independent units -- the encoder state never
crosses clouds and the adaptive-int modes are decided per encode() call (src/v5_codec.cpp:904) -- so a batch is
split by cloud, every rank runs the single-GPU path on its share, and the only exchange is an all-gather of the
per-cloud encoded sizes (so that any rank can lay the batch out / write an index). No payload crosses xGMI.

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
