"""Chain-level sharding across GPUs: one independent SA chain per
GPU/process, distinct RNG streams, and one small exchange per epoch.

The only data-path collective is the one north_star names: a single all-reduce(MIN) of the
packed (best_cost << 8 | rank) word over RCCL/xGMI (8 bytes, latency-bound); the winning
rank then broadcasts its best slab so every chain can continue from it (main.c:75-77 starts
later epochs from the best slab).  torch.distributed is used as plumbing only: backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def chain_seed(seed: int, rank: int) -> int:
    """Distinct, reproducible RNG stream per chain."""
    return (seed ^ (0x9E3779B97F4A7C15 * (rank + 1))) & 0xFFFFFFFFFFFFFFFF if rank else seed


def pack_key(best_cost: int, rank: int) -> int:
    # perplexities stay below 2^44 (enwik8 all-literal is ~1.6e12 < 2^41); 0 means "none yet"
    cost = best_cost if best_cost else (1 << 54) - 1
    return (cost << 8) | (rank & 0xFF)


def exchange_best(chain, dist, device=None):
    """One exchange epoch over torch.distributed (any backend; the CPU tests use gloo).  `chain` offers
    best_cost(), best_packed() -> (u64 words, cost) and adopt_best_packed(words, cost).  The slab travels in
    its packed device form, 8 bytes per position.  Returns (winner_rank, winner_cost).
    On GPUs the native path is `exchange_best_native` (RCCL from the C library, HBM to HBM)."""
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    cost = chain.best_cost()
    key = torch.tensor([pack_key(cost, rank)], dtype=torch.int64, device=device)
    dist.all_reduce(key, op=dist.ReduceOp.MIN)
    k = int(key.item())
    winner, wcost = k & 0xFF, k >> 8
    if wcost == (1 << 54) - 1:
        return winner, 0  # nobody has a best slab yet
    if world == 1:
        return winner, wcost
    if rank == winner:
        words, _ = chain.best_packed()
        buf = torch.from_numpy(words.view(np.int64).copy())
    else:
        buf = torch.empty(chain.n, dtype=torch.int64)
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=winner)
    if rank != winner and (cost == 0 or wcost < cost):
        chain.adopt_best_packed(buf.cpu().numpy().view(np.uint64), wcost)
    return winner, wcost


def make_comm(dist, rank: int, world: int, device: int):
    """An RCCL communicator of the C library for this process group: rank 0 draws the id, torch.distributed
    (whatever its backend) only carries those 128 bytes."""
    from . import binding

    box = [binding.Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return binding.Comm(box[0], rank, world, device)


def exchange_best_native(chain, comm):
    """The north-star exchange, entirely inside the C library: 8-byte ncclAllReduce(min) + ncclBroadcast of
    the packed slab from the winner's HBM to the others'."""
    return chain.exchange_best(comm)
