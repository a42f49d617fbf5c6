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
    """One exchange epoch.  `chain` offers best() -> (slab, cost) and set_best(slab, cost);
    `dist` is torch.distributed (already initialised).  Returns (winner_rank, winner_cost)."""
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    slab, cost = chain.best()
    key = torch.tensor([pack_key(cost, rank)], dtype=torch.int64, device=device)
    dist.all_reduce(key, op=dist.ReduceOp.MIN)
    k = int(key.item())
    winner, wcost = k & 0xFF, k >> 8
    if wcost == (1 << 54) - 1:
        return winner, 0  # nobody has a best slab yet
    if world == 1:
        return winner, wcost
    # winner's slab, field-wise as int32 words (type, dist, len): 12 bytes per position
    if rank == winner:
        words = np.stack([slab["type"].astype(np.int64), slab["dist"].astype(np.int64), slab["len"].astype(np.int64)])
        buf = torch.from_numpy(words.astype(np.int64))
    else:
        buf = torch.empty((3, len(slab)), dtype=torch.int64)
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=winner)
    if rank != winner and (cost == 0 or wcost < cost):
        words = buf.cpu().numpy()
        new = np.zeros(len(slab), dtype=slab.dtype)
        new["type"], new["dist"], new["len"] = words[0], words[1], words[2]
        chain.set_best(new, wcost)
    return winner, wcost
