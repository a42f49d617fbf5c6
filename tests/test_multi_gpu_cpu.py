"""The N>1 path on CPU: two processes over gloo exercise the per-epoch best-slab exchange
(megalania_amd/multi_gpu.py) that bench.py runs over RCCL.  The chains here are stand-ins
holding a slab and a cost; on the GPU the same object is binding.SA."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from megalania_amd import binding, multi_gpu


class FakeChain:
    """packed device form: dist | len << 32 | type << 48, one u64 per position"""

    def __init__(self, n, cost, fill):
        self.n = n
        self.words = np.full(n, (1 << 48) | (1 << 32) | fill, dtype=np.uint64)
        self.cost = cost
        self.adopted = None

    def best_cost(self):
        return self.cost

    def best_packed(self):
        return self.words, self.cost

    def adopt_best_packed(self, words, cost):
        self.words, self.cost, self.adopted = words.copy(), cost, cost


def _worker(rank, world, port, costs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chain = FakeChain(257, costs[rank], fill=rank + 1)
    winner, wcost = multi_gpu.exchange_best(chain, dist)
    assert chain.words.dtype == np.uint64 and len(chain.words) == 257 and (chain.words >> np.uint64(32) == (1 << 16) | 1).all()
    out.put((rank, winner, wcost, chain.cost, int(chain.words[0] & np.uint64(0xFFFFFFFF)), chain.adopted))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("costs,winner", [((5000, 3000), 1), ((700, 900), 0), ((0, 4000), 1), ((0, 0), None)])
def test_exchange_best_two_ranks(costs, winner):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, costs, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if winner is None:
        assert all(r[2] == 0 and r[5] is None for r in res)  # nobody has a best slab yet
        return
    wcost = costs[winner]
    for rank, w, c, mine, fill, adopted in res:
        assert w == winner and c == wcost
        assert mine == wcost and fill == winner + 1  # everybody ends with the winner's slab
        assert (adopted == wcost) == (rank != winner)


def test_chain_seeds_distinct():
    seeds = {multi_gpu.chain_seed(1673551, r) for r in range(8)}
    assert len(seeds) == 8 and multi_gpu.chain_seed(1673551, 0) == 1673551


def test_pack_key_orders_by_cost_then_rank():
    assert multi_gpu.pack_key(10, 7) < multi_gpu.pack_key(11, 0)
    assert multi_gpu.pack_key(10, 1) < multi_gpu.pack_key(10, 2)
    assert multi_gpu.pack_key(0, 0) > multi_gpu.pack_key(1 << 43, 255)  # "no cost yet" never wins
