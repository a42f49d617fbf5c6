"""Two real chains (binding.SA, both on GPU 0) in two processes: the per-epoch best-slab exchange of
megalania_amd/multi_gpu.py over gloo, exactly as bench.py runs it over RCCL with one GPU per rank
(SURVEY 8e).  `-m gpu`; two processes share the card."""
import lzma
import os
import socket

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MGL_NO_AUTOBUILD"] = "1"
    import torch.distributed as dist
    from megalania_amd import binding, corpus, multi_gpu

    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = corpus.enwik_like(4000, 0x51)
    sa = binding.SA(data, accept="single", neighbours_per_step=256, seed=multi_gpu.chain_seed(1673551, rank), iters_per_epoch=len(data))
    sa.run(10 + 30 * rank)  # rank 1 searches longer: it should win
    _, mine = sa.best()
    winner, wcost = multi_gpu.exchange_best(sa, dist)
    slab, cost = sa.best()
    ok = lzma.decompress(binding.emit_stream(data, slab), format=lzma.FORMAT_ALONE) == data
    # next epoch from the common best slab (main.c:75-77)
    sa.begin_epoch(1, from_best=True)
    st = sa.run(5)
    out.put((rank, winner, wcost, mine, cost, ok, st["best_cost"] <= wcost, st["steps"]))
    sa.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_chains_exchange_best_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, w0, c0, mine0, best0, ok0, cont0, n0), (_, w1, c1, mine1, best1, ok1, cont1, n1) = res
    assert w0 == w1 and c0 == c1 == min(mine0, mine1)
    assert best0 == best1 == c0              # the loser adopted the winner's slab, re-costed on its own device
    assert ok0 and ok1 and cont0 and cont1 and n0 == n1 == 5
