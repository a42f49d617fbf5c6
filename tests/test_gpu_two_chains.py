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


def _native_worker(rank, world, path, nonce, out):
    """The exchange entirely inside the C library (mgl_sa_exchange_best) over its host shared-memory transport:
    the all-reduce, the broadcast from the winner's device buffer and the adoption branch with world = 2 --
    the code RCCL runs through on a multi-GPU node, minus the two ncclXxx calls themselves."""
    os.environ["MGL_NO_AUTOBUILD"] = "1"
    os.environ["MGL_COMM_TIMEOUT_S"] = "120"
    import numpy as np
    from megalania_amd import binding, corpus, multi_gpu

    data = corpus.enwik_like(4000, 0x52)
    comm = binding.Comm.shm(path, nonce, rank, world, 0)
    sa = binding.SA(data, accept="single", neighbours_per_step=256, seed=multi_gpu.chain_seed(1673551, rank), iters_per_epoch=len(data))
    first = sa.exchange_best(comm)  # nobody has a best slab yet: no broadcast, same answer everywhere
    sa.run(10 + 30 * (1 - rank))  # rank 0 searches longer this time: it should win
    _, mine = sa.best()
    winner, wcost = sa.exchange_best(comm)
    words, cost = sa.best_packed()
    slab, cost2 = sa.best()
    ok = lzma.decompress(binding.emit_stream(data, slab), format=lzma.FORMAT_ALONE) == data
    sa.begin_epoch(1, from_best=True)  # the adopted slab is re-derived and verified here
    st = sa.run(5)
    again = sa.exchange_best(comm)  # a second exchange on the same communicator (the slab area is reused)
    out.put((rank, first, winner, wcost, mine, cost, cost2, ok, st["best_cost"] <= wcost, again,
             __import__("hashlib").sha256(np.ascontiguousarray(words).tobytes()).hexdigest()))
    sa.close()
    comm.close()


def test_two_chains_exchange_inside_the_c_library(tmp_path):
    path = "/dev/shm/mgl_test_two_chains_%d" % os.getpid() if os.path.isdir("/dev/shm") else str(tmp_path / "comm.shm")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, path, 0xABCD, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[1] == r1[1] == (0, 0)                                              # first exchange: no slab anywhere
    assert r0[2] == r1[2] and r0[3] == r1[3] == min(r0[4], r1[4])               # same winner, the cheaper slab's cost
    assert r0[5] == r1[5] == r0[6] == r1[6] == r0[3]                            # the loser adopted it ...
    assert r0[10] == r1[10]                                                      # ... bit for bit
    assert r0[7] and r1[7] and r0[8] and r1[8]
    assert r0[9] == r1[9] and r0[9][1] <= r0[3]
    assert not os.path.exists(path)


def test_cli_two_chains_on_one_gpu(tmp_path):
    """The C driver, two processes, --transport shm: rendezvous by nonce, an exchange after every epoch, rank 0 writes
    the common best slab's stream, each chain its own checkpoint."""
    import subprocess
    from megalania_amd import build, corpus

    data = corpus.enwik_like(3000, 0x64)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    comm = "/dev/shm/mgl_test_cli_%d" % os.getpid() if os.path.isdir("/dev/shm") else str(tmp_path / "comm.shm")
    with open(comm, "wb") as stale:  # a leftover of an "earlier run"
        stale.write(b"MGLCOMM1" + b"\0" * 200)
    env = dict(os.environ, MGL_COMM_TIMEOUT_S="120")
    cmd = [build.CLI, "--epochs", "2", "--phases", "2", "--neighbours", "128", "--chains", "2", "--device", "0", "--transport", "shm",
           "--comm-file", comm, "--comm-nonce", "424242", "--save-slab", str(tmp_path / "best.slab")]
    ps = [subprocess.Popen(cmd + ["--rank", str(r), str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in (1, 0)]
    outs = [p.communicate(timeout=600) for p in ps]
    assert all(p.returncode == 0 for p in ps), [o[1].decode()[-400:] for o in outs]
    ex = [[ln for ln in o[1].decode().splitlines() if ln.startswith("exchange:")] for o in outs]
    assert len(ex[0]) == 4 and ex[0] == ex[1]          # both chains saw the same winner and cost after every epoch
    assert outs[0][0] == b""                            # rank 1 writes no stream
    assert lzma.decompress(outs[1][0], format=lzma.FORMAT_ALONE) == data
    assert (tmp_path / "best.slab").exists() and (tmp_path / "best.slab.rank1").exists()
    assert not os.path.exists(comm)
