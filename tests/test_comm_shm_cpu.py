"""The C library's host shared-memory transport (mgl_comm_init_shm, include/megalania_hip.h) without a GPU: the
rendezvous (fresh file, nonce, stale files of earlier runs, removal), the barrier, and the first half of
mgl_sa_exchange_best -- the min over the ranks of the packed (best_cost << 8 | rank) key -- across two processes.
The second half (the slab broadcast and the adoption) needs chains, i.e. a GPU: tests/test_gpu_two_chains.py."""
import multiprocessing as mp
import os
import tempfile

import pytest

from megalania_amd import binding, multi_gpu


def _shm_dir():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()


def _rank(path, nonce, rank, world, costs, rounds, out):
    os.environ["MGL_COMM_TIMEOUT_S"] = "30"
    comm = binding.Comm.shm(path, nonce, rank, world)
    got = []
    for r in range(rounds):
        key = multi_gpu.pack_key(costs[r][rank], rank)
        got.append(comm.min_u64(key))
    comm.close()
    out.put((rank, got))


@pytest.mark.parametrize("world", [2, 3])
def test_min_of_packed_keys_across_processes(world):
    path = os.path.join(_shm_dir(), f"mgl_test_comm_{os.getpid()}_{world}")
    # a file an earlier (crashed) run left under the name, with another nonce: must not be mistaken for this run's
    with open(path, "wb") as f:
        f.write(b"\0" * 4096)
    costs = [(5000, 3000, 4000), (700, 900, 800), (0, 4000, 0), (0, 0, 0), (123456789012, 123456789011, 123456789013)]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(path, 0xC0FFEE + world, r, world, costs, len(costs), out)) for r in range(world)]
    # rank 1 first: it has to wait for rank 0's fresh file instead of taking the stale one
    for p in reversed(procs):
        p.start()
    res = dict(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, row in enumerate(costs):
        keys = [multi_gpu.pack_key(row[k], k) for k in range(world)]
        for k in range(world):
            assert res[k][r] == min(keys)
        winner, wcost = min(keys) & 0xFF, min(keys) >> 8
        live = [c for c in row[:world] if c]
        if live:
            assert wcost == min(live) and row[winner] == wcost
        else:
            assert wcost == (1 << 54) - 1  # nobody has a best slab yet
    assert not os.path.exists(path)  # rank 0 removes the name when it leaves


def _lonely(path, nonce, rank, world, out):
    os.environ["MGL_COMM_TIMEOUT_S"] = "1.5"
    try:
        binding.Comm.shm(path, nonce, rank, world)
        out.put("joined")
    except binding.MglError as e:
        out.put(str(e))


def test_a_file_of_another_run_is_refused_and_waits_time_out():
    path = os.path.join(_shm_dir(), f"mgl_test_comm_{os.getpid()}_stale")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    # rank 0 of run A (nonce 1) comes up and waits for its peer; rank 1 of run B (nonce 2) must not join it
    a = ctx.Process(target=_lonely, args=(path, 1, 0, 2, out))
    b = ctx.Process(target=_lonely, args=(path, 2, 1, 2, out))
    a.start(); b.start()
    msgs = [out.get(timeout=60), out.get(timeout=60)]
    a.join(30); b.join(30)
    assert "joined" not in msgs
    assert any("nonce" in m for m in msgs) and any("did not arrive" in m for m in msgs), msgs
    if os.path.exists(path):
        os.unlink(path)
