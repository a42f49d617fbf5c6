"""The native exchange (mgl_sa_exchange_best: RCCL from the C library).  A one-GPU box only allows a
communicator of one rank -- RCCL refuses two ranks on one device -- so this covers library loading,
communicator set-up, the 8-byte all-reduce and the adoption rules; the two-rank protocol itself is covered
over gloo (tests/test_multi_gpu_cpu.py, tests/test_gpu_two_chains.py).  N = 2/4/8 on real GPUs is the
driver's scaling run: unmeasured here."""
import lzma
import subprocess

import numpy as np
import pytest

from megalania_amd import binding, build, corpus

pytestmark = pytest.mark.gpu


def test_exchange_world_of_one_over_rccl():
    data = corpus.enwik_like(5000, 0x61)
    comm = binding.Comm(binding.Comm.unique_id(), 0, 1, 0)
    sa = binding.SA(data, neighbours_per_step=128, seed=3)
    assert sa.exchange_best(comm) == (0, 0)  # nobody has a best slab yet
    st = sa.run(20)
    assert sa.exchange_best(comm) == (0, st["best_cost"])
    words, cost = sa.best_packed()
    slab, cost2 = sa.best()
    assert cost == cost2 == st["best_cost"]
    assert ((words & np.uint64(0xFFFFFFFF)) == slab["dist"]).all() and ((words >> np.uint64(48)) == slab["type"]).all()
    sa.close()
    comm.close()


def test_adopted_slab_is_verified_when_an_epoch_starts_from_it():
    data = corpus.enwik_like(5000, 0x62)
    a = binding.SA(data, neighbours_per_step=128, seed=3)
    b = binding.SA(data, neighbours_per_step=128, seed=4)
    a.run(30)
    words, cost = a.best_packed()
    b.adopt_best_packed(words, cost)
    b.begin_epoch(1, from_best=True)  # re-derived, every packet checked, cost compared
    st = b.run(3)
    assert 0 < st["best_cost"] <= cost
    # a slab that does not reproduce the input, or comes with the wrong cost, is refused at that point
    bad = words.copy()
    pos = int(np.nonzero((bad >> np.uint64(48)) == 2)[0][0])  # a MATCH: point it somewhere else
    bad[pos] = (bad[pos] & ~np.uint64(0xFFFFFFFF)) | np.uint64((int(bad[pos]) & 0xFFFFFFFF) ^ 1)
    for w, c in ((bad, cost), (words, cost + 1)):
        c2 = binding.SA(data, neighbours_per_step=128, seed=5)
        c2.adopt_best_packed(w, c)
        with pytest.raises(binding.MglError):
            c2.begin_epoch(1, from_best=True)
        c2.close()
    a.close()
    b.close()


def test_cli_single_chain_with_communicator(tmp_path):
    """--chains 1 goes through the whole multi-chain code path of the C driver (id file, communicator,
    exchange after every epoch) on the one GPU there is."""
    data = corpus.enwik_like(3000, 0x63)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    r = subprocess.run([build.CLI, "--epochs", "2", "--phases", "2", "--neighbours", "128", "--chains", "1", "--rank", "0",
                        "--comm-file", str(tmp_path / "comm.id"), str(f)], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    assert b"exchange: chain 0 holds the best slab" in r.stderr
    assert lzma.decompress(r.stdout, format=lzma.FORMAT_ALONE) == data
