"""The CPU oracle against the reference itself (oracle/_ref/libmegalania_ref.so, compiled from
/root/reference by `make -C oracle`).  Skipped where that build is absent; the committed
fixtures (test_oracle_golden.py) carry the same pinning everywhere else."""
import numpy as np
import pytest

from _libs import Oracle, Ref, literal_slab, walk
from conftest import rand_bytes
from megalania_amd import corpus

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (no /root/reference here)")


def _evolve(data, iters, seed):
    """A valid, SA-shaped slab made by the reference itself."""
    n = len(data)
    r = Ref(data)
    slab, best = literal_slab(n), literal_slab(n)
    Ref.lib().ref_srand(seed)
    r.sa_iters(slab, best, 0, 0, 0, n, 0, iters)
    return slab


CASES = [
    ("lorem2k", corpus.lorem(2048), 500),
    ("enwik2k", corpus.enwik_like(2048, 0x51), 400),
    ("rand1k", rand_bytes(1024, 3), 200),
    ("zeros", b"\0" * 700, 150),
    ("two", b"ab", 0),  # SA on it would spin forever in the reference too (main.c:81-84: no neighbour exists)
    ("one", b"x", 0),
]


def test_layout_constants():
    L = Ref.lib()
    assert L.ref_sizeof_packet() == 12 and L.ref_num_probs() == 2615 and L.ref_sizeof_state() == 5280
    assert Oracle(b"abc").nprobs == 2615


@pytest.mark.parametrize("name,data,iters", CASES, ids=[c[0] for c in CASES])
def test_walk_topk_emit(name, data, iters):
    o, r = Oracle(data), Ref(data)
    slab = _evolve(data, iters, 1234) if iters else literal_slab(len(data))
    a, b = o.cost_slab(slab, True), r.cost_slab(slab, True)
    assert a["total"] == b["total"] and (a["cum"] == b["cum"]).all()
    assert (a["probs"] == b["probs"]).all() and a["ctx_state"] == b["ctx_state"] and (a["dists"] == b["dists"]).all()
    assert o.emit(slab) == r.emit(slab)
    w = walk(slab)
    rng = np.random.default_rng(5)
    for pos in sorted(set([w[0], w[-1]] + [w[i] for i in rng.integers(0, len(w), 12)])):
        pa, ca = o.top_k(slab, pos, mode=0)
        pb, cb = r.top_k(slab, pos)
        assert (pa == pb).all() and (ca == cb).all(), (name, pos)
    for pos in sorted(set(int(x) for x in rng.integers(0, len(data), 16)) | {0, len(data) - 1}):
        for ml in (273, 3):
            oa, la = o.substrings(pos, ml)
            ob, lb = r.substrings(pos, ml)
            assert (oa == ob).all() and (la == lb).all()


@pytest.mark.parametrize("step", [0, 1, 2])
def test_sa_trajectory(step):
    """Same glibc rand() stream, same costs, same accept decisions, same slabs."""
    data = corpus.enwik_like(1500, 0x99)
    n = len(data)
    o, r = Oracle(data), Ref(data)
    start = _evolve(data, 200, 42) if step else literal_slab(n)
    out = []
    for lib, seed_fn in ((o, Oracle.lib().orc_srand), (r, Ref.lib().ref_srand)):
        slab, best = start.copy(), start.copy()
        seed_fn(1673551 + step)
        res = lib.sa_iters(slab, best, 0, 0, step, n, 0, 350)
        # continue from the returned state, as successive calls must compose
        res2 = lib.sa_iters(slab, best, res["cur"], res["best"], step, n, 350, 500)
        out.append((slab, best, res["trace"], res2["trace"], res2["cur"], res2["best"], res["undo"] + res2["undo"]))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)
