"""Greedy seeding (`mgl_sa_seed_greedy`): an opt-in starting slab that is not in the
reference -- parity unpinned by the reference, pinned here by a restatement of the rule in plain
Python, by the oracle's costing of the seeded slab, by the oracle's batched SA continuing from it,
and by liblzma decoding the stream.  `-m gpu`."""
import lzma
import subprocess

import numpy as np
import pytest

from _libs import Oracle, literal_slab
from megalania_amd import binding, build, corpus

pytestmark = pytest.mark.gpu

LIT, MATCH = 1, 2


def greedy_rule(data: bytes, cand: int, dict_limit: int = 0x400000):
    """The rule of mgl_index.hip:k_greedy_seed, position by position: candidates = the `cand`
    nearest earlier positions with the same 2 bytes + the `cand` nearest with the same 4 bytes,
    inside the window; longest wins, nearest among equals; short matches only when near; and a
    literal instead when the next position starts a longer match (lazy step)."""
    n = len(data)
    best = [(0, 0)] * n  # (length, distance) of the match each position would take, (0, 0) = none
    by2, by4 = {}, {}
    for p in range(n):
        if 0 < p < n - 1:
            maxlen = min(273, n - p)
            cands = list(reversed(by2.get(data[p:p + 2], [])))[:cand]
            if maxlen >= 4:
                cands += list(reversed(by4.get(data[p:p + 4], [])))[:cand]
            best_len, best_q = 0, 0
            for q in cands:
                if p - q - 1 >= dict_limit:
                    continue
                ln = 0
                while ln < maxlen and data[q + ln] == data[p + ln]:
                    ln += 1
                if ln > best_len or (ln == best_len and q > best_q):
                    best_len, best_q = ln, q
            dist = p - best_q
            if best_len >= 4 or (best_len == 3 and dist <= 1 << 14) or (best_len == 2 and dist <= 128):
                best[p] = (best_len, dist)
        # the index holds every position that has a following byte (substring_enumerator.c:39-46)
        if p + 1 < n:
            by2.setdefault(data[p:p + 2], []).append(p)
        if p + 3 < n:
            by4.setdefault(data[p:p + 4], []).append(p)
    out = [(LIT, 0, 1)] * n
    for p in range(n):
        ln, dist = best[p]
        nxt = best[p + 1][0] if p + 2 < n else 0
        if ln and nxt <= ln:
            out[p] = (MATCH, dist - 1, ln)
    return out


def as_list(slab):
    return [(int(t), int(d), int(l)) for t, d, l in zip(slab["type"], slab["dist"], slab["len"])]


@pytest.mark.parametrize("name,data,cand", [
    ("lorem", corpus.lorem(4096), 8),
    ("enwik", corpus.enwik_like(20000, 0x31), 4),
    ("enwik_wide", corpus.enwik_like(6000, 0x32), 256),
    ("runs", b"a" * 700 + b"ab" * 300 + bytes(range(256)) + b"a" * 50, 3),
    ("tiny", b"abcabcabc", 16),
    ("n2", b"ab", 16),
    ("n1", b"x", 16),
])
def test_greedy_slab_matches_the_rule(name, data, cand):
    sa = binding.SA(data, accept="single", neighbours_per_step=16)
    sa.seed_greedy(cand)
    cur, cost = sa.current()
    assert as_list(cur) == greedy_rule(data, cand), name
    # the slab is a valid parse with exactly the cost the oracle gives it, and it decodes
    o = Oracle(data, dict_limit=0x400000)
    slab = np.ascontiguousarray(cur).astype(literal_slab(1).dtype)
    assert cost == o.cost_slab(slab)["total"]
    assert lzma.decompress(binding.emit_stream(data, cur), format=lzma.FORMAT_ALONE) == data
    sa.close()


def test_greedy_tail_entries_with_short_lookahead():
    """The last positions cannot hold 4 bytes: only the 2-byte source is consulted there."""
    data = b"abcdabcdab"
    sa = binding.SA(data, accept="single", neighbours_per_step=16)
    sa.seed_greedy(64)
    cur, _ = sa.current()
    got = as_list(cur)
    assert got == greedy_rule(data, 64)
    assert got[4] == (MATCH, 3, 6) and got[8] == (MATCH, 3, 2) and got[9] == (LIT, 0, 1)


def test_greedy_lazy_step():
    """'bcde' occurs earlier and so does 'abc': at the 'a' the next position starts a longer match,
    so the 'a' stays a literal."""
    data = b"abcX" + b"bcdefgY" + b"abcdefg"
    sa = binding.SA(data, accept="single", neighbours_per_step=16)
    sa.seed_greedy(64)
    cur, _ = sa.current()
    got = as_list(cur)
    assert got == greedy_rule(data, 64)
    assert got[11] == (LIT, 0, 1) and got[12] == (MATCH, 7, 6)
    sa.close()
    sa.close()


def test_search_continues_from_the_seed_like_the_oracle():
    """After seeding, mgl_sa_run is the oracle's batched SA started from that slab."""
    data = corpus.enwik_like(3000, 0x33)
    n, K, seed, steps = len(data), 64, 99, 40
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, iters_per_epoch=steps)
    sa.seed_greedy(32)
    cur, _ = sa.current()
    o = Oracle(data, dict_limit=0x400000)
    slab = np.ascontiguousarray(cur).astype(literal_slab(1).dtype)
    best = literal_slab(n)
    ref = o.sa_batched(slab, best, 0, 0, seed, K, 0, steps, 0, steps)
    for s in range(steps):
        st = sa.run(1)
        assert st["current_cost"] == int(ref["trace"][s, 3]), s
    got, got_cost = sa.current()
    assert got_cost == ref["cur"] and as_list(got) == as_list(slab)
    bst, best_cost = sa.best()
    assert best_cost == ref["best"] and as_list(bst) == as_list(best)
    sa.close()


def test_greedy_seed_beats_the_literal_start_at_equal_budget():
    data = corpus.enwik_like(30000, 0x34)
    res = {}
    for greedy in (False, True):
        sa = binding.SA(data, accept="single", neighbours_per_step=1024, seed=3)
        if greedy:
            sa.seed_greedy(256)
        st = sa.run(300)
        res[greedy] = st["best_cost"]
        bst, _ = sa.best()
        assert lzma.decompress(binding.emit_stream(data, bst), format=lzma.FORMAT_ALONE) == data
        sa.close()
    assert res[True] < res[False]


def test_cli_greedy_seed(tmp_path):
    data = corpus.enwik_like(5000, 0x35)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    outs = {}
    for extra in ([], ["--greedy-seed", "128"]):
        r = subprocess.run([build.CLI, "--epochs", "2", "--phases", "1", "--neighbours", "256"] + extra + [str(f)],
                           capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-400:]
        assert lzma.decompress(r.stdout, format=lzma.FORMAT_ALONE) == data
        outs[bool(extra)] = len(r.stdout)
    # round 2: with bulk steps the plain start catches up with the seed within two epochs; the seed must not hurt
    assert outs[True] <= outs[False] * 1.02


def test_metropolis_rule_trajectory_vs_oracle():
    """mgl_sa_set_temperature (opt-in, not in the reference): step by step the same decisions as the
    oracle's batched SA with the same temperature -- and a different trajectory from the reference's
    rule, with steps that accept a worse neighbour."""
    data = corpus.enwik_like(2500, 0x36)
    n, K, seed, steps = len(data), 32, 11, 120
    temp = 3 * 16384  # three bytes of slack at the start of the epoch
    o = Oracle(data, dict_limit=0x400000)
    runs = {}
    for t in (0, temp):
        sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, iters_per_epoch=steps * K)
        sa.set_temperature(t)
        o.set_temperature(t)
        slab, best = literal_slab(n), literal_slab(n)
        ref = o.sa_batched(slab, best, 0, 0, seed, K, 0, steps * K, 0, steps)
        costs = []
        for s in range(steps):
            st = sa.run(1)
            assert st["current_cost"] == int(ref["trace"][s, 3]), (t, s)
            costs.append(st["current_cost"])
        cur, cur_cost = sa.current()
        bst, best_cost = sa.best()
        assert cur_cost == ref["cur"] and best_cost == ref["best"]
        assert as_list(cur) == as_list(slab) and as_list(bst) == as_list(best)
        assert lzma.decompress(binding.emit_stream(data, bst), format=lzma.FORMAT_ALONE) == data
        runs[t] = costs
        sa.close()
    o.set_temperature(0)
    worse = sum(1 for a, b in zip(runs[temp], runs[temp][1:]) if b > a)
    assert worse > 0 and runs[temp] != runs[0]
    with pytest.raises(binding.MglError):
        sa = binding.SA(data, accept="single", neighbours_per_step=K)
        try:
            sa.set_temperature(1 << 40)
        finally:
            sa.close()


def test_cli_temperature(tmp_path):
    data = corpus.enwik_like(4000, 0x37)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    r = subprocess.run([build.CLI, "--epochs", "2", "--phases", "2", "--neighbours", "256", "--temperature", "1.5", "--greedy-seed", "64", str(f)],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert lzma.decompress(r.stdout, format=lzma.FORMAT_ALONE) == data


@pytest.mark.parametrize("kw", [dict(fullwalk=True), dict(snapshots=False), dict(serial_build=True), dict(lc=3, pb=2)],
                         ids=["fullwalk", "no_snapshots", "serial_build", "lc3pb2"])
def test_greedy_seed_under_every_engine_option(kw):
    """The seed and the search from it do not depend on the engine options (and work with lc/pb != 0):
    same slab, same costs as the default engine, step by step; epochs restart from the best slab."""
    data = corpus.enwik_like(5000, 0x38)
    props = {k: v for k, v in kw.items() if k in ("lc", "pb")}
    ref = binding.SA(data, accept="single", neighbours_per_step=96, seed=21, iters_per_epoch=50, **props)
    sa = binding.SA(data, accept="single", neighbours_per_step=96, seed=21, iters_per_epoch=50, **kw)
    for s in (ref, sa):
        s.seed_greedy(48)
    a, ca = ref.current()
    b, cb = sa.current()
    assert ca == cb and as_list(a) == as_list(b) == greedy_rule(data, 48)
    for _ in range(3):
        assert ref.run(10)["current_cost"] == sa.run(10)["current_cost"]
    for s in (ref, sa):
        s.begin_epoch(1, from_best=True)
    assert ref.run(10)["best_cost"] == sa.run(10)["best_cost"]
    bst, _ = sa.best()
    assert lzma.decompress(binding.emit_stream(data, bst, **props), format=lzma.FORMAT_ALONE) == data
    ref.close()
    sa.close()
