"""The CPU oracle against the committed fixtures (tests/golden/reference_vectors.json), whose
expected values were all produced by the compiled reference (tools/make_golden.py).
Runs everywhere (no GPU, no /root/reference)."""
import lzma
import re

import numpy as np
import pytest

from _libs import Oracle, literal_slab
from conftest import sha, sha_slab, slab_from_rle


def test_cost_table_values():
    """perplexity_table.h:4 / generate_table.py:7-10: T[0]=0, T[i]=-int(log2(i/2048)*2048)."""
    t = Oracle.cost_table()
    assert t[0] == 0 and t[1] == 22528 and t[2] == 20480 and t[3] == 19281 and t[4] == 18432
    assert t[1024] == 2048 and t[2047] == 1 and t[2046] == 2
    # exact integer restatement: T[i] = floor(2048*(11 - log2 i))  <=>  i**2048 <= 2**(22528 - T)
    for i in (1, 2, 3, 5, 31, 100, 1023, 1025, 2017, 2047):
        m = i ** 2048
        bl = m.bit_length()
        ceil_log2 = bl - 1 if m & (m - 1) == 0 else bl
        assert int(t[i]) == 22528 - ceil_log2, i


def test_golden_walks(golden, golden_input):
    for w in golden["walks"]:
        data = golden_input(w["input"])
        o = Oracle(data)
        slab = slab_from_rle(len(data), w["packets"])
        r = o.cost_slab(slab, want_probs=True)
        assert r["total"] == w["total"], w["name"]
        assert len(r["cum"]) == w["npackets"], w["name"]
        assert sha(r["cum"]) == w["cum_sha256"], w["name"]
        if "cum" in w:
            assert [int(x) for x in r["cum"]] == w["cum"], w["name"]
        assert r["ctx_state"] == w["ctx_state"], w["name"]
        assert [int(x) for x in r["dists"]] == w["dists"], w["name"]
        assert sha(r["probs"]) == w["probs_sha256"], w["name"]
        stream = o.emit(slab)
        assert len(stream) == w["stream_len"], w["name"]
        assert sha(np.frombuffer(stream, dtype=np.uint8)) == w["stream_sha256"], w["name"]
        if "stream_hex" in w:
            assert stream.hex() == w["stream_hex"]
        # the stream is a real .lzma file: liblzma must give the input back
        assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data, w["name"]


def test_golden_known_answers(golden):
    """The four hand-checked vectors recorded in SURVEY.md section 8c."""
    by = {w["name"]: w for w in golden["walks"]}
    assert by["hello_literals"]["cum"] == [18432, 36502, 54144, 71017, 87706, 105797, 121915, 138047, 153181,
                                           167693, 182379]
    assert by["hello_literals"]["stream_hex"] == "00" "00004000" "0b00000000000000" "00341a7248012c5aaedeccf5c2442fa0"
    assert by["hello_match"]["cum"][-1] == 132981 and by["hello_match"]["stream_hex"].endswith("00341a7248012c61fe709c0000")
    assert by["abab"]["cum"] == [18432, 36320, 61083, 77467] and by["abab"]["stream_hex"].endswith("003099c8d293f88000")
    assert by["aaab"]["cum"] == [18432, 43101, 61533, 79237, 87429, 103813]
    assert by["lorem4k_literals"]["total"] == 37054336


def test_golden_sa_trajectories(golden, golden_input):
    """main.c:78-102 replayed under glibc rand(): every neighbour cost, every accept decision,
    the final slab (stale entries included) and the best slab must match the reference."""
    for s in golden["sa"]:
        data = golden_input(s["input"])
        n = len(data)
        o = Oracle(data)
        slab, best = literal_slab(n), literal_slab(n)
        Oracle.lib().orc_srand(s["seed"])
        r = o.sa_iters(slab, best, 0, 0, s["step"], s["num_iters"], 0, s["iters"])
        assert [int(x) for x in r["trace"][:, 0]] == s["trace_cost"], s["input"]
        assert [int(x) for x in r["trace"][:, 1]] == s["trace_accept"], s["input"]
        assert r["cur"] == s["cur"] and r["best"] == s["best"] and r["undo"] == s["undo_total"]
        assert sha_slab(slab) == s["slab_sha256"] and sha_slab(best) == s["best_sha256"]


def test_golden_topk(golden, golden_input):
    """top_k_packet_finder.c:120-138: same 20 packets in the same pop order (worst first)."""
    for t in golden["topk"]:
        data = golden_input(t["input"])
        n = len(data)
        o = Oracle(data)
        slab = literal_slab(n) if t["slab"] == "literal" else slab_from_rle(n, golden["evolved_walks"][t["input"]])
        pk, costs = o.top_k(slab, t["position"], mode=0, k=20)
        got = [[int(p["type"]), int(p["dist"]), int(p["len"])] for p in pk]
        want = []
        for p in t["packets"]:
            want.extend([[1, 0, 1]] * p[1] if p[0] == "L" else [p])
        assert got == want, (t["input"], t["position"])
        assert [int(c) for c in costs] == t["costs"]
        # canonical mode: same cost multiset, same best cost (ties are heap-internal)
        pk2, costs2 = o.top_k(slab, t["position"], mode=1, k=20)
        assert sorted(int(c) for c in costs2) == sorted(t["costs"])
        assert list(costs2) == sorted(costs2, reverse=True)


def test_golden_substrings(golden, golden_input):
    """substring_enumerator.c:85-105, incl. the reference's own test expectations
    (tests/substring_enumerator_test.c:36-37,58-59,83-97)."""
    for s in golden["substrings"]:
        data = golden_input(s["input"])
        o = Oracle(data)
        for q in s["queries"]:
            offs, lens = o.substrings(q["pos"], s["max_len"])
            assert len(offs) == q["count"], (s["input"], q["pos"])
            if "offs" in q:
                assert [int(x) for x in offs] == q["offs"] and [int(x) for x in lens] == q["lens"]
            else:
                assert sha(np.stack([offs, lens])) == q["sha256"]
    hello = Oracle(b"hello hello")
    assert [len(hello.substrings(i)[0]) for i in range(11)] == [0, 0, 0, 0, 0, 0, 4, 3, 2, 1, 0]
    assert [len(hello.substrings(i, 3)[0]) for i in range(11)] == [0, 0, 0, 0, 0, 0, 2, 2, 2, 1, 0]
    aabbcc = Oracle(b"aa bb cc")
    assert all(len(aabbcc.substrings(i)[0]) == 0 for i in range(8))


def test_batched_semantics_selfconsistent():
    """orc_neighbour leaves the slab untouched unless asked to keep, its journal reproduces
    the kept slab, and its cost equals a from-scratch walk of that slab."""
    from megalania_amd import corpus
    data = corpus.lorem(1500)
    o = Oracle(data)
    slab = literal_slab(len(data))
    best = slab.copy()
    res = o.sa_batched(slab, best, 0, 0, seed=7, K=8, phase=0, iters_per_epoch=1500, step_begin=0, step_end=12)
    assert res["cur"] == o.cost_slab(slab)["total"]
    assert res["best"] == o.cost_slab(best)["total"] <= res["cur"]
    for j in range(16):
        before = slab.copy()
        ok, cost, diffs = o.neighbour(slab, 7, 99, j, keep=False, K=16)
        assert (slab == before).all()
        if not ok:
            continue
        kept = slab.copy()
        ok2, cost2, _ = o.neighbour(kept, 7, 99, j, keep=True, K=16)
        assert ok2 and cost2 == cost == o.cost_slab(kept)["total"]
        applied = before.copy()
        for d in diffs:
            assert (applied[d["position"]] == d["old"])
            applied[d["position"]] = d["new"]
        assert (applied == kept).all()
        assert lzma.decompress(o.emit(kept), format=lzma.FORMAT_ALONE) == data


def doubled_letters(seed, n):
    """words over a small alphabet with letters doubled at random: SHORT_REP packets everywhere, and repairs that turn them
    back into literals far behind the mutated packet (the case the soft window end has to cover)"""
    import random
    r = random.Random(seed)
    out = bytearray()
    words = [bytes(r.choice(b"abcdefgh") for _ in range(r.randint(2, 6))) for _ in range(12)]
    while len(out) < n:
        for ch in r.choice(words):
            out.append(ch)
            if r.random() < 0.45:
                out.append(ch)
        if r.random() < 0.3:
            out += b"  "
    return bytes(out[:n])


@pytest.mark.parametrize("name,seed", [("lorem", 1), ("lorem", 2), ("enwik", 3), ("enwik", 4), ("runs", 5), ("runs", 6), ("zeros", 7),
                                       ("doubled", 1), ("doubled", 5), ("doubled", 9)])
def test_bulk_steps_on_the_oracle_never_need_the_rollback(name, seed):
    """The bulk step's selection rests on a local argument about windows and rep distances; the
    oracle checks every combined parse and takes a failing step back as a whole, and counts slab entries that two taken
    journals of a step both write (the device writes them in parallel: with soft window ends that did not reach behind
    every *changed* packet the "doubled" inputs produce such entries, and the device's result depended on timing).  On repetitive, text-like and run-heavy
    inputs (rep packets everywhere) that net is never needed, the exact cost the step reports is a full walk's, and the
    stream decodes -- CPU only, so it runs wherever the tests run."""
    from megalania_amd import corpus
    data = {"lorem": corpus.lorem(2200), "enwik": corpus.enwik_like(3000, 0x77 + seed),
            "runs": b"a" * 300 + b"ab" * 200 + bytes(range(64)) * 3 + b"a" * 120 + b"abcabcabd" * 40,
            "zeros": bytes(900) + b"\x01\x02" * 50 + bytes(400), "doubled": doubled_letters(seed, 2600)}[name]
    n = len(data)
    o = Oracle(data)
    slab, best = literal_slab(n), literal_slab(n)
    before, over_before = o.bulk_rollbacks(), o.bulk_overlaps()
    K, steps = (96, 60) if name == "doubled" else (48, 36)
    cur = best_cost = 0
    taken = 0
    for s0 in range(0, steps, 6):
        res = o.sa_batched(slab, best, cur, best_cost, seed=(seed * 7717 if name == "doubled" else seed * 1000003), K=K, phase=0, iters_per_epoch=n,
                           step_begin=s0, step_end=s0 + 6, iter0=s0 * K, modes=[1] * 6)
        cur, best_cost = res["cur"], res["best"]
        taken += int(res["trace"][:, 1].sum())
        assert cur == o.cost_slab(slab)["total"]
        assert best_cost == o.cost_slab(best)["total"] <= cur
        assert lzma.decompress(o.emit(slab), format=lzma.FORMAT_ALONE) == data
    assert o.bulk_rollbacks() == before
    assert o.bulk_overlaps() == over_before  # no slab entry written by two taken journals (the device writes them in parallel)
    assert taken >= 12  # the steps did take moves (several per step on the text-like inputs)
