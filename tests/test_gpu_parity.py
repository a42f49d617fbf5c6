"""GPU parity tests proper: the HIP path, called through the C ABI, against
  (a) the committed reference fixtures (tests/golden, made by the compiled reference), and
  (b) the CPU oracle on the same seeded inputs,
bit-exact (integer path).  Run with `-m gpu` on an MI355X; nothing here reads /root/reference."""
import lzma

import numpy as np
import pytest

from _libs import DIFF as ODIFF
from _libs import Oracle, literal_slab, walk
from conftest import rand_bytes, sha, slab_from_rle
from megalania_amd import binding, corpus

pytestmark = pytest.mark.gpu

_SA = {}


def sa_for(name, data, **kw):
    key = (name, tuple(sorted(kw.items())))
    if key not in _SA:
        _SA[key] = binding.SA(data, **kw)
    return _SA[key]


def P(slab):
    return np.ascontiguousarray(slab).astype(binding.PACKET)


def as_list(pk):
    return [(int(p["type"]), int(p["dist"]), int(p["len"])) for p in pk]


def test_library_sees_gpu():
    lib = binding.hip_lib()
    assert lib.mgl_device_count() >= 1
    assert b"gfx950" in lib.mgl_version()


def test_cost_slab_against_reference_fixtures(golden, golden_input):
    """mgl_cost_slab / mgl_final_state == the reference's perplexity walk: total, every
    per-packet running total, the adapted probabilities, ctx_state and rep distances."""
    for w in golden["walks"]:
        data = golden_input(w["input"])
        sa = sa_for(w["input"], data, neighbours_per_step=8)
        slab = P(slab_from_rle(len(data), w["packets"]))
        r = sa.cost_slab(slab)
        assert r["total"] == w["total"], w["name"]
        assert r["npackets"] == w["npackets"], w["name"]
        assert sha(r["cum"]) == w["cum_sha256"], w["name"]
        fs = sa.final_state(slab)
        assert fs["ctx_state"] == w["ctx_state"] and [int(x) for x in fs["dists"]] == w["dists"], w["name"]
        assert sha(fs["probs"]) == w["probs_sha256"], w["name"]


def test_top_k_against_reference_fixtures_and_oracle(golden, golden_input):
    """mgl_top_k: same cost multiset and same best cost as the reference's finder (ties are
    heap-internal there); exactly the oracle's canonical list."""
    for t in golden["topk"]:
        data = golden_input(t["input"])
        n = len(data)
        sa = sa_for(t["input"], data, neighbours_per_step=8)
        slab = literal_slab(n) if t["slab"] == "literal" else slab_from_rle(n, golden["evolved_walks"][t["input"]])
        pk, costs = sa.top_k(P(slab), t["position"])
        assert sorted(int(c) for c in costs) == sorted(t["costs"]), (t["input"], t["position"])
        assert list(costs) == sorted(costs, reverse=True)  # pop order: worst first
        o = Oracle(data, dict_limit=0x400000)
        opk, ocosts = o.top_k(slab, t["position"], mode=1, k=20)
        assert as_list(pk) == as_list(opk) and [int(c) for c in costs] == [int(c) for c in ocosts]


def test_top_k_off_walk_is_an_error(golden_input):
    data = golden_input("hello")
    sa = sa_for("hello", data, neighbours_per_step=8)
    slab = P(slab_from_rle(11, [["L", 6], [2, 5, 5]]))
    with pytest.raises(binding.MglError):
        sa.top_k(slab, 8)


def test_substrings_against_reference_fixtures(golden, golden_input):
    for s in golden["substrings"]:
        data = golden_input(s["input"])
        sa = sa_for(s["input"], data, neighbours_per_step=8)
        for q in s["queries"]:
            offs, lens = sa.substrings(q["pos"], s["max_len"])
            assert len(offs) == q["count"], (s["input"], q["pos"])
            if "offs" in q:
                assert [int(x) for x in offs] == q["offs"] and [int(x) for x in lens] == q["lens"]
            else:
                assert sha(np.stack([offs, lens])) == q["sha256"]


def _check_neighbours(sa, o, slab, seed, step, K):
    costs, nd, diffs = sa.neighbours(step)
    bad = []
    for j in range(K):
        ok, cost, od = o.neighbour(slab, seed, step, j, keep=False, K=K)
        want = cost if ok else binding.INVALID_COST
        if int(costs[j]) != want:
            bad.append((j, int(costs[j]), want))
            continue
        if ok:
            got = [(int(d["position"]), as_list([d["old"]])[0], as_list([d["new"]])[0]) for d in diffs[j][: nd[j]]]
            exp = [(int(d["position"]), as_list([d["old"]])[0], as_list([d["new"]])[0]) for d in od]
            if got != exp:
                bad.append((j, got, exp))
    assert not bad, bad[:5]
    return costs


ENGINES = pytest.mark.parametrize("fullwalk", [False, True], ids=["incremental", "fullwalk"])


@ENGINES
@pytest.mark.parametrize("name", ["lorem4k", "enwik3k", "reps", "zeros600", "rand2k"])
def test_neighbours_bit_exact_vs_oracle(name, fullwalk, golden, golden_input):
    """Every neighbour of a step: device cost and journal == oracle (same counter RNG, same
    canonical top-K, same mutate/repair rules), from an all-literal and from an evolved base."""
    data = golden_input(name)
    n = len(data)
    K, seed = 128, 99
    sa = sa_for(name + "_nb", data, neighbours_per_step=K, seed=seed, fullwalk=fullwalk)
    o = Oracle(data, dict_limit=0x400000)
    bases = [literal_slab(n)]
    if name in golden["evolved_walks"]:
        bases.append(slab_from_rle(n, golden["evolved_walks"][name]))
    for w in golden["walks"]:
        if w["input"] == name and w["name"].endswith(("_reps", "_mixed")):
            bases.append(slab_from_rle(n, w["packets"]))
    for bi, base in enumerate(bases):
        sa.set_slab(P(base))
        for step in (0, 7 + bi):
            _check_neighbours(sa, o, base, seed, step, K)


@pytest.mark.parametrize("fullwalk,accept", [(False, "single"), (True, "single"), (False, "bulk"), (False, "auto")],
                         ids=["incremental-single", "fullwalk-single", "bulk", "auto"])
def test_sa_run_trajectory_vs_oracle(fullwalk, accept):
    """mgl_sa_run against orc_sa_batched: same accept decisions, same current/best cost every step, same
    final slabs; the stream decodes.  Single steps take the best acceptable neighbour, bulk steps every
    acceptable neighbour that is the best of its window; "auto" is a run of both, replayed in the oracle
    with the modes the library chose."""
    data = corpus.lorem(1800)
    n = len(data)
    K, seed, steps = 48, 1673551, 60
    ipe = steps * K  # evaluations per epoch: neighbour j of in-epoch step s is the reference's iteration s K + j
    sa = binding.SA(data, accept=accept, bulk_threshold=6, neighbours_per_step=K, seed=seed, iters_per_epoch=ipe, fullwalk=fullwalk)
    o = Oracle(data, dict_limit=0x400000)
    slab, best = literal_slab(n), literal_slab(n)
    evals = dropped = 0
    if accept == "auto":
        st = sa.run(steps)
        modes = sa.step_modes()
        assert len(modes) == steps and 0 < modes.sum() < steps, modes  # a mix of both kinds of step
        assert st["bulk_steps"] == modes.sum()
        ref = o.sa_batched(slab, best, 0, 0, seed, K, 0, ipe, 0, steps, modes=modes)
        evals, dropped = st["evaluations"], st["dropped_neighbours"]
        assert st["current_cost"] == ref["cur"]
        assert st["accepted"] == int(ref["trace"][:, 1].sum())
    else:
        modes = np.full(steps, 1 if accept == "bulk" else 0, dtype=np.uint8)
        ref = o.sa_batched(slab, best, 0, 0, seed, K, 0, ipe, 0, steps, modes=modes)
        for s in range(steps):
            st = sa.run(1)
            evals += st["evaluations"]
            dropped += st["dropped_neighbours"]
            assert st["current_cost"] == int(ref["trace"][s, 3]), s
            assert st["accepted"] == int(ref["trace"][s, 1]), s
        if accept == "bulk":
            assert int(ref["trace"][:, 1].max()) > 1  # steps that took several neighbours at once
    assert evals == ref["valid"] and dropped == ref["dropped"]
    cur, cur_cost = sa.current()
    bst, best_cost = sa.best()
    assert cur_cost == ref["cur"] and best_cost == ref["best"]
    assert as_list(cur) == as_list(slab) and as_list(bst) == as_list(best)
    assert sa.cost_slab(cur)["total"] == cur_cost  # the device's independent full walk
    stream = binding.emit_stream(data, bst)
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    # a second epoch from the best slab (main.c:75-77, phase 1): the global step keeps
    # counting, the in-epoch iteration restarts, the current cost is forgotten
    sa.begin_epoch(1, from_best=True)
    slab2 = best.copy()
    st = sa.run(25)
    ref2 = o.sa_batched(slab2, best, 0, ref["best"], seed, K, 1, ipe, steps, steps + 25, modes=sa.step_modes())
    assert st["steps"] == 25 and st["current_cost"] == ref2["cur"] and st["best_cost"] == ref2["best"]
    cur, _ = sa.current()
    assert as_list(cur) == as_list(slab2)
    bst, _ = sa.best()
    assert as_list(bst) == as_list(best)
    sa.close()


@pytest.mark.parametrize("name,K,steps", [("enwik30k", 2048, 120), ("lorem4k", 512, 200), ("reps", 256, 200), ("zeros600", 64, 100), ("elf64k", 2048, 120)])
def test_bulk_steps_keep_the_parse_valid(name, K, steps):
    """Hundreds of neighbours taken at once, step after step: every few steps the slab must still be a parse of the
    input whose every packet reproduces it (k_validate through mgl_sa_set_slab on a second handle), cost what an
    independent CPU walk says, and decode.  (Soft window ends let a neighbour start inside another one's tail: this
    is where an invalid combination would show.)"""
    data = {"enwik30k": lambda: corpus.enwik_like(30000, 0x81), "lorem4k": lambda: corpus.lorem(4096),
            "reps": lambda: (b"abcabcabd" * 300 + corpus.lorem(700)) * 2, "zeros600": lambda: b"\0" * 600 + corpus.lorem(200) + b"\0" * 300,
            "elf64k": lambda: corpus.config_input("c5", 65536)[0]}[name]()
    n = len(data)
    sa = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=77, iters_per_epoch=n)
    chk = binding.SA(data, accept="single", neighbours_per_step=8)
    o = Oracle(data, dict_limit=0x400000)
    taken = 0
    for s in range(0, steps, 20):
        st = sa.run(20)
        taken += st["accepted"]
        cur, cost = sa.current()
        slab = np.ascontiguousarray(cur).astype(literal_slab(1).dtype)
        assert cost == o.cost_slab(slab)["total"], (name, s)
        chk.set_slab(cur)  # raises unless every packet on the walk reproduces the input
        assert chk.current()[1] == cost
        assert lzma.decompress(binding.emit_stream(data, cur), format=lzma.FORMAT_ALONE) == data
    assert taken > steps or name == "zeros600"  # several moves per step on average
    bst, bcost = sa.best()
    assert bcost <= cost and lzma.decompress(binding.emit_stream(data, bst), format=lzma.FORMAT_ALONE) == data
    sa.close()
    chk.close()


def test_windows_and_drop_counter_vs_oracle():
    """Every neighbour's window [target, end) -- what a bulk step's selection works on -- equals the
    oracle's, on both engines, on a repetitive input whose repairs run long; and the count of
    neighbours dropped by the 64-entry journal equals the oracle's (the reference's undo stack is
    unbounded, packet_slab_undo_stack.c:70-77: such neighbours are lost here, and counted)."""
    data = corpus.lorem(3000)
    n, K, seed = len(data), 128, 5
    o = Oracle(data, dict_limit=0x400000)
    slab, best = literal_slab(n), literal_slab(n)
    o.sa_batched(slab, best, 0, 0, seed, K, 0, n, 0, 10, modes=np.ones(10, dtype=np.uint8))
    for fullwalk in (False, True):
        sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, fullwalk=fullwalk)
        sa.set_slab(P(slab))
        costs, nd, _ = sa.neighbours(777)
        win = sa.debug_dump(21, np.uint32).reshape(-1, 2)
        win2 = sa.debug_dump(22, np.uint32)
        ndrop = 0
        for j in range(K):
            st, cost, diffs, w = o.neighbour_ex(slab, seed, 777, j, K=K)
            ndrop += st == -1
            assert (int(costs[j]) == cost) and (st == 1) == (int(costs[j]) != binding.INVALID_COST), j
            if st == 1:
                assert (int(win[j, 0]), int(win[j, 1]), int(win2[j]) & 0x7FFFFFFF, int(win2[j]) >> 31) == w, (j, win[j], win2[j], w)
        sa.close()
    # the drop counter through mgl_sa_run (one single step from the same slab, both engines agree with the oracle)
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, iters_per_epoch=n)
    sa.set_slab(P(slab))
    st = sa.run(1)
    s2, b2 = slab.copy(), best.copy()
    ref = o.sa_batched(s2, b2, 0, 0, seed, K, 0, n, 0, 1)
    assert st["dropped_neighbours"] == ref["dropped"] and st["evaluations"] == ref["valid"]
    sa.close()


@ENGINES
@pytest.mark.parametrize("data", [b"x", b"ab", b"aaaa", rand_bytes(70, 5)], ids=["n1", "n2", "aaaa", "rand70"])
def test_tiny_and_incompressible_inputs(data, fullwalk):
    """Edge cases: inputs where few or no neighbours exist (the reference spins forever at
    main.c:81-84 there); failed generates come back as UINT64_MAX and nothing crashes."""
    K, seed = 16, 5
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, fullwalk=fullwalk)
    o = Oracle(data, dict_limit=0x400000)
    base = literal_slab(len(data))
    assert sa.cost_slab(P(base))["total"] == o.cost_slab(base)["total"]
    _check_neighbours(sa, o, base, seed, 3, K)
    st = sa.run(3)
    assert st["steps"] == 3 and st["evaluations"] + st["failed"] == 3 * K
    cur, cost = sa.current()
    assert cost == o.cost_slab(np.ascontiguousarray(cur).astype(literal_slab(1).dtype))["total"]
    assert lzma.decompress(binding.emit_stream(data, cur), format=lzma.FORMAT_ALONE) == data
    sa.close()


def test_full_size_c2_properties():
    """BASELINE configs[1] at full size (100 000 B, 4 096 neighbours/step): size-independent
    properties -- device cost == an independent CPU walk of the device's slab, winner cost <=
    previous cost, sampled neighbours == oracle, round trip through liblzma."""
    data, _ = corpus.config_input("c2")
    n = len(data)
    K, seed = 4096, 1673551
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed)
    o = Oracle(data, dict_limit=0x400000)
    base = literal_slab(n)
    costs, nd, diffs = sa.neighbours(0)
    for j in (0, 1, 17, 511, 2048, 4095):
        ok, cost, od = o.neighbour(base, seed, 0, j, keep=False, K=K)
        assert int(costs[j]) == (cost if ok else binding.INVALID_COST), j
    prev = None
    for _ in range(4):
        st = sa.run(1)
        cur, cost = sa.current()
        assert cost == st["current_cost"] == o.cost_slab(cur.astype(base.dtype))["total"]
        if prev is not None and not st["accepted"]:
            assert cost == prev
        prev = cost
    bst, best_cost = sa.best()
    stream = binding.emit_stream(data, bst)
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    # the table truncates each event's cost (floor), so the estimate runs a little low:
    # within 0.1 % of the real stream
    assert abs((18 + best_cost / 16384) - len(stream)) <= 8 + len(stream) / 1000
    sa.close()


def test_pb2_elf_shaped_properties():
    """BASELINE configs[4] shape (lc=0 lp=0 pb=2) on a 64 KiB slice: the pb extension has no
    reference implementation (parity unpinned); pinned by oracle equality + liblzma decode."""
    data, _ = corpus.config_input("c5", 65536)
    data = data[16384:32768]
    K, seed = 256, 11
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, pb=2, max_bucket_scan=0)
    o = Oracle(data, pb=2, dict_limit=0x400000)
    base = literal_slab(len(data))
    assert sa.cost_slab(P(base))["total"] == o.cost_slab(base)["total"]
    costs, nd, diffs = sa.neighbours(2)
    for j in range(0, K, 8):
        ok, cost, od = o.neighbour(base, seed, 2, j, keep=False, K=K)
        assert int(costs[j]) == (cost if ok else binding.INVALID_COST), j
    sa.run(3)
    cur, cost = sa.current()
    assert cost == o.cost_slab(cur.astype(base.dtype))["total"]
    stream = binding.emit_stream(data, cur, pb=2)
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    sa.close()


def test_c5_full_size_with_bucket_cap():
    """BASELINE configs[4] at full size: 1 MiB ELF-shaped, pb = 2, bucket scan capped at the 4 096 nearest hits
    (the reference scans every hit: > 10^6 candidates inside a zero run; the cap is the same rule in
    the oracle, for_each_substring).  Top-K inside the longest zero run and sampled neighbours of an evolved slab
    equal the oracle's; pb != 0 has no reference implementation (parity unpinned), the stream decodes."""
    data, _ = corpus.config_input("c5")
    n, K, seed, cap = len(data), 4096, 1673551, 4096
    sa = binding.SA(data, neighbours_per_step=K, seed=seed, pb=2, max_bucket_scan=cap, iters_per_epoch=n)
    o = Oracle(data, pb=2, dict_limit=0x400000, max_bucket_scan=cap)
    base = literal_slab(n)
    z = np.frombuffer(data, dtype=np.uint8) == 0
    edges = np.flatnonzero(np.diff(np.concatenate([[0], z.view(np.int8), [0]])))
    starts, ends = edges[::2], edges[1::2]
    longest = int(np.argmax(ends - starts))
    assert ends[longest] - starts[longest] > 2000  # the padded tables of an ELF
    for p in (int(starts[longest]) + 1500, int((starts[longest] + ends[longest]) // 2), int(ends[longest]) - 3):
        got_pk, got_cost = sa.top_k(P(base), p)
        want_pk, want_cost = o.top_k(base, p, mode=1)
        assert [int(x) for x in got_cost] == [int(x) for x in want_cost], p
        assert as_list(got_pk) == as_list(want_pk), p
    st = sa.run(40)  # the default accept mode: bulk steps from the start
    assert st["bulk_steps"] > 0 and st["full_rebuilds"] == 0
    cur, cost = sa.current()
    slab = np.ascontiguousarray(cur).astype(base.dtype)
    assert cost == o.cost_slab(slab)["total"]
    costs, nd, diffs = sa.neighbours(1000)
    for j in range(0, K, 331):
        ok, c, od = o.neighbour(slab, seed, 1000, j, keep=False, K=K)
        assert int(costs[j]) == (c if ok else binding.INVALID_COST), j
    best, _ = sa.best()
    assert lzma.decompress(binding.emit_stream(data, best, pb=2), format=lzma.FORMAT_ALONE) == data
    sa.close()


def test_c4_shape_100_mb():
    """BASELINE configs[3], one GPU's share: 100 000 000 B enwik-shaped, 16 384 neighbours/step.  Size-independent
    properties after a few steps of the default accept mode (bulk and single): device cost == an independent CPU walk
    of the device's slab, sampled neighbours of that slab == oracle (cost and journal), no fallback to the serial
    builder, no neighbour lost to the full-walk last resort."""
    data, _ = corpus.config_input("c4")
    n, K, seed = len(data), 16384, 1673551
    sa = binding.SA(data, neighbours_per_step=K, seed=seed, iters_per_epoch=n)
    o = Oracle(data, dict_limit=0x400000)
    st = sa.run(6)
    assert st["steps"] == 6 and st["full_rebuilds"] == 0 and st["fallback_neighbours"] == 0
    sa.set_accept_mode("single")
    st2 = sa.run(3)  # the incremental accept path at this size
    assert st2["accepted"] == 3 and st2["full_rebuilds"] == 0 and st2["fallback_neighbours"] == 0
    cur, cost = sa.current()
    slab = np.ascontiguousarray(cur).astype(literal_slab(1).dtype)
    assert cost == o.cost_slab(slab)["total"]
    costs, nd, diffs = sa.neighbours(77)
    for j in (5, 9000, 123, 4567, 12000, 16383):
        ok, c, od = o.neighbour(slab, seed, 77, j, keep=False, K=K)
        assert int(costs[j]) == (c if ok else binding.INVALID_COST), j
        if ok:
            got = [(int(d["position"]), as_list([d["old"]])[0], as_list([d["new"]])[0]) for d in diffs[j][: nd[j]]]
            assert got == [(int(d["position"]), as_list([d["old"]])[0], as_list([d["new"]])[0]) for d in od], j
    sa.close()
    # the batch accept at this size (chains of 10^7 entries rewritten in place, hundreds of chunks per stretch): a step of few
    # neighbours on the greedy-seeded slab takes few moves, so every bulk step patches them in; the patched structures give
    # the cost an independent CPU walk gives, and the cost a rebuild from the slab gives
    sb = binding.SA(data, neighbours_per_step=192, seed=seed, iters_per_epoch=n, accept="bulk")
    sb.seed_greedy(64)
    moves = 0
    for s in range(6):
        moves += sb.run(1)["accepted"]
    assert sb.batch_counters()[0] >= 4 and sb.batch_counters()[1] == 0 and moves > 20
    cur, cost = sb.current()
    slab = np.ascontiguousarray(cur).astype(literal_slab(1).dtype)
    assert cost == o.cost_slab(slab)["total"]
    sb.set_slab(cur)
    assert sb.current()[1] == cost
    sb.close()


@pytest.mark.parametrize("cfg,size", [("c2", 100000), ("c5", 200000), ("c1", 4096), ("c2", 4097), ("c2", 2), ("c2", 3)])
def test_match_index_built_on_device(cfg, size):
    """substring_enumerator.c:26-47: positions bucketed by leading bigram, ascending inside a
    bucket.  The device builds it with a stable two-pass counting sort (mgl_index.hip)."""
    data, _ = corpus.config_input(cfg, size)
    sa = binding.SA(data, accept="single", neighbours_per_step=8)
    d = np.frombuffer(data, dtype=np.uint8).astype(np.uint32)
    keys = (d[:-1] << 8) | d[1:]
    want_pos = np.argsort(keys, kind="stable").astype(np.uint32)
    want_off = np.searchsorted(keys[want_pos], np.arange(65537), side="left").astype(np.uint32)
    got_off = sa.debug_dump(12, np.uint32)
    got_pos = sa.debug_dump(13, np.uint32)
    assert (got_off == want_off).all()
    assert (got_pos == want_pos).all()
    # the second order of the same positions: by the four bytes at the position (zero padded), then by position
    dp = np.concatenate([d, np.zeros(4, dtype=np.uint32)])
    m = len(d) - 1
    idx = np.arange(m)
    key4 = (dp[idx] << 24) | (dp[idx + 1] << 16) | (dp[idx + 2] << 8) | dp[idx + 3]
    want_quad = np.argsort(key4, kind="stable").astype(np.uint32)
    got_quad = sa.debug_dump(18, np.uint32)
    got_qnx = sa.debug_dump(19, np.uint16)
    assert (got_quad == want_quad).all()
    assert (got_qnx == (key4[want_quad] & 0xFFFF).astype(np.uint16)).all()
    # the deeper orders top-K scans by length (mgl_index.hip): positions by their first D bytes, then by position;
    # rank = inverse permutation; run start = first entry with the same D bytes; the byte(s) behind the prefix
    dp = np.concatenate([d, np.zeros(32, dtype=np.uint32)]).astype(np.uint8)
    for D, sel in ((3, (31, 41, 51, 61)), (5, (33, 43, 53, 63)), (7, (35, 45, 55, 65)), (8, (70, 71, 72, 73)), (16, (74, 75, 76, 77))):
        cols = np.stack([dp[idx + k] for k in range(D)])           # D x m
        want = np.lexsort(cols[::-1]).astype(np.uint32)            # stable: ties keep position order
        got = sa.debug_dump(sel[0], np.uint32)
        assert (got == want).all(), D
        rank = sa.debug_dump(sel[1], np.uint32)
        assert (rank[want] == np.arange(m, dtype=np.uint32)).all(), D
        keys = cols[:, want].T                                     # m x D, sorted
        head = np.ones(m, dtype=bool)
        head[1:] = (keys[1:] != keys[:-1]).any(axis=1)
        want_run = np.maximum.accumulate(np.where(head, np.arange(m), 0)).astype(np.uint32)
        assert (sa.debug_dump(sel[2], np.uint32) == want_run).all(), D
        if D < 8:
            assert (sa.debug_dump(sel[3], np.uint8) == dp[want + D]).all(), D
        else:
            nx = np.zeros(m, dtype=np.uint64)
            for k in range(8):
                nx |= dp[want + D + k].astype(np.uint64) << np.uint64(8 * k)
            assert (sa.debug_dump(sel[3], np.uint64) == nx).all(), D
    sa.close()


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 2), (0, 2, 0), (1, 1, 1), (4, 0, 0)])
def test_literal_context_and_position_bits(lc, lp, pb):
    """lc / lp / pb beyond the reference's hard-coded 0/0/0 (main.c:45; xz's default is
    3/0/2).  No reference implementation exists for these (parity unpinned): the device must equal
    the CPU restatement, which follows the standard LZMA context definitions, and the emitted
    stream must decode with liblzma."""
    data, _ = corpus.config_input("c2", 12000)
    K, seed = 192, 5
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, lc=lc, lp=lp, pb=pb)
    o = Oracle(data, lc=lc, lp=lp, pb=pb, dict_limit=0x400000)
    base = literal_slab(len(data))
    assert sa.cost_slab(P(base))["total"] == o.cost_slab(base)["total"]
    costs, nd, diffs = sa.neighbours(1)
    for j in range(0, K, 6):
        ok, cost, od = o.neighbour(base, seed, 1, j, keep=False, K=K)
        assert int(costs[j]) == (cost if ok else binding.INVALID_COST), j
    sa.run(25)
    cur, cost = sa.current()
    assert cost == o.cost_slab(cur.astype(base.dtype))["total"]
    costs, nd, diffs = sa.neighbours(77)
    curo = cur.astype(base.dtype)
    for j in range(0, K, 12):
        ok, c2, od = o.neighbour(curo, seed, 77, j, keep=False, K=K)
        assert int(costs[j]) == (c2 if ok else binding.INVALID_COST), j
    stream = binding.emit_stream(data, cur, lc=lc, lp=lp, pb=pb)
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    assert abs((18 + cost / 16384) - len(stream)) <= 8 + len(stream) / 1000
    sa.close()


def test_full_size_c3_properties():
    """BASELINE configs[2] at full size (10 192 446 B, 16 384 neighbours/step): device cost == an
    independent CPU walk of the device's slab after every step, sampled neighbour costs == oracle
    (from the all-literal slab and from an evolved one), best slab == the parallel builder's and
    the full-walk hook's cost, round trip through liblzma."""
    data, _ = corpus.config_input("c3")
    n = len(data)
    K, seed = 16384, 1673551
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed)
    o = Oracle(data, dict_limit=0x400000)
    base = literal_slab(n)
    assert sa.current()[1] == o.cost_slab(base)["total"]
    costs, nd, diffs = sa.neighbours(0, want_diffs=False)
    for j in (0, 5, 4097, 16383):
        ok, cost, od = o.neighbour(base, seed, 0, j, keep=False, K=K)
        assert int(costs[j]) == (cost if ok else binding.INVALID_COST), j
    for s in range(6):
        st = sa.run(1)
        assert st["full_rebuilds"] == 0 and st["fallback_neighbours"] == 0
    cur, cost = sa.current()
    curo = cur.astype(base.dtype)
    assert cost == st["current_cost"] == o.cost_slab(curo)["total"]
    assert sa.cost_slab(P(curo))["total"] == cost          # the one-wavefront full walk agrees
    costs, nd, diffs = sa.neighbours(77, want_diffs=False)
    for j in (3, 9000):
        ok, c2, od = o.neighbour(curo, seed, 77, j, keep=False, K=K)
        assert int(costs[j]) == (c2 if ok else binding.INVALID_COST), j
    sa.set_slab(cur)                                         # block-parallel rebuild of the evolved slab
    assert sa.current()[1] == cost
    bst, best_cost = sa.best()
    stream = binding.emit_stream(data, bst)
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    assert abs((18 + best_cost / 16384) - len(stream)) <= 8 + len(stream) / 1000
    sa.close()


def test_evolved_c2_neighbours_vs_oracle():
    """Sampled neighbour costs against the oracle on an SA-evolved full-size c2 slab (many matches
    and reps on the walk: the repair path and the second pass are exercised), both engines."""
    data, _ = corpus.config_input("c2")
    K, seed = 4096, 1673551
    sa = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed)
    o = Oracle(data, dict_limit=0x400000)
    sa.run(400)
    cur, cost = sa.current()
    curo = cur.astype(literal_slab(1).dtype)
    assert cost == o.cost_slab(curo)["total"]
    full = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, fullwalk=True)
    full.set_slab(cur)
    ca, _, _ = sa.neighbours(400, want_diffs=False)
    cf, _, _ = full.neighbours(400, want_diffs=False)
    assert (ca == cf).all()
    for j in range(0, K, 257):
        ok, c2, od = o.neighbour(curo, seed, 400, j, keep=False, K=K)
        assert int(ca[j]) == (c2 if ok else binding.INVALID_COST), j
    sa.close()
    full.close()


def test_bulk_rollback_net_restores_the_step(golden_input):
    """The bulk step's safety net (k_validate after the rebuild, k_bulk_rollback + a second rebuild when the combined
    parse fails) has never fired on its own; mgl_debug_set key 3 makes the next bulk steps that took moves count as
    failed.  A step taken back leaves slab, costs and best slab exactly as they were (only the step, iteration and
    evaluation counters move on), and the chain carries on consistently afterwards."""
    data = corpus.enwik_like(30000, 0x524F)
    sa = binding.SA(data, accept="bulk", neighbours_per_step=512, seed=41, iters_per_epoch=10**7)
    o = Oracle(data, dict_limit=0x400000)
    sa.run(3)
    before, cost_before = sa.current()
    best_before, bcost_before = sa.best()
    assert sa.L.mgl_debug_set(sa.h, 3, 2) == 0
    st = sa.run(2)
    assert st["bulk_rollbacks"] == 2 and st["accepted"] == 0 and st["bulk_steps"] == 2 and st["evaluations"] > 0
    after, cost_after = sa.current()
    best_after, bcost_after = sa.best()
    assert cost_after == cost_before and (after == before).all()
    assert bcost_after == bcost_before and (best_after == best_before).all()
    assert cost_after == o.cost_slab(after.astype(literal_slab(1).dtype))["total"]
    st = sa.run(4)
    assert st["bulk_rollbacks"] == 0 and st["accepted"] > 0
    cur, cost = sa.current()
    assert cost < cost_before and cost == o.cost_slab(cur.astype(literal_slab(1).dtype))["total"]
    bst, _ = sa.best()
    assert lzma.decompress(binding.emit_stream(data, bst), format=lzma.FORMAT_ALONE) == data
    # epochs from the best slab still work after a rollback
    sa.begin_epoch(1, from_best=True)
    st = sa.run(2)
    cur, cost = sa.current()
    assert cost == o.cost_slab(cur.astype(literal_slab(1).dtype))["total"]
    sa.close()


def test_bulk_steps_do_not_depend_on_timing(monkeypatch):
    """Two chains with the same seed -- one with its step in two slices, one in three (different launch order and
    timing) -- and the oracle: bulk steps over an input full of SHORT_REP packets (doubled letters), where a repair far
    behind the mutated packet turns SHORT_REPs back into literals.  The taken journals are written in parallel, so
    they must touch disjoint entries: with soft window ends that did not reach behind every changed packet two of
    them could overlap and the result depended on which write came last (seen at 10 MB, where two identical chains
    parted after 19 steps)."""
    import random
    r = random.Random(5)
    out = bytearray()
    words = [bytes(r.choice(b"abcdefgh") for _ in range(r.randint(2, 6))) for _ in range(40)]
    while len(out) < 1_200_000:
        for ch in r.choice(words):
            out.append(ch)
            if r.random() < 0.45:
                out.append(ch)
        if r.random() < 0.3:
            out += b"  "
    data = bytes(out[:1_200_000])
    K = 4096
    monkeypatch.setenv("MGL_NO_ADAPT", "1")
    monkeypatch.setenv("MGL_HALVES", "2")
    a = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=77)
    monkeypatch.setenv("MGL_HALVES", "3")
    b = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=77)
    monkeypatch.delenv("MGL_HALVES")
    taken = 0
    for s in range(48):
        sa_, sb_ = a.run(1), b.run(1)
        assert sa_["current_cost"] == sb_["current_cost"] and sa_["accepted"] == sb_["accepted"], s
        assert sa_["bulk_rollbacks"] == 0 and sa_["bulk_double_writes"] == 0 and sb_["bulk_double_writes"] == 0
        taken += sa_["accepted"]
    ca, cost = a.current()
    cb, _ = b.current()
    assert (ca == cb).all() and taken > 48 * 100
    assert a.cost_slab(ca, want_cum=False)["total"] == cost
    assert lzma.decompress(binding.emit_stream(data, ca), format=lzma.FORMAT_ALONE) == data
    a.close()
    b.close()


@pytest.mark.parametrize("seed", [1, 5])
def test_bulk_trajectory_vs_oracle_on_short_rep_heavy_input(seed):
    """Bulk steps, device against oracle step by step, on the input family where repairs turn SHORT_REP packets far behind
    the mutated packet back into literals (tests/test_oracle_golden.py:doubled_letters; these seeds made two taken
    journals write one slab entry before the soft window end reached behind every changed packet)."""
    from test_oracle_golden import doubled_letters
    data = doubled_letters(seed, 2600)
    n, K, steps = len(data), 96, 60
    sa = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=seed * 7717, iters_per_epoch=n)
    o = Oracle(data, dict_limit=0x400000)
    slab, best = literal_slab(n), literal_slab(n)
    before = o.bulk_overlaps()
    ref = o.sa_batched(slab, best, 0, 0, seed * 7717, K, 0, n, 0, steps, modes=np.ones(steps, dtype=np.uint8))
    assert o.bulk_overlaps() == before
    for s in range(steps):
        st = sa.run(1)
        assert st["current_cost"] == int(ref["trace"][s, 3]) and st["accepted"] == int(ref["trace"][s, 1]), s
        assert st["bulk_double_writes"] == 0 and st["bulk_rollbacks"] == 0  # k_bulk_round's compare-and-swap saw every entry once
    cur, cost = sa.current()
    assert cost == ref["cur"] and as_list(cur) == as_list(slab)
    assert lzma.decompress(binding.emit_stream(data, sa.best()[0]), format=lzma.FORMAT_ALONE) == data
    sa.close()


def test_auto_mode_does_not_depend_on_how_a_run_is_cut_into_calls():
    """MGL_ACCEPT_AUTO picks single or bulk per block of 16 / 4 steps from device counters; a block carries over from one
    mgl_sa_run call to the next, so run(1) x N, run(7) x ..., and run(N) walk the same trajectory (and begin_epoch starts the
    same way whatever the previous epoch left behind)."""
    data = corpus.enwik_like(40000, 0x77)
    K, total = 1024, 84

    def chain(chunks):
        # a threshold of 150 improving neighbours per step: the bulk phase of this input ends after a few dozen steps
        sa = binding.SA(data, accept="auto", bulk_threshold=150, neighbours_per_step=K, seed=99, iters_per_epoch=len(data))
        modes, costs = [], []
        for c in chunks:
            st = sa.run(c)
            modes += list(sa.step_modes())
            costs.append(st["current_cost"])
        sa.begin_epoch(1, from_best=True)
        st = sa.run(20)
        modes2 = list(sa.step_modes())
        cur, cost = sa.current()
        sa.close()
        return modes, costs[-1], modes2, cost, cur

    a = chain([total])
    b = chain([1] * total)
    c = chain([7] * 12)
    assert a[0] == b[0] == c[0] and 0 < sum(a[0]) < total     # both kinds of step occur
    assert a[1] == b[1] == c[1]
    assert a[2] == b[2] == c[2] and a[3] == b[3] == c[3]
    assert (a[4] == b[4]).all() and (a[4] == c[4]).all()


def test_full_size_c3_bulk_steps_vs_oracle(monkeypatch):
    """BASELINE configs[2] at full size in the library's default accept mode (auto: bulk steps from the all-literal slab),
    two chains with different launch orders side by side -- the size at which two taken journals once wrote one slab entry
    (identical chains parted after 19 bulk steps).  After the bulk steps: both chains identical step by step, no rollback, no
    double write, device cost == the oracle's walk of the device's slab, sampled neighbours of the post-bulk slab == oracle
    (cost and journal)."""
    data, _ = corpus.config_input("c3")
    n, K, seed, steps = len(data), 16384, 1673551, 24
    monkeypatch.setenv("MGL_NO_ADAPT", "1")
    monkeypatch.setenv("MGL_HALVES", "2")
    a = binding.SA(data, accept="auto", neighbours_per_step=K, seed=seed)
    monkeypatch.setenv("MGL_HALVES", "3")
    b = binding.SA(data, accept="auto", neighbours_per_step=K, seed=seed)
    monkeypatch.delenv("MGL_HALVES")
    bulk = moves = 0
    for s in range(steps):
        sa_, sb_ = a.run(1), b.run(1)
        assert sa_["current_cost"] == sb_["current_cost"] and sa_["accepted"] == sb_["accepted"] and sa_["evaluations"] == sb_["evaluations"], s
        for st in (sa_, sb_):
            assert st["bulk_rollbacks"] == 0 and st["bulk_double_writes"] == 0 and st["dropped_neighbours"] == 0, (s, st)
        bulk += sa_["bulk_steps"]
        moves += sa_["accepted"]
    assert bulk >= 16 and moves > 16 * 1000          # thousands of moves per bulk step from the all-literal slab
    ca, cost = a.current()
    cb, cost_b = b.current()
    assert cost == cost_b and (ca == cb).all()
    o = Oracle(data, dict_limit=0x400000)
    curo = ca.astype(literal_slab(1).dtype)
    assert o.cost_slab(curo)["total"] == cost
    costs, nd, diffs = a.neighbours(steps, want_diffs=True)
    for j in range(0, K, 1024):                      # 16 sampled neighbours of the post-bulk slab
        ok, c2, od = o.neighbour(curo, seed, steps, j, keep=False, K=K)
        assert int(costs[j]) == (c2 if ok else binding.INVALID_COST), j
        if ok:
            got = [(int(d["position"]), as_list([d["old"]])[0], as_list([d["new"]])[0]) for d in diffs[j][: nd[j]]]
            exp = [(int(d["position"]), as_list([d["old"]])[0], as_list([d["new"]])[0]) for d in od]
            assert got == exp, j
    assert lzma.decompress(binding.emit_stream(data, a.best()[0]), format=lzma.FORMAT_ALONE) == data
    a.close()
    b.close()
