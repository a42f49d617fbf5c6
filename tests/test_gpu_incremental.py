"""The incremental engine's base structures (bitmaps, special-state records, dense
checkpoints, per-context event chains) against the CPU oracle's event trace, and the two
device engines against each other.  `-m gpu`."""
import lzma

import numpy as np
import pytest

from _libs import Oracle, literal_slab, walk
from conftest import slab_from_rle
from megalania_amd import binding, corpus

pytestmark = pytest.mark.gpu

INF = 0xFFFFFFFF


def ref_to_product_ctx(r, lit):
    """reference struct order (lit|len|rep_len|dist|cs) -> product order (cs|len|rep_len|dist|lit)"""
    r = np.asarray(r, dtype=np.int64)
    out = np.empty_like(r)
    a = r < lit
    out[a] = 1847 + r[a]
    b = (r >= lit) & (r < lit + 514)
    out[b] = 432 + (r[b] - lit)
    c = (r >= lit + 514) & (r < lit + 1028)
    out[c] = 946 + (r[c] - lit - 514)
    d = (r >= lit + 1028) & (r < lit + 1415)
    out[d] = 1460 + (r[d] - lit - 1028)
    e = r >= lit + 1415
    out[e] = r[e] - lit - 1415
    return out


def check_base(sa, o, slab, data):
    n = len(data)
    tr = o.trace_events(slab)
    total = sa.nprobs
    ctx = ref_to_product_ctx(tr["ctx"], 0x300)
    off = sa.debug_dump(0, np.uint32)
    ln = sa.debug_dump(1, np.uint32)
    cap = sa.debug_dump(8, np.uint32)
    cpos = sa.debug_dump(2, np.uint32)
    cev = sa.debug_dump(3, np.uint16)
    ev = (tr["bit"].astype(np.uint16) << 15) | tr["prob"]
    fin = o.cost_slab(slab, want_probs=True)["probs"]
    fin_prod = np.empty(total, dtype=np.uint16)
    fin_prod[ref_to_product_ctx(np.arange(total), 0x300)] = fin
    assert int(ln.sum()) == len(ctx)
    order = np.argsort(ctx, kind="stable")
    sctx, spos, sev = ctx[order], tr["pos"][order], ev[order]
    starts = np.searchsorted(sctx, np.arange(total))
    for c in range(total):
        k, m = int(off[c]), int(ln[c])
        assert m + 1 <= cap[c]
        s = starts[c]
        assert (cpos[k:k + m] == spos[s:s + m]).all(), c
        assert (cev[k:k + m] == sev[s:s + m]).all(), c
        assert cpos[k + m] == INF and cev[k + m] == fin_prod[c], c
    # bitmaps
    w = walk(slab)
    on = np.zeros(n, dtype=bool)
    on[w] = True
    onw = sa.debug_dump(4, np.uint64)
    got_on = np.unpackbits(onw.view(np.uint8), bitorder="little")[:n].astype(bool)
    assert (got_on == on).all()
    special = on & (slab["type"] != 1)
    sp0 = sa.debug_dump(5, np.uint64)
    got_sp = np.unpackbits(sp0.view(np.uint8), bitorder="little")[:n].astype(bool)
    assert (got_sp == special).all()
    # state records before every special packet
    st = sa.debug_dump(6, np.uint32).reshape(n, 8)
    pk_state = dict(zip((int(p) for p in tr["pk_pos"]), tr["pk_state"]))
    for p in np.nonzero(special)[0]:
        assert (st[p, :5] == pk_state[int(p)]).all(), p
    # dense checkpoints: model before the first packet at or after every `spacing`-th byte
    ck = sa.debug_dump(7, np.uint16).reshape(-1, (total + 7) // 8 * 8)
    spacing = [s for s in (8, 16, 32, 64) if (n + s - 1) // s == ck.shape[0]]
    assert len(spacing) >= 1, (n, ck.shape)
    spacing = spacing[0]
    probs = np.full(total, 1024, dtype=np.int64)
    ei, nck = 0, 0
    for p in w:
        while nck * spacing <= p:
            assert (ck[nck, :total] == probs).all(), nck
            nck += 1
        while ei < len(ctx) and tr["pos"][ei] == p:
            c, b = ctx[ei], int(tr["bit"][ei])
            v = probs[c]
            probs[c] = v - (v >> 5) if b else v + ((2048 - v) >> 5)
            ei += 1


@pytest.mark.parametrize("serial_build", [False, True], ids=["parallel_build", "serial_build"])
@pytest.mark.parametrize("name", ["lorem4k", "enwik3k", "reps", "zeros600"])
def test_base_structures_match_oracle_trace(name, serial_build, golden, golden_input):
    data = golden_input(name)
    n = len(data)
    sa = binding.SA(data, accept="single", neighbours_per_step=8, serial_build=serial_build)
    o = Oracle(data)
    slabs = [literal_slab(n)]
    if name in golden["evolved_walks"]:
        slabs.append(slab_from_rle(n, golden["evolved_walks"][name]))
    for wk in golden["walks"]:
        if wk["input"] == name and wk["name"].endswith("_reps"):
            slabs.append(slab_from_rle(n, wk["packets"]))
    for slab in slabs:
        sa.set_slab(slab.astype(binding.PACKET))
        check_base(sa, o, slab, data)
    sa.close()


def test_engines_agree_on_c2_neighbours():
    """Full-size BASELINE configs[1]: every neighbour of a step costed by the incremental engine
    equals the full-walk engine (which equals the oracle, test_gpu_parity.py)."""
    data, _ = corpus.config_input("c2")
    K, seed = 1024, 4242
    inc = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed)
    full = binding.SA(data, accept="single", neighbours_per_step=K, seed=seed, fullwalk=True)
    for rounds in range(3):
        for step in (rounds * 10, rounds * 10 + 1):
            ci, ni, di = inc.neighbours(step)
            cf, nf, df = full.neighbours(step)
            assert (ci == cf).all(), np.nonzero(ci != cf)[0][:8]
            assert (ni == nf).all()
            for j in range(K):
                assert (di[j][: ni[j]] == df[j][: nf[j]]).all(), j
        si, sf = inc.run(6), full.run(6)
        assert si["current_cost"] == sf["current_cost"] and si["evaluations"] == sf["evaluations"]
    a, ca = inc.current()
    b, cb = full.current()
    assert ca == cb and (a == b).all()
    inc.close()
    full.close()


def canonical_base(sa, slab):
    """The incremental engine's base structures in an offset-independent form."""
    n, total = sa.n, sa.nprobs
    off = sa.debug_dump(0, np.uint32)
    ln = sa.debug_dump(1, np.uint32)
    cap = sa.debug_dump(8, np.uint32)
    cpos = sa.debug_dump(2, np.uint32)
    cev = sa.debug_dump(3, np.uint16)
    chains = []
    for c in range(total):
        k, m = int(off[c]), int(ln[c])
        assert m + 1 <= cap[c], c
        chains.append((cpos[k:k + m + 1].copy(), cev[k:k + m + 1].copy()))
    on = np.unpackbits(sa.debug_dump(4, np.uint64).view(np.uint8), bitorder="little")[:n].astype(bool)
    sp = np.unpackbits(sa.debug_dump(5, np.uint64).view(np.uint8), bitorder="little")[:n].astype(bool)
    st = sa.debug_dump(6, np.uint32).reshape(n, 8)[:, :5]
    ck = sa.debug_dump(7, np.uint16).reshape(-1, (total + 7) // 8 * 8)[:, :total]
    # the chain index is checked against the chains it indexes, here, whoever built or patched it: entry [c][b] = entries of
    # context c's chain with a position below b << shift (the last column: the chain's length)
    shift = 8 if n <= (1 << 20) else 9 if n <= (1 << 23) else 10
    nsb = (n + (1 << shift) - 1) >> shift
    stride = (nsb + 2 + 3) & ~3
    idx = sa.debug_dump(83, np.uint32).reshape(-1, stride)
    bounds = (np.arange(nsb + 1, dtype=np.int64) << shift)
    for c in range(total):
        pos = chains[c][0][:-1].astype(np.int64)
        want = np.searchsorted(pos, bounds, side="left")
        want[nsb] = len(pos)
        assert (idx[c, : nsb + 1] == want).all(), ("chain index", c, np.nonzero(idx[c, : nsb + 1] != want)[0][:5])
    return dict(chains=chains, on=on, sp=sp, st=st[sp], ck=ck)


def assert_same_base(a, b, what):
    assert (a["on"] == b["on"]).all(), what
    assert (a["sp"] == b["sp"]).all(), what
    assert (a["st"] == b["st"]).all(), (what, np.nonzero((a["st"] != b["st"]).any(axis=1))[0][:5])
    for c, (x, y) in enumerate(zip(a["chains"], b["chains"])):
        assert len(x[0]) == len(y[0]) and (x[0] == y[0]).all() and (x[1] == y[1]).all(), (what, "chain", c)
    bad = np.nonzero((a["ck"] != b["ck"]).any(axis=1))[0]
    assert len(bad) == 0, (what, "checkpoints", bad[:5], np.nonzero(a["ck"][bad[0]] != b["ck"][bad[0]])[0][:5])


@pytest.mark.parametrize("name,K,steps", [("lorem4k", 64, 120), ("enwik3k", 96, 150), ("reps", 48, 120), ("zeros600", 32, 60)])
def test_incremental_accept_equals_rebuild(name, K, steps, golden, golden_input):
    """After every accepted step the incrementally maintained base (bitmaps, special-state
    records, chains, dense checkpoints) is identical to one rebuilt from the slab."""
    data = golden_input(name)
    # a long epoch: sqrt(N) is large against i*i, so uphill moves (main.c:86) keep coming as well
    inc = binding.SA(data, accept="single", neighbours_per_step=K, seed=5, iters_per_epoch=10**7)
    ref = binding.SA(data, accept="single", neighbours_per_step=8, seed=5)
    o = Oracle(data, dict_limit=0x400000)
    accepted = 0
    for s in range(steps):
        st = inc.run(1)
        accepted += st["accepted"]
        if st["accepted"] and (s < 30 or s % 7 == 0):
            cur, cost = inc.current()
            assert cost == o.cost_slab(cur.astype(literal_slab(1).dtype))["total"], s
            ref.set_slab(cur)
            assert_same_base(canonical_base(inc, cur), canonical_base(ref, cur), (name, s))
    assert accepted >= 5
    inc.close()
    ref.close()


@pytest.mark.parametrize("name,K,steps,lcpb", [("lorem4k", 96, 80, {}), ("enwik3k", 128, 80, {}), ("reps", 64, 80, {}), ("zeros600", 32, 40, {}),
                                               ("enwik3k", 128, 60, dict(lc=2, lp=1, pb=2))], ids=["lorem4k", "enwik3k", "reps", "zeros600", "enwik3k-lc2lp1pb2"])
def test_batch_accept_equals_rebuild(name, K, steps, lcpb, golden, golden_input, monkeypatch):
    """Bulk steps that take at most MGL_BATCH_MAX moves patch the base for all of them at once (mgl_kernels5.hip) instead of
    re-deriving it: after every such step bitmaps, special-state records, chains and dense checkpoints are identical to a
    rebuild from the slab, the cost is the oracle's walk of the slab, and a chain with the batch path switched off
    (MGL_NO_BATCH: every bulk step a rebuild) walks the same trajectory."""
    data = golden_input(name)
    inc = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=5, iters_per_epoch=10**7, **lcpb)
    monkeypatch.setenv("MGL_NO_BATCH", "1")
    full = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=5, iters_per_epoch=10**7, **lcpb)
    monkeypatch.delenv("MGL_NO_BATCH")
    ref = binding.SA(data, accept="single", neighbours_per_step=8, seed=5, **lcpb)
    o = Oracle(data, dict_limit=0x400000, **lcpb)
    moves = multi = 0
    for s in range(steps):
        st, sf = inc.run(1), full.run(1)
        assert st["current_cost"] == sf["current_cost"] and st["accepted"] == sf["accepted"] and st["best_cost"] == sf["best_cost"], (name, s)
        assert st["bulk_rollbacks"] == 0 and st["full_rebuilds"] == 0
        moves += st["accepted"]
        multi += st["accepted"] > 1
        if st["accepted"] and (s < 25 or s % 5 == 0):
            cur, cost = inc.current()
            assert (cur == full.current()[0]).all()
            assert cost == o.cost_slab(cur.astype(literal_slab(1).dtype))["total"], s
            ref.set_slab(cur)
            assert_same_base(canonical_base(inc, cur), canonical_base(ref, cur), (name, s))
    assert moves >= 10 and (multi >= 3 or name == "zeros600")   # (600 zeros: one window covers the file, one move per step)
    ia, ib = inc.batch_counters()
    assert ia >= 3                      # the batch path really ran
    assert full.batch_counters() == (0, 0)
    bst, _ = inc.best()
    assert lzma.decompress(binding.emit_stream(data, bst, **lcpb), format=lzma.FORMAT_ALONE) == data
    inc.close(); full.close(); ref.close()


def test_batch_accept_that_gives_up_falls_back_to_the_rebuild(monkeypatch):
    """The batch accept's own fallback -- it has written journals and bitmaps, then a capacity is exceeded while the chains are
    rewritten -- has never been taken by a real step; mgl_debug_set key 5 forces it.  The step must end exactly where the
    chain without the batch path ends (rebuild from the slab the commit left), and the next steps go on patching in place."""
    data = corpus.enwik_like(30000, 0x5151)
    a = binding.SA(data, accept="bulk", neighbours_per_step=256, seed=9, iters_per_epoch=10**7)
    monkeypatch.setenv("MGL_NO_BATCH", "1")
    b = binding.SA(data, accept="bulk", neighbours_per_step=256, seed=9, iters_per_epoch=10**7)
    monkeypatch.delenv("MGL_NO_BATCH")
    ref = binding.SA(data, accept="single", neighbours_per_step=8, seed=5)
    for x in (a, b):
        x.run(12)
    a.debug_set(5, 3)
    for s in range(8):
        sa_, sb_ = a.run(1), b.run(1)
        assert sa_["current_cost"] == sb_["current_cost"] and sa_["accepted"] == sb_["accepted"] and sa_["bulk_rollbacks"] == 0, s
        cur, cost = a.current()
        assert (cur == b.current()[0]).all()
        ref.set_slab(cur)
        assert_same_base(canonical_base(a, cur), canonical_base(ref, cur), ("forced", s))
    acc, fb = a.batch_counters()
    assert fb == 3 and acc >= 12 + 8 - 3 - 4   # three forced fallbacks; the other steps (but the first few, whose windows do not close) patched in place
    a.close(); b.close(); ref.close()


def test_full_size_c2_batch_path_vs_rebuild_path(monkeypatch):
    """BASELINE configs[1] at full size, 160 bulk steps (what the default accept mode runs there; the mode is pinned because AUTO's
    switch-over rule knows what a bulk step costs and so differs between the two chains): the chain whose bulk steps patch
    their moves in (at most MGL_BATCH_MAX of them, else rebuild) against the chain that always rebuilds -- same costs step
    by step, same slab, cost == the oracle's walk, stream round-trips."""
    data, _ = corpus.config_input("c2")
    a = binding.SA(data, neighbours_per_step=4096, seed=1673551, iters_per_epoch=len(data), accept="bulk")
    monkeypatch.setenv("MGL_NO_BATCH", "1")
    b = binding.SA(data, neighbours_per_step=4096, seed=1673551, iters_per_epoch=len(data), accept="bulk")
    monkeypatch.delenv("MGL_NO_BATCH")
    for s in range(16):
        sa_, sb_ = a.run(10), b.run(10)
        for k in ("current_cost", "best_cost", "accepted", "evaluations", "bulk_steps"):
            assert sa_[k] == sb_[k], (s, k)
    cur, cost = a.current()
    assert (cur == b.current()[0]).all()
    assert cost == Oracle(data, dict_limit=0x400000).cost_slab(cur.astype(literal_slab(1).dtype))["total"]
    assert a.batch_counters()[0] > 40 and a.batch_counters()[1] == 0
    assert lzma.decompress(binding.emit_stream(data, a.best()[0]), format=lzma.FORMAT_ALONE) == data
    a.close(); b.close()


@pytest.mark.parametrize("name,K", [("lorem4k", 64), ("enwik3k", 96)])
def test_epoch_snapshots_equal_rebuild(name, K, golden, golden_input):
    """mgl_sa_begin_epoch restores the all-literal / best base structures from device copies;
    the chain must be indistinguishable from one that re-derives them from the slab
    (MGL_F_NO_SNAPSHOTS), across main.c:69-77's phase/epoch schedule."""
    data = golden_input(name)
    steps = (len(data) + K - 1) // K
    a = binding.SA(data, accept="single", neighbours_per_step=K, seed=11, iters_per_epoch=len(data))
    b = binding.SA(data, accept="single", neighbours_per_step=K, seed=11, iters_per_epoch=len(data), snapshots=False)
    o = Oracle(data, dict_limit=0x400000)
    for phase in range(3):
        for epoch in range(3):
            for sa in (a, b):
                sa.begin_epoch(phase, from_best=phase != 0)
            ca, costa = a.current()
            cb, costb = b.current()
            assert costa == costb and (ca == cb).all(), (phase, epoch, "start")
            assert costa == o.cost_slab(ca.astype(literal_slab(1).dtype))["total"]
            assert_same_base(canonical_base(a, ca), canonical_base(b, cb), (name, phase, epoch, "restored"))
            sta, stb = a.run(steps), b.run(steps)
            for k in ("evaluations", "accepted", "improved", "current_cost", "best_cost", "packets"):
                assert sta[k] == stb[k], (phase, epoch, k)
            ca, _ = a.current()
            cb, _ = b.current()
            assert (ca == cb).all()
            assert_same_base(canonical_base(a, ca), canonical_base(b, cb), (name, phase, epoch, "end"))
    ba, bca = a.best()
    bb, bcb = b.best()
    assert bca == bcb and (ba == bb).all() and bca == o.cost_slab(ba.astype(literal_slab(1).dtype))["total"]
    # a best slab pushed in from outside (the multi-GPU exchange) invalidates the device copy
    lit = literal_slab(len(data))
    a.set_best(lit, o.cost_slab(lit)["total"])
    a.begin_epoch(1, from_best=True)
    cur, cost = a.current()
    assert (cur == lit.astype(cur.dtype)).all() and cost == o.cost_slab(lit)["total"]
    st = a.run(3)
    assert st["evaluations"] > 0
    a.close()
    b.close()


@pytest.mark.parametrize("cfg,size,K,steps,accept", [("c2", 100000, 1024, 40, "single"), ("c5", 70000, 256, 30, "single"), ("c2", 5000, 64, 60, "single"),
                                                     ("c2", 1025, 32, 20, "single"), ("c2", 16384, 64, 20, "single"), ("c2", 16385, 64, 20, "single"),
                                                     ("c2", 300001, 2048, 6, "bulk")])
def test_parallel_build_equals_serial_build(cfg, size, K, steps, accept):
    """mgl_pbuild.hip (block-parallel) and k_build (one wavefront) derive identical structures,
    totals and final walk state -- all-literal and SA-evolved slabs, pb = 0 and 2, sizes that end
    inside a block and one byte into a new one, block counts at and one past a group of 64 (the grouped entry
    chase), and 1 172 blocks = 19 groups / 147 chunks / 10 row groups (more than one trip of every staged loop),
    there on a slab thousands of bulk-step moves away from the literal one."""
    data, _ = corpus.config_input(cfg, size)
    props = dict(pb=2, max_bucket_scan=512) if cfg == "c5" else {}
    par = binding.SA(data, accept=accept, neighbours_per_step=K, seed=3, **props)
    ser = binding.SA(data, accept="single", neighbours_per_step=8, seed=3, serial_build=True, snapshots=False, **props)
    o = Oracle(data, dict_limit=0x400000, **({"pb": 2} if cfg == "c5" else {}))
    for round_ in range(2):
        cur, cost = par.current()
        ser.set_slab(cur)
        cur2, cost2 = ser.current()
        assert cost == cost2 and (cur == cur2).all()
        if size <= 5000 or round_ == 0:
            assert cost == o.cost_slab(cur.astype(literal_slab(1).dtype))["total"]
        par.set_slab(cur)  # rebuild `par` from scratch through the parallel builder
        cur3, cost3 = par.current()
        assert cost3 == cost
        assert_same_base(canonical_base(par, cur), canonical_base(ser, cur), (cfg, size, round_))
        par.run(steps)
    par.close()
    ser.close()


def test_parallel_build_serial_segment_path():
    """pb_sim starts a chain segment from a warm-up that brackets the probability; pb_sim_fix redoes
    the segments whose bracket did not close.  That never happens on these inputs, so force it
    (mgl_debug_set key 1) and require the same structures and cost."""
    data, _ = corpus.config_input("c2", 60000)
    a = binding.SA(data, accept="single", neighbours_per_step=512, seed=9)
    a.run(30)
    cur, cost = a.current()
    want = canonical_base(a, cur)
    a.L.mgl_debug_set(a.h, 1, 1)
    a.set_slab(cur)
    acc = a.debug_dump(11, np.uint64)
    assert int(acc[7]) > 10  # segments redone serially
    cur2, cost2 = a.current()
    assert cost2 == cost and (cur2 == cur).all()
    assert_same_base(canonical_base(a, cur), want, "forced pb_sim_fix")
    a.L.mgl_debug_set(a.h, 1, 0)
    a.set_slab(cur)
    assert int(a.debug_dump(11, np.uint64)[7]) == 0
    a.close()


def test_graph_replay_gives_the_same_trajectory(monkeypatch):
    """MGL_GRAPH=1 (opt-in): a step's neighbour launches -- three streams, their events -- are captured once as a HIP graph
    and replayed while nothing about them changes (launch form, buffers): same trajectory as the eager launches, in single
    and in bulk steps, through switches of the launch form (c2's early repair bursts) and across mgl_sa_run calls."""
    data, _ = corpus.config_input("c2")
    K = 4096
    monkeypatch.delenv("MGL_GRAPH", raising=False)
    eager = binding.SA(data, accept="auto", neighbours_per_step=K, seed=31)
    monkeypatch.setenv("MGL_GRAPH", "1")
    graph = binding.SA(data, accept="auto", neighbours_per_step=K, seed=31)
    monkeypatch.delenv("MGL_GRAPH")
    for chunk in (1, 7, 40, 64, 3, 90):
        a, b = eager.run(chunk), graph.run(chunk)
        for k in ("evaluations", "accepted", "current_cost", "best_cost", "packets", "failed", "bulk_steps"):
            assert a[k] == b[k], (chunk, k)
    for x in (eager, graph):
        x.set_accept_mode("single")
    for chunk in (5, 70, 20):
        a, b = eager.run(chunk), graph.run(chunk)
        for k in ("evaluations", "accepted", "current_cost", "best_cost", "packets", "failed"):
            assert a[k] == b[k], (chunk, k)
    ca, cb = eager.current(), graph.current()
    assert ca[1] == cb[1] and (ca[0] == cb[0]).all()
    eager.close(); graph.close()


def test_launch_forms_give_one_trajectory(monkeypatch):
    """The neighbour evaluation runs as two launches (pick + rest, with a second pass), as one kernel,
    or switches between them on the device (k_step_end): same costs for every neighbour, same
    trajectory, same final slab -- on full-size c2 through the early burst of repair picks."""
    data, _ = corpus.config_input("c2")
    K = 4096
    monkeypatch.delenv("MGL_NO_SPLIT", raising=False)
    monkeypatch.delenv("MGL_NO_ADAPT", raising=False)
    adaptive = binding.SA(data, accept="single", neighbours_per_step=K)
    monkeypatch.setenv("MGL_NO_ADAPT", "1")
    split = binding.SA(data, accept="single", neighbours_per_step=K)
    monkeypatch.delenv("MGL_NO_ADAPT")
    monkeypatch.setenv("MGL_NO_SPLIT", "1")
    single = binding.SA(data, accept="single", neighbours_per_step=K)
    monkeypatch.delenv("MGL_NO_SPLIT")
    for s in range(75):
        if s % 5 == 0 or 44 <= s <= 62:
            ca = adaptive.neighbours(s, want_diffs=False)[0]
            cs = split.neighbours(s, want_diffs=False)[0]
            c1 = single.neighbours(s, want_diffs=False)[0]
            assert (ca == c1).all() and (cs == c1).all(), s
        sts = [x.run(1) for x in (adaptive, split, single)]
        for k in ("evaluations", "accepted", "current_cost", "best_cost", "packets", "failed"):
            assert sts[0][k] == sts[2][k] and sts[1][k] == sts[2][k], (s, k)
    a, _ = adaptive.current()
    b, _ = split.current()
    c, _ = single.current()
    assert (a == c).all() and (b == c).all()
    for x in (adaptive, split, single):
        x.close()


@pytest.mark.parametrize("name,size,K,cap", [("lorem", 4096, 64, 0), ("enwik", 200000, 2048, 0), ("enwik_small_lists", 60000, 1024, 16)])
def test_look_ahead_gives_the_same_trajectory(monkeypatch, name, size, K, cap):
    """MGL_LOOKAHEAD=1 (opt-in): the next step's pick + window walk run beside this step's tail on the base as it is before
    the accept; k_la_check keeps what the accepted move cannot have touched and has the rest evaluated again.  Whatever is
    kept or redone, the chain must be the one the plain order produces: same statistics block by block (look-ahead works
    inside a mgl_sa_run call), same slab -- also with the first-pass lists shrunk so that the second pass sees entries from
    both the speculative launch and the fresh evaluations."""
    data = corpus.lorem(size) if name == "lorem" else corpus.enwik_like(size, 0x4C41)
    monkeypatch.setenv("MGL_NO_ADAPT", "1")  # the split form throughout: look-ahead only exists there
    monkeypatch.delenv("MGL_LOOKAHEAD", raising=False)
    # (the look-ahead draws targets as positions -- it keeps the speculative results whose draw lands on the same position after
    # the accept --, so the plain chain it is compared with is given the same rule)
    plain = binding.SA(data, accept="single", neighbours_per_step=K, seed=77, iters_per_epoch=10**7, flags=binding.F_POSITION_TARGETS)
    monkeypatch.setenv("MGL_LOOKAHEAD", "1")
    ahead = binding.SA(data, accept="single", neighbours_per_step=K, seed=77, iters_per_epoch=10**7)
    monkeypatch.delenv("MGL_LOOKAHEAD")
    if cap:
        for x in (plain, ahead):
            assert x.L.mgl_debug_set(x.h, 2, cap) == 0
    accepted = 0
    for chunk in (1, 2, 3, 8, 30, 64, 70):
        a, b = plain.run(chunk), ahead.run(chunk)
        for k in ("evaluations", "accepted", "improved", "current_cost", "best_cost", "packets", "failed", "dropped_neighbours"):
            assert a[k] == b[k], (chunk, k)
        accepted += a["accepted"]
    ca, _ = plain.current()
    cb, _ = ahead.current()
    assert (ca == cb).all() and accepted > 20
    o = Oracle(data, dict_limit=0x400000)
    assert ahead.current()[1] == o.cost_slab(cb.astype(literal_slab(1).dtype))["total"]
    plain.close()
    ahead.close()


def test_second_pass_takes_what_overflows_small_lists():
    """The first-pass change lists are sized by the step (1 024 events each way for small steps), so
    small tests rarely overflow them.  Shrunk to 16 events (mgl_debug_set key 2) most neighbours go
    through the second pass (lists in global memory) -- the trajectory must not notice."""
    data = corpus.lorem(3000)
    runs = []
    for cap in (0, 16):
        sa = binding.SA(data, accept="single", neighbours_per_step=96, seed=5, iters_per_epoch=60)
        if cap:
            assert sa.L.mgl_debug_set(sa.h, 2, cap) == 0
        costs, second = [], 0
        for _ in range(40):
            st = sa.run(1)
            costs.append(st["current_cost"])
            second += st["second_pass_neighbours"]
        cur, cost = sa.current()
        runs.append((costs, cost, [tuple(int(x) for x in r) for r in zip(cur["type"], cur["dist"], cur["len"])], second))
        assert sa.L.mgl_debug_set(sa.h, 2, 12) != 0  # not a multiple of 8
        sa.close()
    assert runs[0][:3] == runs[1][:3]
    assert runs[1][3] > 2 * runs[0][3] and runs[1][3] > 40 * 96 // 2  # most of the 3 840 neighbours took the second pass


def test_counting_kernel_gives_the_same_trajectory_and_counts_bytes():
    """mgl_debug_set key 4: the re-simulation kernel instance that adds up the bytes its loads ask for (bench.py's roofline
    block) computes the same costs, and what it counts is plausible: at least the change lists it read and at least one chain
    chunk per touched context."""
    data, _ = corpus.config_input("c2")
    a = binding.SA(data, accept="single", neighbours_per_step=2048, seed=7, timing=True)
    b = binding.SA(data, accept="single", neighbours_per_step=2048, seed=7, timing=True)
    a.run(30); b.run(30)
    b.debug_set(4, 1)
    sa_, sb_ = a.run(12), b.run(12)
    b.debug_set(4, 0)
    assert sa_["current_cost"] == sb_["current_cost"] and sa_["evaluations"] == sb_["evaluations"] and sa_["accepted"] == sb_["accepted"]
    assert sa_["sim_bytes_counted"] == 0
    if sb_["sim_launches"]:  # the split form ran (the device may have chosen the one-kernel form for some steps)
        assert sb_["gpu_ms_sim"] > 0 and sb_["sim_bytes_counted"] > 48 * 20 * sb_["evaluations"] // 12  # > 20 contexts x one chunk per evaluation, roughly
        assert sb_["sim_bytes_counted"] < 10**6 * sb_["evaluations"]
    assert (a.current()[0] == b.current()[0]).all()
    a.close(); b.close()
