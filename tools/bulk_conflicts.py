#!/usr/bin/env python3
"""Diagnostic: why improving neighbours of an early bulk step are not taken -- windows, soft ends, dependence on rep
distances -- and what other selection orders would take.   python tools/bulk_conflicts.py c2 [first_step=2] [n_steps=4]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
n = len(data)
sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=n, accept="bulk")
def conflict(a, b):  # (target, end, soft, dep)
    if a[0] > b[0]: a, b = b, a
    if a[0] == b[0]: return True
    return not (a[2] <= b[0] and (b[3] == 0 or a[1] <= b[0]))
def greedy(order, wins, ignore_dep=False, hard_only=False):
    taken = []
    for j in order:
        w = wins[j]
        ok = True
        for t in taken:
            a, b = (wins[t], w) if wins[t][0] <= w[0] else (w, wins[t])
            if a[0] == b[0]: ok = False; break
            if hard_only: c = not (a[1] <= b[0])
            elif ignore_dep: c = not (a[2] <= b[0])
            else: c = conflict(a, b)
            if c: ok = False; break
        if ok: taken.append(j)
    return taken
if first > 1: sa.run(first - 1)
for s in range(first, first + nsteps):
    cur = sa.current()[1]
    costs, _, _ = sa.neighbours(s - 1 + 0, want_diffs=False) if False else sa.neighbours(sa_step := (s - 1), want_diffs=False)
    win = sa.debug_dump(21, np.uint32).reshape(-1, 2).astype(np.int64)
    w2 = sa.debug_dump(22, np.uint32).astype(np.int64)
    soft, dep = w2 & 0x7FFFFFFF, w2 >> 31
    ok = (costs != binding.INVALID_COST) & (costs < cur) & (win[:, 1] < 0xFFFFFFFE)
    idx = np.nonzero(ok)[0]
    wins = {int(j): (int(win[j, 0]), int(win[j, 1]), int(soft[j]), int(dep[j])) for j in idx}
    gain = {int(j): int(cur - costs[j]) for j in idx}
    by_key = sorted(wins, key=lambda j: (int(costs[j]), j))
    hard = np.array([wins[j][1] - wins[j][0] for j in by_key]); sf = np.array([wins[j][2] - wins[j][0] for j in by_key])
    deps = np.array([wins[j][3] for j in by_key])
    t0 = greedy(by_key, wins); t1 = greedy(by_key, wins, ignore_dep=True); t2 = greedy(by_key, wins, hard_only=True)
    by_short = sorted(wins, key=lambda j: (wins[j][1] - wins[j][0], int(costs[j])))
    t3 = greedy(by_short, wins)
    by_density = sorted(wins, key=lambda j: (-gain[j] / max(1, wins[j][1] - wins[j][0]), j))
    t4 = greedy(by_density, wins)
    tot = lambda t: sum(gain[j] for j in t) / 16384
    print(f"step {s}: improving {len(idx)}  dep {deps.mean():.2f}  hard window bytes p50 {np.percentile(hard,50):.0f} p90 {np.percentile(hard,90):.0f} max {hard.max()}  soft p50 {np.percentile(sf,50):.0f} p90 {np.percentile(sf,90):.0f}")
    print(f"   rule as is (greedy by key): {len(t0)} taken, {tot(t0):.0f} B   | dep ignored (not valid): {len(t1)}, {tot(t1):.0f} B | hard ends only: {len(t2)}, {tot(t2):.0f} B | short windows first: {len(t3)}, {tot(t3):.0f} B | by gain per window byte: {len(t4)}, {tot(t4):.0f} B")
    sa.run(1)
