#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, json
from megalania_amd import binding, corpus
from conftest import materialise
from test_gpu_incremental import canonical_base
from _libs import walk
g = json.load(open(os.path.join(ROOT, "tests/golden/reference_vectors.json")))
name, K, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
data = materialise(g["inputs"][name])
inc = binding.SA(data, neighbours_per_step=K, seed=5, iters_per_epoch=steps)
ref = binding.SA(data, neighbours_per_step=8, seed=5)
prev = inc.current()[0]
for s in range(steps):
    st = inc.run(1)
    if not st["accepted"]:
        continue
    cur, cost = inc.current()
    ref.set_slab(cur)
    a, b = canonical_base(inc, cur), canonical_base(ref, cur)
    bad = np.nonzero((a["ck"] != b["ck"]).any(axis=1))[0]
    chbad = [c for c, (x, y) in enumerate(zip(a["chains"], b["chains"])) if len(x[0]) != len(y[0]) or (x[0] != y[0]).any() or (x[1] != y[1]).any()]
    if len(bad) or chbad or (a["on"] != b["on"]).any() or (a["sp"] != b["sp"]).any() or (a["st"] != b["st"]).any():
        diff = np.nonzero((prev["type"] != cur["type"]) | (prev["dist"] != cur["dist"]) | (prev["len"] != cur["len"]))[0]
        print("step", s, "journal positions", diff.tolist())
        for p in diff: print("   ", p, tuple(prev[p]), "->", tuple(cur[p]))
        print("chains bad", chbad[:10], "ck rows bad", bad.tolist()[:10])
        w = walk(cur); wp = walk(prev)
        print("new walk tail", w[-8:], "old walk tail", wp[-8:])
        for r in bad[:3]:
            cols = np.nonzero(a["ck"][r] != b["ck"][r])[0]
            print(" row", r, "boundary new", next((p for p in w if p >= r*64), None), "old", next((p for p in wp if p >= r*64), None))
            for c in cols[:8]:
                print("    ctx", c, "inc", a["ck"][r][c], "ref", b["ck"][r][c], "chain tail pos", a["chains"][c][0][-4:], "ev", a["chains"][c][1][-4:])
        break
    prev = cur.copy()
else:
    print("no mismatch in", steps, "steps")
