#!/usr/bin/env python3
"""Diagnostic: a chain whose bulk steps patch the base (batch accept) against one that rebuilds it, step by step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from megalania_amd import binding, corpus
from _libs import Oracle, literal_slab
from test_gpu_incremental import canonical_base

name = sys.argv[1] if len(sys.argv) > 1 else "lorem"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 96
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
data = corpus.lorem(n) if name == "lorem" else corpus.enwik_like(n, 0x51)
inc = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=5, iters_per_epoch=10**7)
os.environ["MGL_NO_BATCH"] = "1"
full = binding.SA(data, accept="bulk", neighbours_per_step=K, seed=5, iters_per_epoch=10**7)
del os.environ["MGL_NO_BATCH"]
ref = binding.SA(data, accept="single", neighbours_per_step=8, seed=5)
o = Oracle(data, dict_limit=0x400000)
for s in range(steps):
    st, sf = inc.run(1), full.run(1)
    cur, cost = inc.current()
    curf, costf = full.current()
    oc = o.cost_slab(cur.astype(literal_slab(1).dtype))["total"]
    same_slab = bool((cur == curf).all())
    print(f"step {s}: accepted {st['accepted']}/{sf['accepted']} cost inc {cost} full {costf} oracle(inc slab) {oc} slab_same {same_slab} batch {inc.batch_counters()} hdr {inc.debug_dump(81, np.uint32)[:8]} acc {inc.debug_dump(82, np.int64)}", flush=True)
    ref.set_slab(cur)
    a, b = canonical_base(inc, cur), canonical_base(ref, cur)
    for k in ("on", "sp"):
        if not (a[k] == b[k]).all():
            print("  differs:", k, np.nonzero(a[k] != b[k])[0][:10])
    if a["st"].shape != b["st"].shape or not (a["st"] == b["st"]).all():
        print("  differs: sp_state", a["st"].shape, b["st"].shape)
    bad = [c for c, (x, y) in enumerate(zip(a["chains"], b["chains"])) if len(x[0]) != len(y[0]) or not (x[0] == y[0]).all() or not (x[1] == y[1]).all()]
    if bad:
        c = bad[0]
        x, y = a["chains"][c], b["chains"][c]
        print("  chains differ:", len(bad), "first ctx", c, "len", len(x[0]), len(y[0]))
        m = min(len(x[0]), len(y[0]))
        d = np.nonzero((x[0][:m] != y[0][:m]) | (x[1][:m] != y[1][:m]))[0]
        print("   first diff idx", d[:5], "inc", list(zip(x[0][d[:3]], x[1][d[:3]])) if len(d) else None, "ref", list(zip(y[0][d[:3]], y[1][d[:3]])) if len(d) else None)
    ckbad = np.nonzero((a["ck"] != b["ck"]).any(axis=1))[0]
    if len(ckbad):
        print("  checkpoints differ at rows", ckbad[:8])
    if cost != costf or not same_slab or bad or len(ckbad):
        break
