#!/usr/bin/env python3
"""Search progress of one chain: estimated / real stream size against steps, evaluations and wall time.
  python tools/progress_curve.py c2 [accept=auto|single|bulk] [steps=2000] [K]"""
import json, lzma, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megalania_amd import binding, corpus

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
accept = sys.argv[2] if len(sys.argv) > 2 else "auto"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
K = int(sys.argv[4]) if len(sys.argv) > 4 else {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, desc = corpus.config_input(cfg)
n = len(data)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=n, accept=accept, **props)
marks = sorted({1, 2, 4, 8, 16, 24, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, steps})
done = evals = acc = bulk = 0
t0 = time.perf_counter()
pts = []
for m in marks:
    if m > steps:
        break
    st = sa.run(m - done)
    done = m
    evals += st["evaluations"]; acc += st["accepted"]; bulk += st["bulk_steps"]
    pts.append(dict(steps=done, evaluations=evals, accepted=acc, bulk_steps=bulk, est_bytes=round(18 + st["best_cost"] / 16384, 1),
                    seconds=round(time.perf_counter() - t0, 3), packets=st["packets"]))
    print(pts[-1], flush=True)
best, cost = sa.best()
lcpb = {k: props[k] for k in ("lc", "lp", "pb") if k in props}
out = dict(config=cfg, input=desc, n=n, K=K, accept=accept, points=pts)
if n <= (32 << 20):
    stream = binding.emit_stream(data, best, **lcpb)
    out["stream_bytes"] = len(stream)
    out["roundtrip"] = lzma.decompress(stream, format=lzma.FORMAT_ALONE) == bytes(data)
    out["cost_equals_full_walk"] = sa.cost_slab(best, want_cum=False)["total"] == cost
print(json.dumps(out))
