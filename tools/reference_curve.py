#!/usr/bin/env python3
"""Size-vs-iterations curve of the compiled reference (oracle/_ref: main.c:78-102 under glibc rand(), seed
1673551, phase 0, from the all-literal slab) on a BASELINE config input.  Run in the build container
(the reference does not travel); the result is committed as tests/golden/reference_curve_<cfg>.json and
read by bench.py's equal-budget gate."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs
from megalania_amd import corpus

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
marks = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1000,3000,10000,30000,100000".split(","))]
data, desc = corpus.config_input(cfg)
n = len(data)
eng = _libs.Ref(data)
SEED = int(os.environ.get("MGL_CURVE_SEED", "1673551"))  # main.c:68; other seeds: the reference's own spread (tests/golden/reference_spread_c2.json)
_libs.Ref.lib().ref_srand(SEED)
slab, best = _libs.literal_slab(n), _libs.literal_slab(n)
cur = bst = 0
done, t0, pts = 0, time.perf_counter(), []
for m in marks:
    while done < m:
        step = min(200, m - done)
        r = eng.sa_iters(slab, best, cur, bst, 0, n, done, done + step)
        cur, bst = r["cur"], r["best"]
        done += step
    stream = len(eng.emit(best))
    pts.append(dict(iterations=done, best_cost=bst, est_bytes=18 + bst / 16384, stream_bytes=stream, seconds=time.perf_counter() - t0))
    print(pts[-1], file=sys.stderr, flush=True)
out = dict(config=cfg, input=desc, n=n, seed=SEED, phase=0, source="oracle/_ref (compiled from /root/reference/src, glibc rand())", points=pts)
json.dump(out, open(os.environ.get("MGL_CURVE_OUT") or os.path.join(ROOT, "tests", "golden", f"reference_curve_{cfg}.json"), "w"), indent=1)
