#!/usr/bin/env python3
"""Long-run consistency: every N steps the device's current cost must equal an independent full walk
of its slab (mgl_cost_slab: the one-wavefront walk kernel), and no fallback / rebuild may have been
needed.  python tools/consistency_soak.py c2 30000 2500"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
data, _ = corpus.config_input(cfg)
K = {"c2": 4096, "c3": 16384, "c5": 4096}[cfg]
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=max(len(data), steps), **props)
done, t0 = 0, time.perf_counter()
while done < steps:
    st = sa.run(every)
    done += every
    cur, cost = sa.current()
    walk = sa.cost_slab(cur)["total"]
    assert walk == cost == st["current_cost"], (done, walk, cost)
    print(f"{cfg} step {done}: cost {cost} == full walk; packets {st['packets']}, accepted {st['accepted']}/{every}, rebuilds {st['full_rebuilds']}, "
          f"last-resort {st['fallback_neighbours']}, second pass {st['second_pass_neighbours']}, {time.perf_counter() - t0:.1f} s", flush=True)
bst, bc = sa.best()
assert sa.cost_slab(bst)["total"] == bc
print("best", bc, "ok")
