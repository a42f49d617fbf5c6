#!/usr/bin/env python3
"""Long run in the default accept mode with periodic consistency checks: device cost == the device's independent full
walk of the exported slab, the slab validates on a second handle (every packet reproduces the input), the stream decodes.
  python tools/soak_auto.py c2 [steps=6000] [every=500]"""
import lzma, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=len(data), **props)
chk = binding.SA(data, neighbours_per_step=8, accept="single", **props)
lcpb = {k: props[k] for k in ("pb",) if k in props}
done, t0 = 0, time.time()
tot = dict(bulk_steps=0, bulk_rollbacks=0, accepted=0, dropped_neighbours=0, full_rebuilds=0, fallback_neighbours=0)
while done < steps:
    st = sa.run(every); done += every
    for k in tot: tot[k] += st[k]
    cur, cost = sa.current()
    assert sa.cost_slab(cur, want_cum=False)["total"] == cost, "cost mismatch at %d" % done
    chk.set_slab(cur)
    best, bcost = sa.best()
    stream = binding.emit_stream(data, best, **lcpb)
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    print(f"{cfg} step {done}: est {18 + bcost / 16384:.1f} B stream {len(stream)} {tot} {time.time() - t0:.1f}s ok", flush=True)
# epochs from the best slab (main.c:75-77) keep working after bulk steps
for ph in (1, 2):
    sa.begin_epoch(ph, from_best=True)
    st = sa.run(200)
    cur, cost = sa.current()
    assert sa.cost_slab(cur, want_cum=False)["total"] == cost
print("soak ok")
