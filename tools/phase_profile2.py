#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the neighbour kernels on an evolved slab (MGL_F_PROFILE); with MGL_PROF_BIG=1 the
counters belong to the second pass instead of the regular launch.   python tools/phase_profile2.py c3"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, desc = corpus.config_input(cfg)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, timing=True, iters_per_epoch=len(data), flags=binding.F_PROFILE, **props)
done = 0
while done < 6000:
    p = sa.run(128); done += p["steps"]
    if p["bulk_steps"] == 0: break
sa.set_accept_mode("single")
base = sa.debug_dump(9, np.uint64).copy()
st = sa.run(16)
p = sa.debug_dump(9, np.uint64)
d = p[:32].astype(np.int64) - base[:32].astype(np.int64)
names = ["state_at", "model_at", "top-K", "window walk", "chain_sim", "repair: model", "repair: overlay sim", "repair: top-K"]
tot = float(d[:8].sum())
print(f"{cfg} after {done} steps ({'second pass' if os.environ.get('MGL_PROF_BIG') else 'regular launch'}): nbr {st['gpu_ms_neighbours']/16:.3f} ms/step, second-pass neighbours/step {st['second_pass_neighbours']/16:.1f}")
for i, nm in enumerate(names):
    cnt = int(d[8 + i])
    print(f"  {nm:12s} {100*d[i]/max(tot,1):5.1f}%  calls={cnt}  avg cycles/call={d[i]/max(1,cnt):.0f}  max(all time)={int(p[16+i])}")
raw = p[32:]
life = (raw & np.uint64(0xFFFFFFFFFF)).astype(np.float64)
ok = life > 0
print(f"  wave lifetime cycles (last written): n={ok.sum()} mean={life[ok].mean():.0f} p50={np.percentile(life[ok],50):.0f} p99={np.percentile(life[ok],99):.0f} max={life[ok].max():.0f}")
