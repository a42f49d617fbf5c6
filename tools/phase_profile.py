#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_neighbours2 (MGL_F_PROFILE).  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
import ctypes as C

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
data, desc = corpus.config_input(cfg)
K = 4096
sa = binding.SA(data, neighbours_per_step=K, timing=True)
sa.cfg.flags |= 4
sa.close()
sa = binding.SA.__new__(binding.SA)
binding.SA.__init__(sa, data, neighbours_per_step=K, timing=True)
# re-create with the profile flag
sa.close()
L = binding.hip_lib()
c = binding.Config(1673551, K, 20, 0, 0, 0, 0, 1 | 4)
buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
h = L.mgl_sa_create(buf.ctypes.data_as(C.c_void_p), len(buf), binding.Properties(0, 0, 0), C.byref(c))
sa.h, sa.cfg, sa.data, sa.n, sa.K = h, c, buf, len(buf), K
if warm:
    sa.run(warm)
z = np.zeros(16, dtype=np.uint64)
st = sa.run(8)
p = sa.debug_dump(9, np.uint64)
names = ["state_at", "model_at", "top-K", "window walk", "chain_sim"]
tot = float(p[:5].sum())
print(f"{cfg}: {desc}; after {warm} warm steps; packets={st['packets']}; nbr kernel avg {st['gpu_ms_neighbours']/8:.3f} ms; rebuild avg {st['gpu_ms_rebuild']/8:.3f} ms")
for i, nm in enumerate(names):
    cnt = int(p[8 + i])
    print(f"  {nm:12s} {100*p[i]/tot:5.1f}%  calls={cnt}  avg cycles/call={p[i]/max(1,cnt):.0f}  max={int(p[16+i])}")
print("  fallback (full-walk) neighbours in last step:", int(sa.debug_dump(10, np.uint32)[0]))
print("  chain_sim max iterations:", int(p[21]), "ctx", int(p[22]), "chain len", int(p[23]) >> 32, "pending flags", int(p[23]) & 3, "n_ins", (int(p[23]) >> 8) & 0xFFF, "n_rem", (int(p[23]) >> 20) & 0xFFF)
raw = p[32:]
life = (raw & np.uint64(0xFFFFFFFFFF)).astype(np.float64)
events = ((raw >> np.uint64(40)) & np.uint64(0xFFF)).astype(np.int64)
walked = ((raw >> np.uint64(52)) & np.uint64(0xFFF)).astype(np.int64)
ok = life > 0
life, events, walked = life[ok], events[ok], walked[ok]
print(f"  wave lifetime cycles (last launch): n={len(life)} mean={life.mean():.0f} p50={np.percentile(life,50):.0f} p90={np.percentile(life,90):.0f} p99={np.percentile(life,99):.0f} max={life.max():.0f}")
print(f"  change events per neighbour: mean={events.mean():.0f} p90={np.percentile(events,90):.0f} max={events.max()};  packets walked: mean={walked.mean():.1f} max={walked.max()}")
order = np.argsort(-life)[:12]
print("  slowest waves (cycles, events, packets walked):", [(int(life[i]), int(events[i]), int(walked[i])) for i in order])
print("  correlation lifetime~events:", round(float(np.corrcoef(life, events)[0, 1]), 3), " lifetime~walked:", round(float(np.corrcoef(life, walked)[0, 1]), 3))
