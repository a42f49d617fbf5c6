#!/usr/bin/env python3
"""A/B helper: evolved-slab single-step time.  python tools/ab_single.py c3 [prepare_steps=700] [steps=60]  (env knobs apply)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
prep = int(sys.argv[2]) if len(sys.argv) > 2 else 700
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=len(data), timing=True, accept="bulk")
sa.run(prep)
sa.set_accept_mode("single")
sa.run(40)
t = time.perf_counter(); st = sa.run(steps); dt = time.perf_counter() - t
print(f"{cfg} knobs={ {k:v for k,v in os.environ.items() if k.startswith('MGL_') and k!='MGL_NO_AUTOBUILD'} }: {dt/steps*1e3:.3f} ms/step wall, nbr {st['gpu_ms_neighbours']/steps:.3f}, apply {st['gpu_ms_rebuild']/steps:.3f}, {st['evaluations']/dt/1e6:.2f} M evals/s, 2nd pass/step {st['second_pass_neighbours']/steps:.1f}")
sa.close()
