#!/usr/bin/env python3
"""c4 (100 MB) timings: create, bulk steps, single steps, greedy seed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
t=time.time(); data,_=corpus.config_input("c4"); print("input", round(time.time()-t,2), flush=True)
t=time.time(); sa=binding.SA(data, neighbours_per_step=16384, iters_per_epoch=len(data), timing=True); print("create", round(time.time()-t,2), flush=True)
for mode, steps in (("bulk", 6), ("single", 20)):
    sa.set_accept_mode(mode)
    t=time.time(); st=sa.run(steps); dt=time.time()-t
    print(mode, steps, "steps", round(dt/steps*1e3,2), "ms/step; nbr", round(st["gpu_ms_neighbours"]/steps,3), "apply", round(st["gpu_ms_rebuild"]/steps,3), "accepted", st["accepted"], "est MB", round(st["best_cost"]/16384/1e6,3), flush=True)
t=time.time(); sa.seed_greedy(64); print("greedy seed", round(time.time()-t,2), "est MB", round(sa.current()[1]/16384/1e6,3) if False else "", flush=True)
sa.set_accept_mode("single")
t=time.time(); st=sa.run(20); dt=time.time()-t
print("single after seed", round(dt/20*1e3,2), "ms/step; nbr", round(st["gpu_ms_neighbours"]/20,3), "apply", round(st["gpu_ms_rebuild"]/20,3), "cur MB", round(st["current_cost"]/16384/1e6,3), "imp/step", st["improving_neighbours"]/20, "2nd", st["second_pass_neighbours"], flush=True)
sa.set_accept_mode("bulk")
t=time.time(); st=sa.run(6); dt=time.time()-t
print("bulk after seed", round(dt/6*1e3,2), "ms/step; accepted", st["accepted"], "cur MB", round(st["current_cost"]/16384/1e6,3), flush=True)
sa.close()
