import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
data,_=corpus.config_input("c5")
sa=binding.SA(data, neighbours_per_step=4096, pb=2, max_bucket_scan=4096, iters_per_epoch=len(data), timing=True, accept="bulk")
for rnd in range(12):
    sa.set_accept_mode("bulk")
    t=time.time(); st=sa.run(25); dt=time.time()-t
    print(rnd,"bulk", round(dt/25*1e3,2),"ms/step nbr",round(st["gpu_ms_neighbours"]/25,2),"apply",round(st["gpu_ms_rebuild"]/25,2),"acc",st["accepted"],"2nd",st["second_pass_neighbours"],"fb",st["fallback_neighbours"],"drop",st["dropped_neighbours"],"pk",st["packets"], flush=True)
    sa.set_accept_mode("single")
    t=time.time(); st=sa.run(5); dt=time.time()-t
    print(rnd,"single", round(dt/5*1e3,2),"ms/step nbr",round(st["gpu_ms_neighbours"]/5,2),"apply",round(st["gpu_ms_rebuild"]/5,2),"2nd",st["second_pass_neighbours"],"fb",st["fallback_neighbours"],"drop",st["dropped_neighbours"], flush=True)
    if dt/5 > 0.03: break
sa.close()
