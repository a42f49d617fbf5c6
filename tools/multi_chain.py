#!/usr/bin/env python3
"""Diagnostic: M independent SA chains on ONE GPU (one host thread and one HIP stream each; the C ABI
is one host thread per handle).  Per-step serial stretches of one chain (accept path, second pass,
kernel tails) are filled by the other chains' kernels.   python tools/multi_chain.py c2 2 2000"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus, multi_gpu
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
data, _ = corpus.config_input(cfg)
K = {"c2": 4096, "c3": 16384}[cfg]
for m in sorted({1, M}):
    chains = [binding.SA(data, neighbours_per_step=K, seed=multi_gpu.chain_seed(1673551, r), iters_per_epoch=len(data)) for r in range(m)]
    for c in chains:
        c.run(100)
    res = [None] * m
    def work(i):
        res[i] = chains[i].run(steps)
    th = [threading.Thread(target=work, args=(i,)) for i in range(m)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    ev = sum(r["evaluations"] for r in res)
    print(f"{cfg}: {m} chain(s) on one GPU: {ev / el / 1e6:.2f} M evals/s aggregate, {el / steps * 1e3:.3f} ms per step round; best costs {[r['best_cost'] for r in res]}", flush=True)
    for c in chains: c.close()
