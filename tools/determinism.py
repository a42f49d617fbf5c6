#!/usr/bin/env python3
"""Two chains with the same seed but different launch orders (MGL_HALVES 2 / 3; on small inputs split form / one-kernel form)
must stay identical: every taken journal of a bulk step is written in parallel, every accepted move folded in place.
   python tools/determinism.py c3 [steps=1500] [chunk=50] [accept=auto] [greedy=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 50
accept = sys.argv[4] if len(sys.argv) > 4 else "auto"
greedy = int(sys.argv[5]) if len(sys.argv) > 5 else 0
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
def make(env):
    for k in ("MGL_HALVES", "MGL_NO_ADAPT", "MGL_NO_SPLIT", "MGL_LOOKAHEAD"): os.environ.pop(k, None)
    os.environ.update(env)
    sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=len(data), accept=accept, **props)
    if greedy:
        sa.seed_greedy(greedy)
    for k in env: os.environ.pop(k, None)
    return sa
big = len(data) > (1 << 20)
a = make({"MGL_HALVES": "2", "MGL_NO_ADAPT": "1"} if big else {"MGL_NO_ADAPT": "1"})
b_env = {"MGL_HALVES": "3", "MGL_NO_ADAPT": "1"} if big else {"MGL_NO_SPLIT": "1"}
if os.environ.get("MGL_DET_LOOKAHEAD"):  # second chain: the opt-in look-ahead instead (split form pinned)
    os.environ.pop("MGL_DET_LOOKAHEAD")
    b_env = {"MGL_LOOKAHEAD": "1", "MGL_NO_ADAPT": "1"}
b = make(b_env)
done = 0
while done < steps:
    sa_, sb_ = a.run(chunk), b.run(chunk); done += chunk
    same = all(sa_[k] == sb_[k] for k in ("current_cost", "best_cost", "accepted", "evaluations", "bulk_steps"))
    if not same or done % (10 * chunk) == 0:
        print(cfg, "step", done, "same" if same else "DIFFERENT", sa_["current_cost"], sb_["current_cost"], sa_["accepted"], sb_["accepted"], sa_["bulk_steps"], sb_["bulk_steps"], flush=True)
    if not same:
        sys.exit(1)
ca, _ = a.current(); cb, _ = b.current()
assert (ca == cb).all()
print(cfg, "identical over", done, "steps")
