#!/usr/bin/env python3
"""bench.py -- SA neighbour-cost evaluations per second on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N=1: plain python; N>1: torchrun)

One "step" = one pass of the hot path over one batch: K_nb candidate neighbours of the current slab
generated and costed exactly (each cost = the u64 perplexity of one complete parse of the input), one
decision, the base structures brought up to date.

Workload.  N = 1: BASELINE configs[2], the largest single-GPU configuration -- dickens-shaped 10 192 446 B
(synthetic, seeded: no corpora exist offline), 16 384 neighbours/step, lc=lp=pb=0, top-K 20.  N > 1:
configs[3], one independent chain per GPU on an enwik8-shaped 100 000 000 B input, 16 384 neighbours/step
(weak scaling, no data-path collective).  The per-epoch best-slab exchange (one 8-byte RCCL all-reduce + one
broadcast of the packed slab, from the C library: configs[3] runs it every 10 000 steps) is run once, right after
the timed steps, and timed on its own ("exchange"): it is not a step.  `--config` selects another workload.

State measured.  The configs are long runs (10^5 .. 10^6 steps); what such a run does almost all of the time is
step an *evolved* slab.  So the set-up phase first runs `--prepare-steps` search steps from the all-literal slab
in the library's default accept mode (bulk steps while they pay, then single steps; untimed, reported under
"prepare"), then W untimed warm-up steps, then exactly K timed steps, bracketed by barrier + device synchronise.
At 100 MB the bulk phase of a search from the all-literal slab lasts minutes (a bulk step re-derives 28 GB of
structures): there the set-up is the library's greedy seed (mgl_sa_seed_greedy, an evolved parse in half a second)
and the timed steps are single-accept steps; the line says so.
`value` = neighbour evaluations that produced a cost, summed over ranks, / max-over-ranks wall time.  The rate on
the young slab (first steps from the all-literal slab) is reported beside it under "young_slab".

`roofline`: the dominant kernels are the three launches of the incremental neighbour evaluation (pick, window
walk, chain re-simulation).  `achieved` = their HBM traffic per step from rocprofv3 PMC counters (FETCH_SIZE and
WRITE_SIZE in their own passes, fetch doubled per the gfx950 note = an upper bound; profiles/<tag>_pmc_<cfg>.json,
raw rows beside it) / their duration per step measured live here with HIP events on the library's streams;
`frac` = achieved / 8 TB/s.  The kernels are latency-bound, not HBM-bound: the wait / issue counters of the same
profile are copied into `limiter`.  `algorithmic_equiv_gbs` is SURVEY 8d's figure (N + 12 P bytes per evaluation,
as if every evaluation streamed the whole parse, which the incremental kernels do not do): not a fraction of anything.
`cpu_baseline`: the compiled reference (oracle/_ref, kind "reference") or else the CPU oracle (kind "port") on a
bounded sample of the same input, rank 0, N = 1 only.  `gates`: SURVEY 8d's correctness gates, taken after the
timed region, and the size comparison with the reference at equal evaluations and at equal steps.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
PROFILE_TAG = "r03"
DEFAULT_K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}
PREPARE_CAP = {"c1": 400, "c2": 1500, "c3": 6000, "c4": 1500, "c5": 2000}  # steps; the bulk phase normally ends well before
DOMINANT = ("k_neighbours2", "k_sim")  # pick, walk, re-simulation, their one-kernel form and the second pass
SIM = "k_sim"  # the dominant kernel: the chain re-simulation (2 + 1 launches per step of the split form)


def source_hash():
    """What a stored rocprofv3 profile was taken of: the kernels' sources (tools/pmc_to_json.py stamps the same hash)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "megalania_amd", "csrc")
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def _ref_engine(data):
    import _libs

    if _libs.Ref.available():
        return _libs.Ref(data), "reference", _libs.Ref.lib().ref_srand
    return _libs.Oracle(data), "port", _libs.Oracle.lib().orc_srand


def cpu_baseline(data, seconds):
    """Reference-semantics SA iterations (main.c:78-102) on one host thread, from the all-literal slab."""
    import _libs

    n = len(data)
    eng, kind, seed_fn = _ref_engine(data)
    slab, best = _libs.literal_slab(n), _libs.literal_slab(n)
    seed_fn(1673551)
    cur = bst = 0
    done = 0
    chunk = 100 if n <= (1 << 20) else 4
    t0 = time.perf_counter()
    while True:
        r = eng.sa_iters(slab, best, cur, bst, 0, n, done, done + chunk)
        cur, bst = r["cur"], r["best"]
        done += chunk
        el = time.perf_counter() - t0
        if el >= seconds or done >= 40000:
            break
    return dict(value=done / el, unit="evals/s", cores=1, kind=kind, nproc=os.cpu_count(), cpu=cpu_model(),
                sample=f"{done} SA iterations (seed 1673551, from the all-literal slab) of the same "
                       f"{n} B input in {el:.1f} s on 1 thread", iterations=done, best_cost=bst, est_bytes_best=18 + bst / 16384)


def _cpu_worker(args):
    data, seconds, seed = args
    import _libs

    n = len(data)
    eng, _, seed_fn = _ref_engine(data)
    seed_fn(seed)
    slab, best = _libs.literal_slab(n), _libs.literal_slab(n)
    cur = bst = 0
    chunk = 100 if n <= (1 << 20) else 2
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        r = eng.sa_iters(slab, best, cur, bst, 0, n, done, done + chunk)
        cur, bst = r["cur"], r["best"]
        done += chunk
    return done, time.perf_counter() - t0


def cpu_baseline_all_cores(data, seconds):
    """Independent chains, one process per host core (the reference is not re-entrant: global rand(), stateful finder)."""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    # each worker holds the slab pair (24 n bytes) and the reference's index (8 n): bound the total
    per_worker = 48 * len(data) + (64 << 20)
    cores = max(1, min(cores, int((32 << 30) // per_worker)))
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(data, seconds, 1673551 + i) for i in range(cores)])
    total = sum(d for d, _ in res)
    wall = max(t for _, t in res)
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libmegalania_ref.so")) else "port"
    return dict(value=total / wall, unit="evals/s", cores=cores, nproc=os.cpu_count(), cpu=cpu_model(), kind=kind,
                per_core=total / wall / cores,
                sample=f"{total} SA iterations over {cores} independent chains (one process per core) in {wall:.1f} s")


def load_pmc(cfg):
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_{cfg}.json")
    if not os.path.exists(path):
        return None, path
    with open(path) as f:
        return json.load(f), path


CONFIG_ITERS = {"c1": 1000, "c2": 100000, "c3": 1000000}  # BASELINE.json configs: SA iterations named per config


def reference_curve(cfg):
    path = os.path.join(ROOT, "tests", "golden", f"reference_curve_{cfg}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def reference_spread(cfg):
    """The compiled reference's own spread over srand seeds on this input (tests/golden/reference_spread_<cfg>.json,
    MGL_CURVE_SEED=n tools/reference_curve.py): {iterations: [est_bytes per seed]}."""
    path = os.path.join(ROOT, "tests", "golden", f"reference_spread_{cfg}.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        d = json.load(f)
    out = {}
    for run in d["runs"]:
        for p in run["points"]:
            out.setdefault(p["iterations"], []).append(p["est_bytes"])
    return out


def run_to_marks(sa, K, ev_marks, st_marks=()):
    """Run a chain until it has made each number of evaluations in `ev_marks` -- successful ones, as the reference's
    iteration counter counts them (main.c:81-84 retries a failed generate without counting it) -- and each number of
    steps in `st_marks`; returns {mark: (est_bytes, evaluations, steps)} for both."""
    ev_marks, st_marks = sorted(set(ev_marks)), sorted(set(st_marks))
    at_ev, at_st, done, evals, est = {}, {}, 0, 0, None
    while True:
        while ev_marks and evals >= ev_marks[0]:
            at_ev[ev_marks.pop(0)] = (est, evals, done)
        while st_marks and done >= st_marks[0]:
            at_st[st_marks.pop(0)] = (est, evals, done)
        if not ev_marks and not st_marks:
            return at_ev, at_st
        want = []
        if ev_marks:
            want.append(max(1, -(-(ev_marks[0] - evals) // K)))
        if st_marks:
            want.append(st_marks[0] - done)
        st = sa.run(min(want))
        done += st["steps"]
        evals += st["evaluations"]
        est = 18 + st["best_cost"] / 16384


def seed_spread_gate(binding, data, K, props, cfg, accept="auto", seeds=(1673551, 1673551 + 7919, 1673551 + 2 * 7919)):
    """Equal evaluations, means over seeds on both sides: the device's chains (three seeds) against the reference's runs
    (the golden curve's seed and those of reference_spread_<cfg>.json)."""
    spread = reference_spread(cfg)
    curve = reference_curve(cfg)
    if not spread or not curve:
        return None
    for p in curve["points"]:
        if p["iterations"] in spread:
            spread[p["iterations"]] = spread[p["iterations"]] + [p["est_bytes"]]
    marks = sorted(spread)
    rows = {m: [] for m in marks}
    evs = {m: [] for m in marks}
    n = len(data)
    for sd in seeds:
        sa = binding.SA(data, neighbours_per_step=K, seed=sd, iters_per_epoch=n, accept=accept, **props)
        at_ev, _ = run_to_marks(sa, K, marks)
        for m in marks:
            rows[m].append(at_ev[m][0])
            evs[m].append(at_ev[m][1])
        sa.close()
    out = []
    for m in marks:
        g, r = rows[m], spread[m]
        out.append(dict(evaluations=m, gpu_evaluations=evs[m], gpu_seeds=len(g), gpu_mean=round(sum(g) / len(g), 1), gpu_min=round(min(g), 1), gpu_max=round(max(g), 1),
                        reference_seeds=len(r), reference_mean=round(sum(r) / len(r), 1), reference_min=round(min(r), 1), reference_max=round(max(r), 1),
                        gpu_mean_le_reference_mean=sum(g) / len(g) <= sum(r) / len(r), gpu_mean_le_reference_max=sum(g) / len(g) <= max(r)))
    return out


def size_gates(binding, data, K, props, cpu, cfg, accept="auto"):
    """Estimated stream size (18 + perplexity / 16384, main.c:97) of a fresh chain in the default accept mode against
    the reference path's at (a) equal evaluations and (b) equal steps = reference iterations, both stated."""
    pts = []
    curve = reference_curve(cfg)
    if curve:
        pts += [(p["iterations"], p["est_bytes"], "tests/golden/reference_curve_%s.json (compiled reference, this input)" % cfg)
                for p in curve["points"]]
    if cpu and cpu.get("iterations"):
        pts.append((cpu["iterations"], cpu["est_bytes_best"], "this run's cpu_baseline sample"))
    if not pts:
        return None
    n = len(data)
    budget = CONFIG_ITERS.get(cfg)
    if budget and all(it != budget for it, _, _ in pts):
        pts.append((budget, None, "no reference figure: BASELINE.json's iteration budget for this config (the reference path would need days)"))
    sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=n, accept=accept, **props)
    at_ev, at_st = run_to_marks(sa, K, [it for it, _, _ in pts], [it for it, ref, _ in pts if it <= 4096 and ref is not None])
    sa.close()
    out = []
    for it, ref_bytes, src in pts:
        est, evals, steps = at_ev[it]
        row = dict(reference_iterations=it, reference_est_bytes=None if ref_bytes is None else round(ref_bytes, 1), reference_source=src, accept_mode=accept,
                   config_budget=(it == budget),
                   equal_evaluations=dict(gpu_steps=steps, gpu_evaluations=evals, gpu_est_bytes=round(est, 1),
                                          gpu_le_reference=None if ref_bytes is None else est <= ref_bytes))
        if it in at_st:
            est, evals, steps = at_st[it]
            row["equal_steps"] = dict(gpu_steps=steps, gpu_evaluations=evals, gpu_est_bytes=round(est, 1), gpu_le_reference=est <= ref_bytes)
        out.append(row)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default=None)
    ap.add_argument("--size", type=int, default=0, help="override the input size (bytes)")
    ap.add_argument("--neighbours", type=int, default=0)
    ap.add_argument("--prepare-steps", type=int, default=-1,
                    help="search steps of the set-up phase (default: until the library's bulk phase is over, capped per config and at --prepare-seconds)")
    ap.add_argument("--prepare-seconds", type=float, default=90.0)
    ap.add_argument("--accept", default=None, choices=["auto", "single", "bulk"],
                    help="accept mode of the warm-up and timed steps (default: single; the set-up phase always runs in auto)")
    ap.add_argument("--greedy-seed", type=int, default=-1, help="set-up from the greedy seed with this many candidates (default: 64 above 32 MB, else off)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the young-slab rate and the size gates")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher of N rank processes (one per GPU) and
        # never touches a GPU itself (torch.cuda.device_count() does not initialise the runtime); the ranks print the
        # one JSON line.  With fewer devices than ranks (a one-GPU box) the ranks share GPU 0 and the exchange goes
        # through the C library's host transport -- a rehearsal of the code path, and the line says so.
        import socket
        import subprocess

        import torch

        env = dict(os.environ)
        if torch.cuda.device_count() < args.gpus:
            env["MGL_BENCH_SHARE_GPU"] = "1"
            env.setdefault("MGL_BENCH_BACKEND", "gloo")
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    # stdout carries the one JSON line and nothing else: whatever libraries write to descriptor 1 meanwhile (gloo's connection
    # notes, a make run of the in-tree build) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs", file=sys.stderr)
    dist = None
    backend = None
    shared_gpu = False
    if world > 1:
        import torch
        import torch.distributed as dist

        # rehearsal on a one-GPU box: MGL_BENCH_BACKEND=gloo MGL_BENCH_SHARE_GPU=1 runs the same code path with every
        # rank on GPU 0; the exchange is still the C library's (mgl_sa_exchange_best), over its host shared-memory
        # transport instead of RCCL (which refuses two ranks per device); torch.distributed only carries barriers
        backend = os.environ.get("MGL_BENCH_BACKEND", "nccl")
        if os.environ.get("MGL_BENCH_SHARE_GPU"):
            local_rank = 0
            shared_gpu = True
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    n_gpus = world if world > 1 else 1
    cfg = args.config or ("c3" if n_gpus == 1 else "c4")

    from megalania_amd import binding, build as _build, corpus, multi_gpu

    if int(os.environ.get("LOCAL_RANK", "0")) == 0 and not os.environ.get("MGL_NO_AUTOBUILD"):
        _build.build_all()  # in-tree native build, no-op when current (there is no CPU search path to fall back to)
    if dist is not None:
        dist.barrier()

    data, desc = corpus.config_input(cfg, args.size or None)
    n = len(data)
    K = args.neighbours or DEFAULT_K[cfg]
    # c5 (ELF-shaped): inside long zero runs a top-K query has > 10^6 candidates in the reference; the
    # bench caps the bucket scan at the 4096 nearest hits and says so.
    props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
    big_input = n > (32 << 20)
    greedy = args.greedy_seed if args.greedy_seed >= 0 else (64 if big_input else 0)
    # the timed steps are single-accept steps: once the bulk phase is over the library's AUTO mode still takes the odd
    # block of bulk steps when improving neighbours pile up around its threshold, which would make a 20-step timed
    # region a coin toss between two very different step costs; the long run's steady state is the single step
    accept = args.accept or "single"
    sa = binding.SA(data, neighbours_per_step=K, seed=multi_gpu.chain_seed(1673551, rank), iters_per_epoch=n,
                    device=local_rank, timing=True, accept="auto", **props)
    comm = None
    if dist is not None and backend == "nccl":
        comm = multi_gpu.make_comm(dist, rank, world, local_rank)  # the C library's own RCCL communicator
    elif dist is not None and shared_gpu:
        box = [(f"/dev/shm/mgl_bench_{os.getpid()}", int.from_bytes(os.urandom(7), "little")) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = binding.Comm.shm(box[0][0], box[0][1], rank, world, local_rank)  # the C library's host transport

    def sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # set-up: the chain as a long run has it.  By default: until a whole chunk of steps ran (almost) without a bulk step
    # (the library's AUTO mode has left its bulk phase), every rank for itself
    t_prep = time.perf_counter()
    prep = None
    if greedy:
        sa.seed_greedy(greedy)
        _, seed_cost = sa.current()
        prep = dict(steps=0, evaluations=0, bulk_steps=0, accepted=0, bulk_rollbacks=0, best_cost=seed_cost, greedy_candidates=greedy)
    elif args.prepare_steps != 0:
        prep = dict(steps=0, evaluations=0, bulk_steps=0, accepted=0, bulk_rollbacks=0, best_cost=0)
        cap = args.prepare_steps if args.prepare_steps > 0 else PREPARE_CAP[cfg]
        chunk = cap if args.prepare_steps > 0 else (64 if n <= (1 << 20) else 128)
        while prep["steps"] < cap:
            p = sa.run(min(chunk, cap - prep["steps"]))
            for k in ("steps", "evaluations", "bulk_steps", "accepted", "bulk_rollbacks"):
                prep[k] += p[k]
            prep["best_cost"] = p["best_cost"]
            # over: a chunk with (almost) no bulk step -- around its threshold AUTO still takes the odd block of four
            if args.prepare_steps < 0 and (p["bulk_steps"] * 16 <= p["steps"] or time.perf_counter() - t_prep > args.prepare_seconds):
                break
    prepare = prep["steps"] if prep else 0
    t_prep = time.perf_counter() - t_prep
    sa.set_accept_mode(accept)
    sa.run(args.warmup)
    sync()
    t0 = time.perf_counter()
    st = sa.run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    exchange = None
    if dist is not None:
        t1 = time.perf_counter()
        if comm is not None:
            winner, wcost = multi_gpu.exchange_best_native(sa, comm)
        else:
            winner, wcost = multi_gpu.exchange_best(sa, dist, device=coll_device)
        sync()
        t1 = time.perf_counter() - t1
        exchange = {"ms": t1 * 1e3, "winner_rank": winner, "winner_est_bytes": 18 + wcost / 16384, "slab_bytes_broadcast": 8 * n,
                    "transport": ("RCCL from the C library (mgl_sa_exchange_best)" if backend == "nccl" else
                                  "the C library's host shared-memory transport (mgl_sa_exchange_best; ranks share one GPU: a rehearsal)") if comm is not None
                                 else f"torch.distributed/{backend} through host memory",
                    "amortised_ms_per_step_at_10000_steps_per_exchange": t1 * 1e3 / 10000}

    # the dominant kernel's algorithmic bytes, counted by the kernel itself (k_sim<true>: every chain position / event /
    # list entry its loads ask for) over a few more steps of the same state, outside the timed region
    counted = None
    if rank == 0 and st.get("sim_launches"):
        sa.debug_set(4, 1)
        for _ in range(4):  # (the device now and then tries the one-kernel form for a few steps: no k_sim launches in those)
            cs = sa.run(8)
            if cs["sim_launches"]:
                break
        sa.debug_set(4, 0)
        if cs["sim_launches"]:
            counted = dict(steps=cs["steps"], launches=cs["sim_launches"], bytes=cs["sim_bytes_counted"], evaluations=cs["evaluations"],
                           ms=cs["gpu_ms_sim"])
    # the same state in the library's default accept mode (auto: steps take every compatible improving neighbour and patch
    # them in at once, or rebuild when they are many), for the same number of steps: what a user of the defaults gets
    default_mode = None
    if rank == 0:
        sa.set_accept_mode("auto")
        sa.run(4)
        ba0 = sa.batch_counters()
        t1 = time.perf_counter()
        ds = sa.run(args.steps)
        t1 = time.perf_counter() - t1
        ba1 = sa.batch_counters()
        default_mode = {"evals_per_s": ds["evaluations"] / t1, "ms_per_step": t1 / max(1, ds["steps"]) * 1e3, "steps": ds["steps"],
                        "moves_per_step": ds["accepted"] / max(1, ds["steps"]), "bulk_steps": ds["bulk_steps"],
                        "steps_patched_in_place": ba1[0] - ba0[0], "steps_rebuilt": ds["bulk_steps"] - (ba1[0] - ba0[0]),
                        "moves_per_s": ds["accepted"] / t1,
                        "single_mode_moves_per_s": st["accepted"] / elapsed if elapsed > 0 else None,
                        "note": "accept mode auto on the state the timed steps left; `value` above is accept mode single (one move per step at most)"}
    evals, walked = st["evaluations"], st["packets_evaluated"]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, float(evals), float(walked)], dtype=torch.float64, device=coll_device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, evals, walked = float(tmax[0]), float(t[1]), float(t[2])

    if rank == 0:
        P = st["packets"]
        b_eval = n + 12 * P  # every input byte once + one 12-byte packet record per packet of the parse
        launches = max(1, st["neighbour_launches"])
        nbr_ms = st["gpu_ms_neighbours"] / launches  # all neighbour kernels of a step, HIP events on the library's streams
        evals_per_step = st["evaluations"] / max(1, st["steps"])
        pmc, pmc_path = load_pmc(cfg)
        sim_ms = st["gpu_ms_sim"] / st["sim_launches"] if st.get("sim_launches") else None  # HIP events on the streams k_sim runs on
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "k_sim (chain re-simulation), the dominant kernel: one launch per slice of the step's neighbours + one over the second "
                          "pass's list; per touched probability context a lane follows the base chain from the first change until the "
                          "perturbed probability re-joins the base trajectory",
                "avg_launch_ms": sim_ms, "launches_timed": st.get("sim_launches", 0),
                "definition": "achieved = the kernel's ALGORITHMIC bytes per launch (chain positions and events and change-list entries its loads ask "
                              "for, counted by the kernel itself in a few extra steps after the timed region) / its average launch duration measured "
                              "live with HIP events on its streams; traffic = HBM bytes per launch from rocprofv3 FETCH_SIZE + WRITE_SIZE (raw) of "
                              "the stored profile, null when that profile was taken of other kernel sources than the ones running",
                "algorithmic_equiv_gbs": evals_per_step * b_eval / (nbr_ms * 1e-3) / 1e9 if nbr_ms > 0 else None,
                "algorithmic_note": "SURVEY 8d's figure -- evaluations/step x (N + 12 P) / time of all neighbour kernels: what a streaming evaluator "
                                    "would move; the incremental kernels price only the changed window, so this is not a fraction of anything",
                "b_eval_bytes": b_eval, "packets_on_walk": P, "packets_walked_per_eval": walked / max(1.0, evals)}
        if counted and sim_ms:
            per_launch = counted["bytes"] / counted["launches"]
            roof["algorithmic_bytes_per_launch"] = per_launch
            # per evaluation: over the steps that ran the split form (k_sim launches per such step: one per slice of the step's
            # neighbours -- two above 1 MiB -- and one over the second pass's list; steps of a trial of the one-kernel form launch none)
            per_step = 3 if n > (1 << 20) else 2
            roof["algorithmic_bytes_per_evaluation"] = per_launch * per_step / max(1.0, counted["evaluations"] / max(1, counted["steps"]))
            roof["counted_over"] = (f"{counted['launches'] // per_step} of {counted['steps']} steps after the timed region "
                                    f"({counted['launches']} launches, kernel's own byte counters on)")
            roof["achieved"] = per_launch / (sim_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        nk = {"avg_ms_per_step_live": nbr_ms, "launches_timed": launches}  # all neighbour kernels of a step together
        if pmc:
            ks = pmc["kernels"]
            fresh = pmc.get("source_hash") == source_hash()
            dom = {k: v for k, v in ks.items() if any(k.startswith(d) for d in DOMINANT)}
            sim = [v for k, v in ks.items() if k.startswith(SIM)]
            roof["traffic_source"] = os.path.relpath(pmc_path, ROOT)
            roof["traffic_profile_matches_build"] = fresh
            if fresh and sim:
                roof["traffic"] = sum(v.get("hbm_bytes_raw_per_launch", 0.0) * v["launches_per_step"] for v in sim) / sum(v["launches_per_step"] for v in sim)
                roof["traffic_upper_gfx950_fetch_x2"] = sum(v.get("hbm_bytes_upper_per_launch", 0.0) * v["launches_per_step"] for v in sim) / sum(v["launches_per_step"] for v in sim)
                roof["profile_avg_launch_ms"] = sum(v["us_per_step"] for v in sim) / sum(v["launches_per_step"] for v in sim) / 1e3
                if sim_ms:
                    roof["hbm_frac_raw_counters"] = roof["traffic"] / (sim_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            elif not fresh:
                roof["traffic_note"] = "the stored profile was taken of other kernel sources (source_hash differs): no counter figure is claimed for this build"
            if fresh:
                raw = sum(v.get("hbm_bytes_raw_per_launch", 0.0) * v["launches_per_step"] for v in dom.values())
                upper = sum(v.get("hbm_bytes_upper_per_launch", 0.0) * v["launches_per_step"] for v in dom.values())
                if raw and nbr_ms > 0:
                    nk.update(hbm_bytes_raw_per_step=raw, hbm_bytes_upper_per_step=upper, frac_raw_counters=raw / (nbr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              frac_upper_gfx950_fetch_x2=upper / (nbr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              profile_us_per_step=sum(v["us_per_step"] for v in dom.values()))
                roof["limiter"] = {"kind": "latency / issue (dependent L2 + LDS accesses at 3-16 wavefronts per CU), not HBM",
                                   "per_kernel": {k: {c: v[c] for c in ("avg_us", "launches_per_step", "vgpr", "lds", "SQ_WAVES", "SQ_WAVE_CYCLES",
                                                                        "SQ_BUSY_CYCLES", "wait_any_frac", "issue_stall_frac", "SQ_INSTS_VALU",
                                                                        "SQ_INSTS_SALU", "hbm_gbs_raw") if c in v} for k, v in dom.items()}}
        else:
            roof["traffic_source"] = f"missing: {os.path.relpath(pmc_path, ROOT)} (tools/collect_roofline.sh {cfg})"
        roof["neighbour_kernels"] = nk
        lcpb = f"{props.get('lc', 0)}/{props.get('lp', 0)}/{props.get('pb', 0)}"
        out = {
            "metric": "SA neighbour-cost evals/s",
            "value": evals / elapsed,
            "unit": "evals/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{cfg}: {desc}, {n} B, {K} neighbours/step, top-K 20, lc/lp/pb={lcpb}"
                                   + (f", bucket scan capped at {props['max_bucket_scan']}" if props.get("max_bucket_scan") else "")
                                   + (f"; slab state: greedy seed ({greedy} candidates); timed steps in accept mode {accept}" if greedy else
                                      f"; slab state: after {prepare} search steps from the all-literal slab (accept mode auto); timed steps in accept mode {accept}"),
                       "chains": n_gpus, "parallelism": f"{n_gpus} independent chain(s), " + ("ALL ON ONE GPU (rehearsal of the N > 1 path, not a scaling figure)" if shared_gpu else "1 per GPU")
                                   + (", best-slab exchange timed separately" if n_gpus > 1 else "")},
            "roofline": roof,
            "final": {"current_cost": st["current_cost"], "best_cost": st["best_cost"],
                      "est_bytes_best": 18 + st["best_cost"] / 16384, "accepted": st["accepted"],
                      "bulk_steps_timed": st["bulk_steps"], "improving_neighbours": st["improving_neighbours"],
                      "dropped_neighbours": st["dropped_neighbours"],
                      "gpu_ms_apply_avg": st["gpu_ms_rebuild"] / launches,
                      "gpu_ms_total": st["gpu_ms_total"], "full_rebuilds": st["full_rebuilds"],
                      "fallback_neighbours": st["fallback_neighbours"],
                      "second_pass_neighbours": st["second_pass_neighbours"]},
        }
        if default_mode:
            out["default_mode"] = default_mode
        if exchange:
            out["exchange"] = exchange
        if prep:
            out["prepare"] = {"greedy_candidates": prep.get("greedy_candidates"), "steps": prep["steps"], "seconds": t_prep, "evaluations": prep["evaluations"], "bulk_steps": prep["bulk_steps"], "bulk_rollbacks": prep["bulk_rollbacks"],
                              "moves_accepted": prep["accepted"], "est_bytes_best": 18 + prep["best_cost"] / 16384,
                              "evals_per_s": prep["evaluations"] / t_prep if t_prep > 0 else None}
        if prep and t_prep > 0 and not greedy:
            # what a user of the default (AUTO) mode gets over the whole run so far: the set-up phase is a search in AUTO mode
            # from the all-literal slab (mostly bulk steps, each a parallel rebuild), the timed steps are single steps
            out["whole_run"] = {"evals_per_s": (prep["evaluations"] + st["evaluations"]) / (t_prep + elapsed),
                                "default_auto_mode_evals_per_s": prep["evaluations"] / t_prep,
                                "default_auto_mode_ms_per_step": t_prep / max(1, prep["steps"]) * 1e3,
                                "steps": prep["steps"] + st["steps"], "seconds": t_prep + elapsed,
                                "note": "set-up search (accept mode auto, from the all-literal slab) + timed steps; `value` is the steady state "
                                        "of a long run (single steps on the evolved slab)"}
        gates = {}
        if n <= (16 << 20):
            # correctness gates reported with the number, after the timed region: the best slab's cost
            # re-derived by the device's independent full walk, and its stream through liblzma
            import lzma
            best, best_cost = sa.best()
            lc = {k: props[k] for k in ("lc", "lp", "pb") if k in props}
            stream = binding.emit_stream(data, best, **lc)
            try:
                ok = lzma.decompress(stream, format=lzma.FORMAT_ALONE) == bytes(data)
            except lzma.LZMAError:
                ok = False
            gates.update(best_cost_equals_full_walk=sa.cost_slab(best, want_cum=False)["total"] == best_cost,
                         lzma_roundtrip=ok, stream_bytes=len(stream))
        sa.close()
        if n_gpus == 1 and not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import build_oracle
            build_oracle.build_oracle()
            out["cpu_baseline"] = cpu_baseline(data, args.cpu_seconds)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(data, min(args.cpu_seconds, 8.0))
            out["vs_cpu_reference_1_thread"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        if n_gpus == 1 and not args.no_secondary:
            # the rate on the young slab (what round 1 reported) and the size comparison with the reference path
            y = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=n, timing=True, accept="single", **props)
            y.run(5)
            t1 = time.perf_counter()
            ys = y.run(20)
            t1 = time.perf_counter() - t1
            y.close()
            out["young_slab"] = {"evals_per_s": ys["evaluations"] / t1, "ms_per_step": t1 / 20 * 1e3,
                                 "note": "single steps 5..25 from the all-literal slab (99 % literals: the state round 1's line was taken in)"}
            sg = size_gates(binding, data, K, props, out.get("cpu_baseline"), cfg)
            if sg:
                gates["size_vs_reference"] = sg
            if cfg != "c2":
                c2, _ = corpus.config_input("c2")
                sg2 = size_gates(binding, c2, DEFAULT_K["c2"], {}, None, "c2")
                if sg2:
                    gates["size_vs_reference_c2"] = sg2
                # the same with every step a bulk step: AUTO goes back to single steps (one accept per 4 096 evaluations) as soon
                # as that is faster per second, which costs progress per evaluation -- this is the per-evaluation best of the engine
                sg3 = size_gates(binding, c2, DEFAULT_K["c2"], {}, None, "c2", accept="bulk")
                if sg3:
                    gates["size_vs_reference_c2_bulk"] = sg3
                sg4 = seed_spread_gate(binding, c2, DEFAULT_K["c2"], {}, "c2")
                if sg4:
                    gates["size_vs_reference_c2_means_over_seeds"] = sg4
        if gates:
            out["gates"] = gates
        print(json.dumps(out), file=json_out, flush=True)
    else:
        sa.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
