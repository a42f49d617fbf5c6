#!/usr/bin/env python3
"""Generate tests/golden/*.json from the COMPILED REFERENCE (oracle/_ref).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle`).  Every expected value in the fixtures is produced by the reference's own
code (through oracle/ref_harness.c); the oracle is used only to *construct interesting input
slabs* (any valid slab is a legitimate input), never to produce an expected value.

Fixtures are data: inputs (bytes as hex, or a generator name + args), packet lists, and the
reference's outputs (costs, states, top-K lists, callback sequences, stream bytes).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from _libs import LITERAL, LONG_REP, MATCH, SHORT_REP, Oracle, Ref, literal_slab, slab_from_list, walk  # noqa: E402
from megalania_amd import corpus  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_slab(slab) -> str:
    """field-wise hash (the 12-byte record has 5 padding bytes whose content is arbitrary)"""
    return sha(np.concatenate([slab["type"].astype(np.uint32), slab["dist"], slab["len"].astype(np.uint32)]))


def rand_bytes(n, seed):
    return (corpus._stream(seed, 11, 0, n) & np.uint64(0xFF)).astype(np.uint8).tobytes()


def rep_data():
    """Segments copied around so that rep0..rep3 all become usable."""
    segs = [rand_bytes(24 + 7 * i, 0x100 + i) for i in range(5)]
    order = [0, 1, 2, 3, 4, 0, 1, 2, 3, 0, 2, 1, 3, 0, 0, 4, 1, 3, 2, 0, 1, 1, 2, 3, 4, 0]
    out = b""
    for k, i in enumerate(order):
        out += segs[i] + bytes([65 + k % 7])
    return out + b"zz" + segs[2][:9] + b"q" + segs[2][:9] + b"q" + segs[4]


INPUTS = {
    "hello": dict(hex=b"hello hello".hex()),
    "abab": dict(hex=b"abababababababab".hex()),
    "aaab": dict(hex=b"aaaaaaaabaaaaaaa".hex()),
    "lorem512": dict(gen="lorem", n=512),
    "lorem4k": dict(gen="lorem", n=4096),
    "enwik3k": dict(gen="enwik_like", n=3000, seed=0xE5),
    "rand2k": dict(gen="rand_bytes", n=2048, seed=0x77),
    "reps": dict(hex=rep_data().hex()),
    "zeros600": dict(hex=(b"\0" * 600).hex()),
    "far": dict(gen="far", n=1_200_000, seed=0xFA),
}


def materialise(spec) -> bytes:
    if "hex" in spec:
        return bytes.fromhex(spec["hex"])
    if spec["gen"] == "lorem":
        return corpus.lorem(spec["n"])
    if spec["gen"] == "enwik_like":
        return corpus.enwik_like(spec["n"], spec["seed"])
    if spec["gen"] == "rand_bytes":
        return rand_bytes(spec["n"], spec["seed"])
    if spec["gen"] == "far":
        return far_data(spec["n"], spec["seed"])
    raise KeyError(spec)


def far_data(n, seed):
    """512 random bytes that recur 70 000 bytes in and again at the very end: match distances
    in the 2^16 and 2^20 bands (pos-slot >= 32, 11..15 direct bits)."""
    key = rand_bytes(512, seed)
    body = bytearray(corpus.enwik_like(n - 1024, seed ^ 0x33))
    body[70_000:70_256] = key[:256]
    return key + bytes(body) + key


def pk_list(slab, positions=None):
    """Packets along the walk; runs of literals are run-length coded as ["L", count]."""
    positions = walk(slab) if positions is None else positions
    out = []
    for p in positions:
        t, d, l = int(slab[p]["type"]), int(slab[p]["dist"]), int(slab[p]["len"])
        if t == LITERAL and out and out[-1][0] == "L":
            out[-1][1] += 1
        elif t == LITERAL:
            out.append(["L", 1])
        else:
            out.append([t, d, l])
    return out


def scripted_parse(data: bytes, prefer):
    """A valid parse that exercises rare packet kinds: at each position take the first
    candidate kind in a rotating preference list.  Construction only (oracle enumerator)."""
    o = Oracle(data)
    n = len(data)
    slab = literal_slab(n)
    pos, turn = 0, 0
    while pos < n:
        cands, _ = o.top_k(slab, pos, mode=1, k=64)
        pick = None
        order = prefer[turn % len(prefer):] + prefer[: turn % len(prefer)]
        for want in order:
            sel = [c for c in cands if (c["type"], c["dist"] if c["type"] == LONG_REP else 0) == want
                   or (want[0] == MATCH and c["type"] == MATCH and want[1] == 0)]
            if want[0] == LITERAL:
                pick = (LITERAL, 0, 1)
                break
            if sel:
                best = max(sel, key=lambda c: int(c["len"]))
                pick = (int(best["type"]), int(best["dist"]), int(best["len"]))
                break
        if pick is None:
            pick = (LITERAL, 0, 1)
        slab[pos] = pick
        pos += pick[2]
        turn += 1
    return slab


def main():
    assert Ref.available(), "build oracle/_ref first: make -C oracle"
    os.makedirs(OUT, exist_ok=True)
    L = Ref.lib()
    meta = dict(
        made_by="tools/make_golden.py from oracle/_ref/libmegalania_ref.so (compiled /root/reference/src)",
        sizeof_packet=L.ref_sizeof_packet(), sizeof_state=L.ref_sizeof_state(), num_probs=L.ref_num_probs(),
    )

    # ---- 1. cost walks ---------------------------------------------------------------
    walks = []

    def add_walk(name, inp, slab, note):
        data = materialise(INPUTS[inp])
        r = Ref(data)
        res = r.cost_slab(slab, want_probs=True)
        stream = r.emit(slab)
        entry = dict(name=name, input=inp, note=note, packets=pk_list(slab), total=res["total"],
                     npackets=len(res["cum"]), ctx_state=res["ctx_state"], dists=[int(x) for x in res["dists"]],
                     probs_sha256=sha(res["probs"]), cum_sha256=sha(res["cum"]), stream_sha256=sha(np.frombuffer(stream, dtype=np.uint8)),
                     stream_len=len(stream))
        if len(res["cum"]) <= 64:
            entry["cum"] = [int(x) for x in res["cum"]]
            entry["stream_hex"] = stream.hex()
        else:
            entry["cum_head"] = [int(x) for x in res["cum"][:8]]
            entry["cum_tail"] = [int(x) for x in res["cum"][-8:]]
        walks.append(entry)

    add_walk("hello_literals", "hello", literal_slab(11), "SURVEY 8c known answer")
    add_walk("hello_match", "hello", slab_from_list(11, [(1, 0, 1)] * 6 + [(2, 5, 5)]), "SURVEY 8c known answer")
    add_walk("abab", "abab", slab_from_list(16, [(1, 0, 1), (1, 0, 1), (2, 1, 6), (4, 0, 8)]), "SURVEY 8c known answer")
    add_walk("aaab", "aaab", slab_from_list(16, [(1, 0, 1), (2, 0, 7), (1, 0, 1), (1, 0, 1), (3, 0, 1), (4, 0, 5)]),
             "SURVEY 8c known answer")
    for inp in ("lorem512", "lorem4k", "enwik3k", "rand2k", "reps", "zeros600"):
        add_walk(inp + "_literals", inp, literal_slab(len(materialise(INPUTS[inp]))), "all-literal slab")
    prefs = {
        "longest": [(MATCH, 0), (LITERAL, 0)],
        "reps": [(LONG_REP, 3), (LONG_REP, 2), (LONG_REP, 1), (LONG_REP, 0), (SHORT_REP, 0), (MATCH, 0), (LITERAL, 0)],
        "mixed": [(SHORT_REP, 0), (MATCH, 0), (LITERAL, 0), (LONG_REP, 1), (LITERAL, 0), (LONG_REP, 0), (MATCH, 0)],
    }
    far = materialise(INPUTS["far"])
    fs = literal_slab(len(far))
    fs[70_512] = (MATCH, 70_511, 200)            # key[0:200] seen 70 512 bytes earlier
    fs[70_712] = (MATCH, 70_511, 56)             # same distance again, as a plain MATCH
    fs[len(far) - 512] = (MATCH, len(far) - 512 - 1, 273)
    fs[len(far) - 239] = (LONG_REP, 0, 239)
    add_walk("far_matches", "far", fs, "distances 70 511 and 1 199 487: 11 and 15 direct bits")
    for inp in ("lorem4k", "enwik3k", "reps", "zeros600", "rand2k"):
        for pname, pref in prefs.items():
            slab = scripted_parse(materialise(INPUTS[inp]), pref)
            add_walk(f"{inp}_{pname}", inp, slab, f"scripted parse, preference '{pname}'")

    # ---- 2. SA trajectories (reference semantics, glibc rand) ---------------------------
    sa = []
    evolved = {}
    for inp, iters, step in (("lorem512", 400, 0), ("lorem4k", 600, 0), ("enwik3k", 500, 0), ("reps", 400, 1)):
        data = materialise(INPUTS[inp])
        n = len(data)
        r = Ref(data)
        slab, best = literal_slab(n), literal_slab(n)
        L.ref_srand(1673551)
        res = r.sa_iters(slab, best, 0, 0, step, n, 0, iters)
        evolved[inp] = slab.copy()
        sa.append(dict(input=inp, seed=1673551, step=step, num_iters=n, iters=iters, cur=res["cur"], best=res["best"],
                       undo_total=res["undo"], trace_cost=[int(x) for x in res["trace"][:, 0]],
                       trace_accept=[int(x) for x in res["trace"][:, 1]], slab_sha256=sha_slab(slab), best_sha256=sha_slab(best),
                       final_packets=pk_list(slab)))
        add_walk(inp + "_evolved", inp, slab, f"slab after {iters} reference SA iterations")

    # ---- 3. top-K lists -----------------------------------------------------------------
    topk = []
    for inp, slab in (("lorem4k", evolved["lorem4k"]), ("enwik3k", evolved["enwik3k"]), ("reps", evolved["reps"]),
                      ("lorem512", literal_slab(512)), ("zeros600", literal_slab(600)), ("hello", literal_slab(11))):
        data = materialise(INPUTS[inp])
        r = Ref(data)
        w = walk(slab)
        picks = sorted(set([w[0], w[1], w[len(w) // 3], w[len(w) // 2], w[(2 * len(w)) // 3], w[-2], w[-1]]))
        for pos in picks:
            pk, costs = r.top_k(slab, pos)
            topk.append(dict(input=inp, slab="evolved" if inp in evolved and slab is evolved.get(inp) else "literal",
                             position=int(pos), packets=pk_list(pk, range(len(pk))), costs=[int(c) for c in costs]))
    # ---- 4. match-index callbacks -------------------------------------------------------
    subs = []
    for inp, maxlen in (("hello", 273), ("hello", 3), ("abab", 273), ("lorem512", 273), ("zeros600", 273)):
        data = materialise(INPUTS[inp])
        r = Ref(data)
        per_pos = []
        for pos in range(len(data)) if len(data) <= 16 else (0, 1, 100, 446, 447, 500, len(data) - 2, len(data) - 1):
            offs, lens = r.substrings(pos, maxlen)
            ent = dict(pos=pos, count=len(offs))
            if len(offs) <= 40:
                ent["offs"] = [int(x) for x in offs]
                ent["lens"] = [int(x) for x in lens]
            else:
                ent["sha256"] = sha(np.stack([offs, lens]))
            per_pos.append(ent)
        subs.append(dict(input=inp, max_len=maxlen, queries=per_pos))

    doc = dict(meta=meta, inputs=INPUTS, walks=walks, sa=sa, topk=topk, substrings=subs)
    # evolved slabs are inputs of the top-K fixtures; only walked packets matter there
    doc["evolved_walks"] = {k: pk_list(v) for k, v in evolved.items()}
    with open(os.path.join(OUT, "reference_vectors.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", os.path.join(OUT, "reference_vectors.json"), os.path.getsize(os.path.join(OUT, "reference_vectors.json")), "bytes")
    print(len(walks), "walks,", len(sa), "SA runs,", len(topk), "top-K lists")


if __name__ == "__main__":
    main()
