"""ctypes bindings for the CPU oracle (oracle/_build/liboracle.so) and, when present, the
compiled reference (oracle/_ref/libmegalania_ref.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("MGL_ORACLE_SO") or os.path.join(ROOT, "oracle", "_build", "liboracle.so")  # the override: tests/test_sanitizers.py
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmegalania_ref.so")

# lzma_packet.h:13-17 -- 12-byte record: type u8 @0, dist u32 @4, len u16 @8
PACKET = np.dtype([("type", "u1"), ("dist", "u4"), ("len", "u2")], align=True)
assert PACKET.itemsize == 12
DIFF = np.dtype([("position", "u4"), ("old", PACKET), ("new", PACKET)], align=True)
assert DIFF.itemsize == 28

LITERAL, MATCH, SHORT_REP, LONG_REP = 1, 2, 3, 4

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
szp = C.POINTER(C.c_size_t)


def ptr(a, t=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def literal_slab(n: int) -> np.ndarray:
    s = np.zeros(n, dtype=PACKET)
    s["type"] = LITERAL
    s["len"] = 1
    return s


def slab_from_list(n: int, packets) -> np.ndarray:
    """packets: list of (type, dist, len) laid down consecutively from position 0."""
    s = literal_slab(n)
    pos = 0
    for t, d, l in packets:
        s[pos] = (t, d, l)
        pos += l
    assert pos <= n
    return s


def walk(slab: np.ndarray):
    pos, out = 0, []
    n = len(slab)
    lens = slab["len"]
    while pos < n:
        out.append(pos)
        pos += int(lens[pos])
    return out


def _build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
        os.path.join(ROOT, "oracle", "mgl_oracle.c")
    ):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/liboracle.so"],
                              stdout=subprocess.DEVNULL)


class Oracle:
    """One orc_ctx over a byte buffer."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            _build_oracle()
            L = C.CDLL(ORACLE_SO)
            L.orc_new.restype = C.c_void_p
            L.orc_new.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32]
            L.orc_free.argtypes = [C.c_void_p]
            L.orc_num_probs.restype = C.c_size_t
            L.orc_num_probs.argtypes = [C.c_void_p]
            L.orc_cost_table.restype = u16p
            L.orc_cost_slab.restype = C.c_uint64
            L.orc_cost_slab.argtypes = [C.c_void_p] + [C.c_void_p] * 6
            L.orc_substrings.restype = C.c_size_t
            L.orc_substrings.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
            L.orc_top_k.restype = C.c_size_t
            L.orc_top_k.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
            L.orc_srand.argtypes = [C.c_uint]
            L.orc_sa_iters.restype = C.c_int
            L.orc_sa_iters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.orc_draw.restype = C.c_uint32
            L.orc_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
            L.orc_neighbour.restype = C.c_int
            L.orc_neighbour.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
            L.orc_sa_batched.restype = C.c_int
            L.orc_sa_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_uint32, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.orc_bulk_rollbacks.restype = C.c_uint64
            L.orc_bulk_rollbacks.argtypes = []
            L.orc_bulk_overlaps.restype = C.c_uint64
            L.orc_bulk_overlaps.argtypes = []
            L.orc_neighbour_ex.restype = C.c_int
            L.orc_neighbour_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
            L.orc_set_max_bucket_scan.argtypes = [C.c_void_p, C.c_uint32]
            L.orc_set_strata.argtypes = [C.c_void_p, C.c_uint32]
            L.orc_set_temperature.argtypes = [C.c_void_p, C.c_uint64]
            L.orc_emit.restype = C.c_size_t
            L.orc_emit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
            cls._lib = L
        return cls._lib

    def __init__(self, data: bytes, lc=0, lp=0, pb=0, dict_limit=0, max_bucket_scan=0, position_targets=False):
        """position_targets: the batched mode draws targets as positions (mgl_sa_config.flags & MGL_F_POSITION_TARGETS) instead of
        the device's default, stratified by packet ordinal over the K neighbours of a step (then neighbour() needs K)."""
        self.stratified = not position_targets
        self.L = self.lib()
        self.data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self.n = len(self.data)
        self.h = self.L.orc_new(ptr(self.data), self.n, lc, lp, pb, dict_limit)
        self.nprobs = self.L.orc_num_probs(self.h)
        if max_bucket_scan:
            self.L.orc_set_max_bucket_scan(self.h, max_bucket_scan)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    @staticmethod
    def cost_table() -> np.ndarray:
        p = Oracle.lib().orc_cost_table()
        return np.ctypeslib.as_array(p, shape=(2048,)).copy()

    def cost_slab(self, slab, want_probs=False):
        cum = np.zeros(self.n, dtype=np.uint64)
        npk = C.c_size_t(0)
        probs = np.zeros(self.nprobs, dtype=np.uint16)
        cs = C.c_uint8(0)
        dists = np.zeros(4, dtype=np.uint32)
        total = self.L.orc_cost_slab(self.h, ptr(slab), ptr(cum), C.addressof(npk), ptr(probs),
                                     C.addressof(cs), ptr(dists))
        r = dict(total=int(total), cum=cum[: npk.value].copy(), ctx_state=cs.value, dists=dists)
        if want_probs:
            r["probs"] = probs
        return r

    def trace_events(self, slab):
        """(ctx, bit, prob_before, pos) per coded bit in coding order + (pos, ctx_state, dists) per packet."""
        cap = 9 * self.n + 64
        ctx = np.zeros(cap, dtype=np.uint32)
        bit = np.zeros(cap, dtype=np.uint8)
        prob = np.zeros(cap, dtype=np.uint16)
        pos = np.zeros(cap, dtype=np.uint32)
        pk_pos = np.zeros(self.n, dtype=np.uint32)
        pk_state = np.zeros(5 * self.n, dtype=np.uint32)
        npk = C.c_size_t(0)
        fn = self.L.orc_trace_events
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        ne = fn(self.h, ptr(slab), ptr(ctx), ptr(bit), ptr(prob), ptr(pos), cap, ptr(pk_pos), ptr(pk_state), self.n,
                C.addressof(npk))
        assert ne <= cap
        return dict(ctx=ctx[:ne], bit=bit[:ne], prob=prob[:ne], pos=pos[:ne], pk_pos=pk_pos[: npk.value],
                    pk_state=pk_state[: 5 * npk.value].reshape(-1, 5))

    def substrings(self, pos, max_len=273, cap=1 << 20):
        offs = np.zeros(cap, dtype=np.uint32)
        lens = np.zeros(cap, dtype=np.uint32)
        cnt = self.L.orc_substrings(self.h, pos, max_len, ptr(offs), ptr(lens), cap)
        assert cnt <= cap
        return offs[:cnt].copy(), lens[:cnt].copy()

    def top_k(self, slab, position, mode=0, k=20):
        out = np.zeros(64, dtype=PACKET)
        costs = np.zeros(64, dtype=np.uint64)
        cnt = self.L.orc_top_k(self.h, ptr(slab), position, mode, k, ptr(out), ptr(costs))
        assert cnt != (1 << 64) - 1, "position not on the walk"
        return out[:cnt].copy(), costs[:cnt].copy()

    def sa_iters(self, slab, best, cur, best_cost, step, num_iters, i_begin, i_end):
        trace = np.zeros(2 * max(1, i_end - i_begin), dtype=np.uint64)
        cur_c, best_c, undo = C.c_uint64(cur), C.c_uint64(best_cost), C.c_uint64(0)
        t = self.L.orc_sa_iters(self.h, ptr(slab), ptr(best), C.addressof(cur_c), C.addressof(best_c), step,
                                num_iters, i_begin, i_end, ptr(trace), C.addressof(undo))
        return dict(cur=cur_c.value, best=best_c.value, trace=trace[: 2 * t].reshape(-1, 2).copy(), undo=undo.value)

    def _strata(self, K):
        assert not self.stratified or K, "stratified targets: pass K (the device's neighbours_per_step)"
        self.L.orc_set_strata(self.h, K if self.stratified else 0)

    def neighbour(self, slab, seed, step, j, keep=False, cap=4096, K=None):
        self._strata(K)
        cost = C.c_uint64(0)
        nd = C.c_size_t(0)
        diffs = np.zeros(cap, dtype=DIFF)
        ok = self.L.orc_neighbour(self.h, ptr(slab), seed, step, j, int(keep), C.addressof(cost), ptr(diffs),
                                  C.addressof(nd), cap)
        assert nd.value <= cap
        return bool(ok), cost.value, diffs[: nd.value].copy()

    def neighbour_ex(self, slab, seed, step, j, keep=False, cap=4096, K=None):
        self._strata(K)
        """status (1 ok / 0 no candidate / -1 dropped by the journal capacity), cost, diffs, (target, end, soft end, dep)"""
        cost = C.c_uint64(0)
        nd = C.c_size_t(0)
        diffs = np.zeros(cap, dtype=DIFF)
        win = np.zeros(4, dtype=np.uint32)
        st = self.L.orc_neighbour_ex(self.h, ptr(slab), seed, step, j, int(keep), C.addressof(cost), ptr(diffs),
                                     C.addressof(nd), cap, ptr(win))
        return st, cost.value, diffs[: min(nd.value, cap)].copy(), tuple(int(x) for x in win)

    def set_temperature(self, temperature: int):
        self.L.orc_set_temperature(self.h, temperature)

    def sa_batched(self, slab, best, cur, best_cost, seed, K, phase, iters_per_epoch, step_begin, step_end,
                   iter0=0, modes=None):
        """modes: None = every step takes the single best acceptable neighbour; else one byte per step
        (0 single, 1 bulk).  trace columns: smallest acceptable cost, accepted, acceptable, current cost."""
        nsteps = max(1, step_end - step_begin)
        trace = np.zeros(4 * nsteps, dtype=np.uint64)
        m = None if modes is None else np.ascontiguousarray(modes, dtype=np.uint8)
        assert m is None or len(m) >= step_end - step_begin
        cur_c, best_c, valid, dropped = C.c_uint64(cur), C.c_uint64(best_cost), C.c_uint64(0), C.c_uint64(0)
        self._strata(K)
        self.L.orc_sa_batched(self.h, ptr(slab), ptr(best), C.addressof(cur_c), C.addressof(best_c), seed, K, phase,
                              iters_per_epoch, iter0, step_begin, step_end, ptr(m), ptr(trace), C.addressof(valid),
                              C.addressof(dropped))
        return dict(cur=cur_c.value, best=best_c.value, trace=trace.reshape(-1, 4).copy(), valid=valid.value,
                    dropped=dropped.value)

    def bulk_rollbacks(self) -> int:
        """bulk steps the oracle took back since the library was loaded (a process-wide counter)"""
        return int(self.L.orc_bulk_rollbacks())

    def bulk_overlaps(self) -> int:
        """slab entries that two taken journals of one bulk step both wrote, since the library was loaded"""
        return int(self.L.orc_bulk_overlaps())

    def emit(self, slab) -> bytes:
        cap = 2 * self.n + 1024
        out = np.zeros(cap, dtype=np.uint8)
        ln = self.L.orc_emit(self.h, ptr(slab), ptr(out), cap)
        assert ln <= cap
        return out[:ln].tobytes()


class Ref:
    """The compiled reference (only in the build container; the .so travels to the GPU box)."""

    _lib = None

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(REF_SO)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(REF_SO)
            L.ref_sizeof_packet.restype = C.c_size_t
            L.ref_sizeof_state.restype = C.c_size_t
            L.ref_num_probs.restype = C.c_size_t
            L.ref_cost_slab.restype = C.c_uint64
            L.ref_cost_slab.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
            L.ref_top_k.restype = C.c_size_t
            L.ref_top_k.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
            L.ref_substrings.restype = C.c_size_t
            L.ref_substrings.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_size_t]
            L.ref_srand.argtypes = [C.c_uint]
            L.ref_sa_iters.restype = C.c_int
            L.ref_sa_iters.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.ref_emit.restype = C.c_size_t
            L.ref_emit.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
            cls._lib = L
        return cls._lib

    def __init__(self, data: bytes):
        self.L = self.lib()
        self.data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self.n = len(self.data)
        self.nprobs = self.L.ref_num_probs()

    def cost_slab(self, slab, want_probs=False):
        cum = np.zeros(self.n, dtype=np.uint64)
        npk = C.c_size_t(0)
        probs = np.zeros(self.nprobs, dtype=np.uint16)
        cs = C.c_uint8(0)
        dists = np.zeros(4, dtype=np.uint32)
        total = self.L.ref_cost_slab(ptr(self.data), self.n, ptr(slab), ptr(cum), C.addressof(npk), ptr(probs),
                                     C.addressof(cs), ptr(dists))
        r = dict(total=int(total), cum=cum[: npk.value].copy(), ctx_state=cs.value, dists=dists)
        if want_probs:
            r["probs"] = probs
        return r

    def substrings(self, pos, max_len=273, cap=1 << 20):
        offs = np.zeros(cap, dtype=np.uint32)
        lens = np.zeros(cap, dtype=np.uint32)
        cnt = self.L.ref_substrings(ptr(self.data), self.n, pos, max_len, ptr(offs), ptr(lens), cap)
        assert cnt <= cap
        return offs[:cnt].copy(), lens[:cnt].copy()

    def top_k(self, slab, position, k=20):
        out = np.zeros(64, dtype=PACKET)
        costs = np.zeros(64, dtype=np.uint64)
        s = slab.copy()
        cnt = self.L.ref_top_k(ptr(self.data), self.n, ptr(s), position, k, ptr(out), ptr(costs))
        assert cnt != (1 << 64) - 1
        return out[:cnt].copy(), costs[:cnt].copy()

    def sa_iters(self, slab, best, cur, best_cost, step, num_iters, i_begin, i_end):
        trace = np.zeros(2 * max(1, i_end - i_begin), dtype=np.uint64)
        cur_c, best_c, undo = C.c_uint64(cur), C.c_uint64(best_cost), C.c_uint64(0)
        t = self.L.ref_sa_iters(ptr(self.data), self.n, ptr(slab), ptr(best), C.addressof(cur_c), C.addressof(best_c),
                                step, num_iters, i_begin, i_end, ptr(trace), C.addressof(undo))
        return dict(cur=cur_c.value, best=best_c.value, trace=trace[: 2 * t].reshape(-1, 2).copy(), undo=undo.value)

    def emit(self, slab) -> bytes:
        cap = 2 * self.n + 1024
        out = np.zeros(cap, dtype=np.uint8)
        ln = self.L.ref_emit(ptr(self.data), self.n, ptr(slab), ptr(out), cap)
        assert ln <= cap
        return out[:ln].tobytes()
