"""ctypes binding of the C ABI (include/megalania_hip.h) and of the C host library
(megalania_amd/host/mgl_host.h).  Thin: argument marshalling and error translation only.

There is deliberately no CPU fallback: if libmegalania_hip.so is missing, cannot be loaded,
or finds no GPU, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(HERE, "_build", "libmegalania_hip.so")
HOST_SO = os.environ.get("MGL_HOST_SO") or os.path.join(HERE, "_build", "libmegalania_host.so")  # the override: a sanitizer build (tests/test_sanitizers.py)

# lzma_packet.h:13-17 layout
PACKET = np.dtype([("type", "u1"), ("dist", "u4"), ("len", "u2")], align=True)
DIFF = np.dtype([("position", "u4"), ("old", PACKET), ("new", PACKET)], align=True)
assert PACKET.itemsize == 12 and DIFF.itemsize == 28

LITERAL, MATCH, SHORT_REP, LONG_REP = 1, 2, 3, 4
INVALID_COST = (1 << 64) - 1
F_TIMING = 1
F_FULLWALK = 2
F_PROFILE = 4
F_NO_SNAPSHOTS = 8
F_SERIAL_BUILD = 16
F_POSITION_TARGETS = 32
ACCEPT_AUTO, ACCEPT_SINGLE, ACCEPT_BULK = 0, 1, 2

HIP_SYMBOLS = [
    "mgl_version", "mgl_last_error", "mgl_device_count", "mgl_sa_create", "mgl_sa_destroy", "mgl_sa_begin_epoch",
    "mgl_sa_set_slab", "mgl_sa_seed_greedy", "mgl_sa_set_temperature", "mgl_sa_set_accept_mode", "mgl_sa_step_modes", "mgl_sa_set_best", "mgl_sa_run", "mgl_sa_current", "mgl_sa_best", "mgl_cost_slab", "mgl_final_state", "mgl_top_k",
    "mgl_substrings", "mgl_neighbours", "mgl_rng_draw_at", "mgl_debug_dump", "mgl_debug_set",
    "mgl_comm_unique_id", "mgl_comm_init", "mgl_comm_init_shm", "mgl_comm_min_u64", "mgl_comm_destroy", "mgl_comm_rank", "mgl_comm_world", "mgl_sa_exchange_best",
    "mgl_sa_best_packed", "mgl_sa_adopt_best_packed",
]
HOST_SYMBOLS = [
    "mgl_lzma_state_init", "mgl_lzma_state_free", "mgl_lzma_encode_packet", "mgl_lzma_encode_header",
    "mgl_range_encoder_new", "mgl_range_encoder_free", "mgl_perplexity_encoder_new", "mgl_file_output_new",
    "mgl_memory_output_new", "mgl_emit_stream",
]


class MglError(RuntimeError):
    pass


class Properties(C.Structure):
    _fields_ = [("lc", C.c_uint8), ("lp", C.c_uint8), ("pb", C.c_uint8)]


class Config(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("neighbours_per_step", C.c_uint32), ("top_k", C.c_uint32),
                ("dict_limit", C.c_uint32), ("max_bucket_scan", C.c_uint32), ("iters_per_epoch", C.c_uint64),
                ("device", C.c_int32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("evaluations", C.c_uint64), ("failed", C.c_uint64), ("accepted", C.c_uint64),
                ("improved", C.c_uint64), ("current_cost", C.c_uint64), ("best_cost", C.c_uint64),
                ("packets", C.c_uint64), ("packets_evaluated", C.c_uint64), ("gpu_ms_total", C.c_double),
                ("gpu_ms_neighbours", C.c_double), ("gpu_ms_rebuild", C.c_double), ("neighbour_launches", C.c_uint64),
                ("full_rebuilds", C.c_uint64), ("fallback_neighbours", C.c_uint64), ("second_pass_neighbours", C.c_uint64),
                ("bulk_steps", C.c_uint64), ("dropped_neighbours", C.c_uint64), ("improving_neighbours", C.c_uint64),
                ("bulk_rollbacks", C.c_uint64), ("bulk_double_writes", C.c_uint64),
                ("gpu_ms_sim", C.c_double), ("sim_launches", C.c_uint64), ("sim_bytes_counted", C.c_uint64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class MemorySink(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("cap", C.c_size_t), ("len", C.c_size_t)]


class OutputInterface(C.Structure):
    _fields_ = [("write", C.c_void_p), ("private_data", C.c_void_p)]


_hip = None
_host = None


def hip_lib():
    """Load libmegalania_hip.so.  Raises if it was not built -- no fallback."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise MglError(f"{HIP_SO} is missing: run `python -m megalania_amd.build` (hipcc, gfx950). "
                           "There is no CPU implementation of the search path.")
        # MGL_HIP_SO: another build of the same library (tools/ab_builds.sh compares two builds on one state)
        L = C.CDLL(os.environ.get("MGL_HIP_SO") or HIP_SO)
        L.mgl_version.restype = C.c_char_p
        L.mgl_last_error.restype = C.c_char_p
        L.mgl_device_count.restype = C.c_int
        L.mgl_sa_create.restype = C.c_void_p
        L.mgl_sa_create.argtypes = [C.c_void_p, C.c_size_t, Properties, C.POINTER(Config)]
        L.mgl_sa_destroy.argtypes = [C.c_void_p]
        L.mgl_sa_begin_epoch.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        L.mgl_sa_set_slab.argtypes = [C.c_void_p, C.c_void_p]
        L.mgl_sa_seed_greedy.argtypes = [C.c_void_p, C.c_uint32]
        L.mgl_sa_set_temperature.argtypes = [C.c_void_p, C.c_uint64]
        L.mgl_sa_set_best.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mgl_sa_set_accept_mode.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.mgl_sa_step_modes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mgl_sa_run.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Stats)]
        L.mgl_sa_current.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.mgl_sa_best.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.mgl_cost_slab.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_size_t)]
        L.mgl_final_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.c_void_p]
        L.mgl_top_k.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        L.mgl_substrings.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
        L.mgl_neighbours.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.mgl_debug_dump.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mgl_debug_set.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.mgl_comm_unique_id.argtypes = [C.c_void_p]
        L.mgl_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mgl_comm_init_shm.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.mgl_comm_min_u64.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.mgl_comm_destroy.argtypes = [C.c_void_p]
        L.mgl_comm_rank.argtypes = [C.c_void_p]
        L.mgl_comm_world.argtypes = [C.c_void_p]
        L.mgl_sa_exchange_best.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.mgl_sa_best_packed.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.mgl_sa_adopt_best_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mgl_rng_draw_at.restype = C.c_uint32
        L.mgl_rng_draw_at.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise MglError(f"{HOST_SO} is missing: run `python -m megalania_amd.build`")
        L = C.CDLL(HOST_SO)
        L.mgl_emit_stream.restype = C.c_bool
        L.mgl_emit_stream.argtypes = [C.c_void_p, C.c_size_t, Properties, C.c_void_p, C.POINTER(OutputInterface)]
        L.mgl_memory_output_new.argtypes = [C.POINTER(OutputInterface), C.POINTER(MemorySink)]
        _host = L
    return _host


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def literal_slab(n: int) -> np.ndarray:
    s = np.zeros(n, dtype=PACKET)
    s["type"] = LITERAL
    s["len"] = 1
    return s


def emit_stream(data: bytes, slab: np.ndarray, lc=0, lp=0, pb=0) -> bytes:
    """Host emission (C): header + range coder over the slab's walk -> .lzma bytes."""
    L = host_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    cap = 2 * len(buf) + 1024
    out = np.zeros(cap, dtype=np.uint8)
    sink = MemorySink(out.ctypes.data, cap, 0)
    oi = OutputInterface()
    L.mgl_memory_output_new(C.byref(oi), C.byref(sink))
    slab = np.ascontiguousarray(slab, dtype=PACKET)
    if not L.mgl_emit_stream(_ptr(buf), len(buf), Properties(lc, lp, pb), _ptr(slab), C.byref(oi)):
        raise MglError("mgl_emit_stream failed (invalid slab?)")
    assert sink.len <= cap
    return out[: sink.len].tobytes()


class SA:
    """One mgl_sa handle = one SA chain resident on one GPU."""

    def __init__(self, data: bytes, neighbours_per_step=4096, seed=1673551, top_k=20, lc=0, lp=0, pb=0,
                 dict_limit=0, max_bucket_scan=0, iters_per_epoch=0, device=0, timing=False, fullwalk=False,
                 snapshots=True, serial_build=False, flags=0, accept=None, bulk_threshold=0):
        self.L = hip_lib()
        if self.L.mgl_device_count() < 1:
            raise MglError("no HIP device visible: the search path has no CPU implementation")
        self.data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self.n = len(self.data)
        self.K = neighbours_per_step
        self.props = Properties(lc, lp, pb)
        self.cfg = Config(seed, neighbours_per_step, top_k, dict_limit, max_bucket_scan, iters_per_epoch, device,
                          (F_TIMING if timing else 0) | (F_FULLWALK if fullwalk else 0)
                          | (0 if snapshots else F_NO_SNAPSHOTS) | (F_SERIAL_BUILD if serial_build else 0) | flags)
        self.h = self.L.mgl_sa_create(_ptr(self.data), self.n, self.props, C.byref(self.cfg))
        if not self.h:
            raise MglError(self.L.mgl_last_error().decode())
        self.nprobs = 1847 + (0x300 << (lc + lp))
        if accept is not None:
            self.set_accept_mode(accept, bulk_threshold)

    def set_accept_mode(self, mode, bulk_threshold=0):
        """ACCEPT_AUTO (default) / ACCEPT_SINGLE / ACCEPT_BULK, or the strings "auto" / "single" / "bulk"."""
        mode = {"auto": ACCEPT_AUTO, "single": ACCEPT_SINGLE, "bulk": ACCEPT_BULK}.get(mode, mode)
        self._chk(self.L.mgl_sa_set_accept_mode(self.h, mode, bulk_threshold))

    def step_modes(self) -> np.ndarray:
        """per step of the last run(): 0 single, 1 bulk"""
        cnt = C.c_size_t(0)
        self._chk(self.L.mgl_sa_step_modes(self.h, None, 0, C.byref(cnt)))
        out = np.zeros(max(1, cnt.value), dtype=np.uint8)
        self._chk(self.L.mgl_sa_step_modes(self.h, _ptr(out), cnt.value, C.byref(cnt)))
        return out[: cnt.value]

    def close(self):
        if getattr(self, "h", None):
            self.L.mgl_sa_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise MglError(f"rc={rc}: {self.L.mgl_last_error().decode()}")

    def begin_epoch(self, phase=0, from_best=False):
        self._chk(self.L.mgl_sa_begin_epoch(self.h, phase, int(from_best)))

    def set_slab(self, slab):
        slab = np.ascontiguousarray(slab, dtype=PACKET)
        self._chk(self.L.mgl_sa_set_slab(self.h, _ptr(slab)))

    def seed_greedy(self, candidates: int = 256):
        """Current slab := greedy LZ parse made on the device (opt-in starting point)."""
        self._chk(self.L.mgl_sa_seed_greedy(self.h, candidates))

    def set_temperature(self, temperature: int):
        """Opt-in Metropolis accept rule, temperature in cost units (16384 per byte); 0 = reference rule."""
        self._chk(self.L.mgl_sa_set_temperature(self.h, temperature))

    def set_best(self, slab, perplexity: int):
        slab = np.ascontiguousarray(slab, dtype=PACKET)
        self._chk(self.L.mgl_sa_set_best(self.h, _ptr(slab), perplexity))

    def best_packed(self):
        """packets_best in the packed device form (u64 per position) and its cost"""
        out = np.zeros(self.n, dtype=np.uint64)
        cost = C.c_uint64(0)
        self._chk(self.L.mgl_sa_best_packed(self.h, _ptr(out), C.byref(cost)))
        return out, cost.value

    def best_cost(self) -> int:
        cost = C.c_uint64(0)
        self._chk(self.L.mgl_sa_best_packed(self.h, None, C.byref(cost)))
        return cost.value

    def adopt_best_packed(self, packed, perplexity: int):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        assert len(packed) == self.n
        self._chk(self.L.mgl_sa_adopt_best_packed(self.h, _ptr(packed), perplexity))

    def exchange_best(self, comm: "Comm"):
        """collective over the communicator's ranks (RCCL); returns (winner rank, winner cost)"""
        w, c = C.c_int(-1), C.c_uint64(0)
        self._chk(self.L.mgl_sa_exchange_best(self.h, comm.h, C.byref(w), C.byref(c)))
        return w.value, c.value

    def run(self, steps: int) -> dict:
        st = Stats()
        self._chk(self.L.mgl_sa_run(self.h, steps, C.byref(st)))
        return st.asdict()

    def current(self):
        out = np.zeros(self.n, dtype=PACKET)
        cost = C.c_uint64(0)
        self._chk(self.L.mgl_sa_current(self.h, _ptr(out), C.byref(cost)))
        return out, cost.value

    def best(self):
        out = np.zeros(self.n, dtype=PACKET)
        cost = C.c_uint64(0)
        self._chk(self.L.mgl_sa_best(self.h, _ptr(out), C.byref(cost)))
        return out, cost.value

    def cost_slab(self, slab, want_cum=True):
        slab = np.ascontiguousarray(slab, dtype=PACKET)
        total, npk = C.c_uint64(0), C.c_size_t(0)
        cum = np.zeros(self.n, dtype=np.uint64) if want_cum else None
        self._chk(self.L.mgl_cost_slab(self.h, _ptr(slab), C.byref(total), _ptr(cum), C.byref(npk)))
        return dict(total=total.value, npackets=npk.value, cum=None if cum is None else cum[: npk.value].copy())

    def final_state(self, slab):
        slab = np.ascontiguousarray(slab, dtype=PACKET)
        probs = np.zeros(self.nprobs, dtype=np.uint16)
        cs = C.c_uint8(0)
        dists = np.zeros(4, dtype=np.uint32)
        self._chk(self.L.mgl_final_state(self.h, _ptr(slab), _ptr(probs), self.nprobs, C.byref(cs), _ptr(dists)))
        return dict(probs=probs, ctx_state=cs.value, dists=dists)

    def top_k(self, slab, position):
        slab = np.ascontiguousarray(slab, dtype=PACKET)
        out = np.zeros(64, dtype=PACKET)
        costs = np.zeros(64, dtype=np.uint64)
        cnt = C.c_size_t(0)
        self._chk(self.L.mgl_top_k(self.h, _ptr(slab), position, _ptr(out), _ptr(costs), C.byref(cnt)))
        return out[: cnt.value].copy(), costs[: cnt.value].copy()

    def substrings(self, pos, max_len=273, cap=1 << 20):
        offs = np.zeros(cap, dtype=np.uint32)
        lens = np.zeros(cap, dtype=np.uint32)
        cnt = C.c_size_t(0)
        self._chk(self.L.mgl_substrings(self.h, pos, max_len, _ptr(offs), _ptr(lens), cap, C.byref(cnt)))
        assert cnt.value <= cap
        return offs[: cnt.value].copy(), lens[: cnt.value].copy()

    def batch_counters(self):
        """(bulk steps whose moves were patched into the base by the batch accept, bulk steps that began one and fell back to the rebuild)"""
        v = self.debug_dump(80, np.uint64)
        return int(v[0]), int(v[1])

    def debug_set(self, key: int, value: int):
        self._chk(self.L.mgl_debug_set(self.h, key, value))

    def debug_dump(self, what: int, dtype) -> np.ndarray:
        need = C.c_size_t(0)
        probe = np.zeros(1, dtype=np.uint8)
        self.L.mgl_debug_dump(self.h, what, _ptr(probe), 0, C.byref(need))
        out = np.zeros(max(1, need.value), dtype=np.uint8)
        self._chk(self.L.mgl_debug_dump(self.h, what, _ptr(out), need.value, C.byref(need)))
        return out[: need.value].view(dtype)

    def neighbours(self, global_step: int, want_diffs=True, diff_cap=64):
        costs = np.zeros(self.K, dtype=np.uint64)
        nd = np.zeros(self.K, dtype=np.uint32)
        diffs = np.zeros((self.K, diff_cap), dtype=DIFF) if want_diffs else None
        self._chk(self.L.mgl_neighbours(self.h, global_step, _ptr(costs), _ptr(diffs), _ptr(nd), diff_cap))
        return costs, nd, diffs


class Comm:
    """One communicator behind the C ABI (mgl_comm_*).  Comm(uid, rank, world, device): RCCL, rank 0 makes the id and
    every rank joins.  Comm.shm(path, nonce, rank, world, device): the library's host shared-memory transport."""

    @classmethod
    def shm(cls, path: str, nonce: int, rank: int, world: int, device: int = 0) -> "Comm":
        self = cls.__new__(cls)
        self.L = hip_lib()
        self.h = C.c_void_p()
        if self.L.mgl_comm_init_shm(C.byref(self.h), os.fsencode(path), nonce, rank, world, device) != 0:
            self.h = None
            raise MglError(self.L.mgl_last_error().decode())
        self.rank, self.world = rank, world
        return self

    def min_u64(self, mine: int) -> int:
        out = C.c_uint64(0)
        if self.L.mgl_comm_min_u64(self.h, mine, C.byref(out)) != 0:
            raise MglError(self.L.mgl_last_error().decode())
        return out.value

    @staticmethod
    def unique_id() -> bytes:
        L = hip_lib()
        buf = (C.c_uint8 * 128)()
        if L.mgl_comm_unique_id(buf) != 0:
            raise MglError(L.mgl_last_error().decode())
        return bytes(buf)

    def __init__(self, uid: bytes, rank: int, world: int, device: int):
        self.L = hip_lib()
        assert len(uid) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self.h = C.c_void_p()
        if self.L.mgl_comm_init(C.byref(self.h), buf, rank, world, device) != 0:
            raise MglError(self.L.mgl_last_error().decode())
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "h", None):
            self.L.mgl_comm_destroy(self.h)
            self.h = None

    __del__ = close
