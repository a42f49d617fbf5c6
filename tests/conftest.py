import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from megalania_amd import corpus  # noqa: E402
import _libs  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Native pieces are built in-tree (git-ignored): make sure they exist and are current -- a no-op
    when `python -m megalania_amd.build` / __graft_entry__.build() has already run."""
    if os.environ.get("MGL_NO_AUTOBUILD"):
        return
    from megalania_amd import build as _build
    _build.build_all()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_oracle
    build_oracle.build_oracle()


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_slab(slab) -> str:
    """field-wise hash (the 12-byte record has 5 padding bytes whose content is arbitrary)"""
    return sha(np.concatenate([slab["type"].astype(np.uint32), slab["dist"], slab["len"].astype(np.uint32)]))


def rand_bytes(n, seed):
    return (corpus._stream(seed, 11, 0, n) & np.uint64(0xFF)).astype(np.uint8).tobytes()


def far_data(n, seed):
    key = rand_bytes(512, seed)
    body = bytearray(corpus.enwik_like(n - 1024, seed ^ 0x33))
    body[70_000:70_256] = key[:256]
    return key + bytes(body) + key


def materialise(spec) -> bytes:
    """Turn a fixture's input descriptor into bytes (mirror of tools/make_golden.py)."""
    if "hex" in spec:
        return bytes.fromhex(spec["hex"])
    g = spec["gen"]
    if g == "lorem":
        return corpus.lorem(spec["n"])
    if g == "enwik_like":
        return corpus.enwik_like(spec["n"], spec["seed"])
    if g == "rand_bytes":
        return rand_bytes(spec["n"], spec["seed"])
    if g == "far":
        return far_data(spec["n"], spec["seed"])
    raise KeyError(spec)


def slab_from_rle(n, packets) -> np.ndarray:
    """Inverse of tools/make_golden.py:pk_list -- lay the walked packets down from byte 0."""
    s = _libs.literal_slab(n)
    pos = 0
    for p in packets:
        if p[0] == "L":
            pos += p[1]
        else:
            s[pos] = (p[0], p[1], p[2])
            pos += p[2]
    assert pos == n, (pos, n)
    return s


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


_INPUT_CACHE = {}


@pytest.fixture(scope="session")
def golden_input(golden):
    def get(name) -> bytes:
        if name not in _INPUT_CACHE:
            _INPUT_CACHE[name] = materialise(golden["inputs"][name])
        return _INPUT_CACHE[name]

    return get
