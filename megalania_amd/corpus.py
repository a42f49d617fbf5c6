"""Seeded synthetic inputs shaped like BASELINE.json's configs.

No corpora exist in the container or on the GPU box and there is no network, so every
bench / parity input is generated here, deterministically, from a counter-based
splitmix64 stream (no dependence on numpy's Generator bit streams).

  C1  lorem4k   4 096 B      the 445-char lorem-ipsum paragraph + ' ', repeated, truncated
  C2  enwik5    100 000 B    enwik-shaped: Zipf(1.0) words, punctuation, [[links]], <page> wrappers
  C3  dickens   10 192 446 B prose-only variant
  C4  enwik8    100 000 000 B C2 generator, other seed
  C5  elf1m     1 048 576 B  first 1 MiB of /usr/bin/python3.10 when its sha256 matches the
                             survey's (same image on both boxes), else a synthetic ELF-like mix

If real corpora are found under $MGL_CORPUS_DIR (enwik8, dickens) they are used instead
and `describe()` says so.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

_LOREM = (
    "Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor "
    "incididunt ut labore et dolore magna aliqua. Ut enim ad minim veniam, quis nostrud "
    "exercitation ullamco laboris nisi ut aliquip ex ea commodo consequat. Duis aute irure "
    "dolor in reprehenderit in voluptate velit esse cillum dolore eu fugiat nulla pariatur. "
    "Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia deserunt "
    "mollit anim id est laborum."
)
LOREM4K_SHA256 = "e983b3dab739d02999a9d4089820e29dfd1030f640e8f5ce5c8ebffcd9719bc7"
ELF1M_SHA256_PREFIX = "23f95401"

_U64 = np.uint64
_GOLD = _U64(0x9E3779B97F4A7C15)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised (uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = z + _GOLD
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _stream(seed: int, lane: int, start: int, count: int) -> np.ndarray:
    """count u64 values number start.. of stream (seed, lane)."""
    with np.errstate(over="ignore"):
        base = _mix64(np.array([seed ^ (lane * 0xD1B54A32D192ED03 & 0xFFFFFFFFFFFFFFFF)], dtype=_U64))[0]
        idx = np.arange(start, start + count, dtype=_U64)
        return _mix64(base + idx * _GOLD)


def _uniform(seed: int, lane: int, start: int, count: int) -> np.ndarray:
    return (_stream(seed, lane, start, count) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def lorem(n: int = 4096) -> bytes:
    reps = n // (len(_LOREM) + 1) + 1
    return ((_LOREM + " ") * reps)[:n].encode("ascii")


# English letter frequencies (per mille, a..z)
_LETTER_W = np.array(
    [82, 15, 28, 43, 127, 22, 20, 61, 70, 2, 8, 40, 24, 67, 75, 19, 1, 60, 63, 91, 28, 10, 24, 2, 20, 1],
    dtype=np.float64,
)


def _vocabulary(seed: int, size: int = 8192):
    """size synthetic words: lengths 1..12 (short words are the frequent ones), letters by
    English frequency.  Returns (flat byte array, start offsets, lengths)."""
    u_len = _uniform(seed, 1, 0, size)
    rank = np.arange(size)
    # frequent words are short: mean length grows slowly with rank
    mean = 2.0 + 1.1 * np.log2(2.0 + rank / 6.0)
    lens = np.clip(np.rint(mean * (0.55 + 0.9 * u_len)), 1, 14).astype(np.int64)
    total = int(lens.sum())
    cdf = np.cumsum(_LETTER_W) / _LETTER_W.sum()
    letters = np.searchsorted(cdf, _uniform(seed, 2, 0, total), side="right").clip(0, 25)
    flat = (letters + ord("a")).astype(np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    return flat, starts, lens


_SEPS = [b" ", b", ", b". ", b".\n\n", b"; ", b" (", b") ", b"]] ", b" [[", b"\n"]


def _text(n: int, seed: int, markup: bool) -> bytes:
    flat, starts, lens = _vocabulary(seed)
    vsize = len(lens)
    # Zipf(1.0) inverse CDF over ranks
    w = 1.0 / np.arange(1, vsize + 1)
    zcdf = np.cumsum(w) / w.sum()
    sep_bytes = [np.frombuffer(s, dtype=np.uint8) for s in _SEPS]
    if markup:
        sep_p = np.array([0.80, 0.07, 0.055, 0.008, 0.006, 0.006, 0.006, 0.017, 0.017, 0.015])
    else:
        sep_p = np.array([0.84, 0.08, 0.06, 0.012, 0.008, 0.0, 0.0, 0.0, 0.0, 0.0])
    sep_cdf = np.cumsum(sep_p) / sep_p.sum()
    page_open = np.frombuffer(b"</text></page>\n<page><title>", dtype=np.uint8)
    page_mid = np.frombuffer(b"</title><text>", dtype=np.uint8)

    out = np.empty(n + 4096, dtype=np.uint8)
    filled = 0
    word_no = 0
    chunk_words = 1 << 18
    while filled < n:
        ids = np.searchsorted(zcdf, _uniform(seed, 3, word_no, chunk_words), side="right").clip(0, vsize - 1)
        # topical locality: a quarter of the tokens repeat a word seen 1..64 tokens earlier
        u_rep = _uniform(seed, 8, word_no, chunk_words)
        lag = 1 + (_stream(seed, 9, word_no, chunk_words) & _U64(63)).astype(np.int64)
        src_i = np.arange(chunk_words) - lag
        rep = (u_rep < 0.25) & (src_i >= 0)
        ids = np.where(rep, ids[np.maximum(src_i, 0)], ids)
        seps = np.searchsorted(sep_cdf, _uniform(seed, 4, word_no, chunk_words), side="right").clip(0, len(_SEPS) - 1)
        wl = lens[ids]
        sl = np.array([len(s) for s in sep_bytes], dtype=np.int64)[seps]
        tok_len = wl + sl
        offs = np.concatenate([[0], np.cumsum(tok_len)])
        total = int(offs[-1])
        buf = np.empty(total, dtype=np.uint8)
        # words: gather from the flat vocabulary
        tok_of = np.repeat(np.arange(chunk_words), wl)
        within = np.arange(int(wl.sum())) - np.repeat(np.cumsum(wl) - wl, wl)
        buf[offs[tok_of] + within] = flat[starts[ids][tok_of] + within]
        # separators
        for k, sb in enumerate(sep_bytes):
            sel = np.nonzero(seps == k)[0]
            if len(sel) == 0:
                continue
            base = offs[sel] + wl[sel]
            for b in range(len(sb)):
                buf[base + b] = sb[b]
        # capitalise the word after a sentence end
        cap = np.nonzero((seps[:-1] == 2) | (seps[:-1] == 3))[0] + 1
        buf[offs[cap]] -= 32
        if markup:
            # every ~2k words close the page and open a new one titled by the next word
            marks = np.arange(1500 + (word_no % 700), chunk_words - 2, 2048)
            pieces = []
            prev = 0
            for m in marks:
                pieces.append(buf[prev:offs[m]])
                pieces.append(page_open)
                pieces.append(buf[offs[m]:offs[m] + wl[m]])
                pieces.append(page_mid)
                prev = offs[m]
            pieces.append(buf[prev:])
            buf = np.concatenate(pieces)
        take = min(len(buf), len(out) - filled)
        out[filled:filled + take] = buf[:take]
        filled += take
        word_no += chunk_words
    head = b"<page><title>Synthetic</title><text>" if markup else b""
    body = out[: n - len(head)].tobytes()
    return head + body


def enwik_like(n: int, seed: int) -> bytes:
    return _text(n, seed, markup=True)


def prose_like(n: int, seed: int) -> bytes:
    return _text(n, seed, markup=False)


def _synthetic_elf(n: int, seed: int) -> bytes:
    """60 % opcode-like Markov bytes, 25 % zero-padded tables, 15 % string table."""
    out = np.empty(n, dtype=np.uint8)
    u = _uniform(seed, 5, 0, n)
    ops = np.array([0x48, 0x8B, 0x89, 0xE8, 0x0F, 0x85, 0xC3, 0x55, 0x5D, 0x41, 0xFF, 0x00, 0x24, 0x45, 0x83, 0xC0],
                   dtype=np.uint8)
    code = np.where(u < 0.7, ops[(u * 997).astype(np.int64) % len(ops)], (u * 256 * 31).astype(np.int64) % 256).astype(np.uint8)
    out[:] = code
    a, b = int(n * 0.60), int(n * 0.85)
    tbl = np.zeros(b - a, dtype=np.uint8)
    idx = np.arange(0, b - a - 8, 8)
    tbl[idx] = (_stream(seed, 6, 0, len(idx)) & _U64(0xFF)).astype(np.uint8)
    tbl[idx + 1] = (_stream(seed, 7, 0, len(idx)) & _U64(0x0F)).astype(np.uint8)
    out[a:b] = tbl
    strs = prose_like(n - b, seed ^ 0x55).replace(b" ", b"\0")
    out[b:] = np.frombuffer(strs, dtype=np.uint8)
    return out.tobytes()


def elf1m(n: int = 1 << 20) -> tuple[bytes, str]:
    path = "/usr/bin/python3.10"
    try:
        with open(path, "rb") as f:
            blob = f.read(1 << 20)
        if len(blob) == (1 << 20) and hashlib.sha256(blob).hexdigest().startswith(ELF1M_SHA256_PREFIX):
            return blob[:n], "first 1 MiB of /usr/bin/python3.10 (sha256 23f95401...)"
    except OSError:
        pass
    return _synthetic_elf(n, 0xEF), "synthetic ELF-like mix (seed 0xEF)"


def _real(name: str):
    d = os.environ.get("MGL_CORPUS_DIR")
    if d and os.path.isfile(os.path.join(d, name)):
        with open(os.path.join(d, name), "rb") as f:
            return f.read()
    return None


def config_input(cfg: str, n: int | None = None) -> tuple[bytes, str]:
    """(bytes, description) for BASELINE config 'c1'..'c5'; n overrides the size."""
    cfg = cfg.lower()
    if cfg == "c1":
        return lorem(n or 4096), "lorem4k (sha256 e983b3da...)"
    if cfg == "c2":
        real = _real("enwik8")
        if real is not None:
            return real[: n or 100_000], "real enwik8 prefix from $MGL_CORPUS_DIR"
        return enwik_like(n or 100_000, 0xE5), "synthetic enwik-shaped text, seed 0xE5"
    if cfg == "c3":
        real = _real("dickens")
        if real is not None:
            return real[: n or len(real)], "real dickens from $MGL_CORPUS_DIR"
        return prose_like(n or 10_192_446, 0xD1), "synthetic prose, seed 0xD1"
    if cfg == "c4":
        real = _real("enwik8")
        if real is not None:
            return real[: n or 100_000_000], "real enwik8 from $MGL_CORPUS_DIR"
        return enwik_like(n or 100_000_000, 0xE8), "synthetic enwik-shaped text, seed 0xE8"
    if cfg == "c5":
        return elf1m(n or (1 << 20))
    raise ValueError(f"unknown config {cfg!r}")
