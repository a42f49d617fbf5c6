"""In-tree build of the native pieces (no JIT cache: the built files travel with the repo).

  _build/libmegalania_hip.so   hipcc --offload-arch=gfx950   csrc/mgl_api.hip (+ kernels)
  _build/libmegalania_host.so  gcc (C)                       host/mgl_host.c
  _build/megalania-hip         gcc (C) CLI                   host/main.c, links both

`python -m megalania_amd.build` or `build_all()`.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_build")

HIP_SRC = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))]
HIP_SRC.append(os.path.join(ROOT, "include", "megalania_hip.h"))
HOST_SRC = [os.path.join(HERE, "host", f) for f in ("mgl_host.c", "mgl_host.h")]
HOST_SRC += [os.path.join(HERE, "csrc", "mgl_model.h"), os.path.join(HERE, "csrc", "mgl_cost_table.inc"),
             os.path.join(ROOT, "include", "megalania_interfaces.h"), os.path.join(ROOT, "include", "megalania_hip.h")]

HIP_SO = os.path.join(OUT, "libmegalania_hip.so")
HOST_SO = os.path.join(OUT, "libmegalania_host.so")
CLI = os.path.join(OUT, "megalania-hip")


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    # everything a build says goes to stderr: stdout belongs to the caller (bench.py prints one JSON line there)
    print("+", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd, stdout=sys.stderr)


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def build_hip(force=False):
    os.makedirs(OUT, exist_ok=True)
    if force or _stale(HIP_SO, HIP_SRC):
        _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
              "-o", HIP_SO, os.path.join(HERE, "csrc", "mgl_api.hip")])
    return HIP_SO


def build_host(force=False):
    os.makedirs(OUT, exist_ok=True)
    cc = os.environ.get("CC", "gcc")
    if force or _stale(HOST_SO, HOST_SRC):
        _run([cc, "-std=gnu11", "-O2", "-g", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", HOST_SO,
              os.path.join(HERE, "host", "mgl_host.c")])
    if force or _stale(CLI, HOST_SRC + [os.path.join(HERE, "host", "main.c"), HIP_SO]):
        _run([cc, "-std=gnu11", "-O2", "-g", "-Wall", "-Wextra", "-o", CLI, os.path.join(HERE, "host", "main.c"),
              os.path.join(HERE, "host", "mgl_host.c"), "-L" + OUT, "-lmegalania_hip", "-Wl,-rpath,$ORIGIN"])
    return HOST_SO


def build_all(force=False):
    """The product's native pieces only.  The test-side checkers (oracle/) are built by
    tools/build_oracle.py -- nothing in this package knows about them."""
    build_hip(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
