"""Test infrastructure: build the CPU oracle (oracle/_build/liboracle.so) and, where
/root/reference exists (the build container only), the reference itself into oracle/_ref/
plus the drop-in link test program.  Called by tests/conftest.py, __graft_entry__.build() and
bench.py's cpu_baseline leg -- never by the product package."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_oracle():
    cmd = ["make", "-C", os.path.join(ROOT, "oracle")]
    print("+", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd, stdout=sys.stderr)


if __name__ == "__main__":
    build_oracle()
