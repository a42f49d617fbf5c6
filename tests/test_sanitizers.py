"""AddressSanitizer + UBSan over the C that runs on the host (GPU sanitizers are not available on this pool): the CPU
oracle (oracle/mgl_oracle.c) and the product's C host library (megalania_amd/host/mgl_host.c: range coder, header,
emission over the two vtables), both rebuilt with -fsanitize=address,undefined (oracle/Makefile `sanitizers`), then the
golden-vector and host tests run again in a child interpreter that loads those builds.  Any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_golden_walks_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitizers"], stdout=sys.stderr)
    env = dict(os.environ,
               LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               MGL_ORACLE_SO=os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so"),
               MGL_HOST_SO=os.path.join(ROOT, "oracle", "_build", "libmegalania_host_asan.so"),
               MGL_NO_AUTOBUILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_host.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
