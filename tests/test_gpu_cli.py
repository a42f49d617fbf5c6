"""The C driver end to end on a GPU: `megalania-hip <file>` writes an LZMA-alone stream to stdout
that xz / liblzma decode back to the input (the reference's only end-to-end check)."""
import lzma
import shutil
import subprocess

import pytest

from megalania_amd import build, corpus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("args,size", [(["--epochs", "2", "--phases", "2", "--neighbours", "256"], 3000),
                                       (["--epochs", "1", "--phases", "1", "--neighbours", "512", "--steps", "40", "--pb", "2"], 5000)])
def test_cli_roundtrip(tmp_path, args, size):
    data = corpus.enwik_like(size, 0x77)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    r = subprocess.run([build.CLI] + args + [str(f)], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert b"current file size" in r.stderr
    stream = r.stdout
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    assert len(stream) < len(data) * 0.8
    if shutil.which("xz"):
        x = subprocess.run(["xz", "-dc", "--format=lzma"], input=stream, capture_output=True, timeout=60)
        assert x.returncode == 0 and x.stdout == data


def test_cli_empty_and_missing(tmp_path):
    f = tmp_path / "empty"
    f.write_bytes(b"")
    r = subprocess.run([build.CLI, str(f)], capture_output=True, timeout=60)
    assert r.returncode == 0 and r.stdout == b""  # main.c:40-42
    r = subprocess.run([build.CLI, str(tmp_path / "nope")], capture_output=True, timeout=60)
    assert r.returncode != 0 and r.stdout == b""
    r = subprocess.run([build.CLI], capture_output=True, timeout=60)
    assert r.returncode != 0 and b"usage" in r.stderr  # main.c:29-32
