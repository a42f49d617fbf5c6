"""Drop-in proof: oracle/_ref/megalania_dropin is the reference's main.c with its
SA loop replaced by the C ABI, linked with the reference's OWN unchanged range_encoder.c, lzma_header_encoder.c,
lzma_packet_encoder.c, lzma_state.c, probability_model.c, lzma_packet.c, file_output.c, packet_slab.c,
memory_mapper.c (compiled in place in the build container; the binary travels) and libmegalania_hip.so.
Its stream must equal the standalone path's (same search, host emitter of megalania_amd/host) byte for byte and
decode to the input.  Also: the reference's own SA end states (golden fixtures, glibc rand() trajectories) cost
the same on the device."""
import lzma
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import slab_from_rle
from megalania_amd import binding, corpus

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "megalania_dropin")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/megalania_dropin is built where /root/reference exists")
def test_reference_emission_objects_emit_the_hip_search_result(tmp_path):
    data = corpus.enwik_like(20000, 0x71)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    steps, K = 80, 512
    r = subprocess.run([DROPIN, str(f), str(steps), str(K)], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    stream = r.stdout
    assert lzma.decompress(stream, format=lzma.FORMAT_ALONE) == data
    if shutil.which("xz"):
        x = subprocess.run(["xz", "-dc", "--format=lzma"], input=stream, capture_output=True, timeout=120)
        assert x.returncode == 0 and x.stdout == data
    # the same search through the standalone path: same slab, same bytes out of the rewritten host emitter
    sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=len(data))
    st = sa.run(steps)
    best, cost = sa.best()
    assert f"best perplexity: {cost}".encode() in r.stderr
    assert binding.emit_stream(data, best) == stream
    assert abs((18 + cost / 16384) - len(stream)) <= 4  # main.c:97's estimate against the real stream
    assert st["accepted"] > steps  # bulk steps took part
    sa.close()


def test_reference_sa_end_states_cost_the_same_on_the_device(golden, golden_input):
    """golden["sa"]: slabs the compiled reference reached under glibc rand() (main.c:78-102).  The device's walk
    gives them the reference's own figure, accepts them as valid parses, and they round-trip."""
    for s in golden["sa"]:
        data = golden_input(s["input"])
        slab = slab_from_rle(len(data), s["final_packets"]).astype(binding.PACKET)
        sa = binding.SA(data, neighbours_per_step=16)
        assert sa.cost_slab(slab, want_cum=False)["total"] == s["cur"], s["input"]
        sa.set_slab(slab)  # k_validate: every packet reproduces the input
        assert sa.current()[1] == s["cur"]
        assert lzma.decompress(binding.emit_stream(data, slab), format=lzma.FORMAT_ALONE) == data
        sa.close()
