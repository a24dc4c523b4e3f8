"""CPU-side checks of the drop-in libuhdr.so: symbol surface, the unaccelerated path is the
reference's, and asking for acceleration without a GPU fails loudly instead of computing on the CPU."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import facade_util as F
from tests import fixture720

pytestmark = pytest.mark.skipif(not F.built(), reason="facade not built (needs /root/reference at build time)")


def test_facade_exports_exactly_the_43_api_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", F.FACADE], capture_output=True, text=True, check=True).stdout
    syms = sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T")
    assert syms == sorted(F.API_SYMBOLS)


def _write_fixture(d):
    g = fixture720.gold()
    g["p010"].tofile(os.path.join(d, "in.p010"))
    g["yuv420"].tofile(os.path.join(d, "in.yuv420"))
    return os.path.join(d, "in.p010"), os.path.join(d, "in.yuv420")


def test_unaccelerated_facade_is_the_reference_on_config_1():
    """BASELINE config 1 through the relinked sample app without -u: the reference's CPU path, 85 449 bytes."""
    with tempfile.TemporaryDirectory() as d:
        p, y = _write_fixture(d)
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "cpu.jpg", False, d)
        assert rc == 0, err
        assert trace == []  # no stage touched the seam
        assert os.path.getsize(os.path.join(d, "cpu.jpg")) == 85449


def test_acceleration_without_a_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with tempfile.TemporaryDirectory() as d:
        p, y = _write_fixture(d)
        rc, _, err, _ = F.encode_api1(p, y, 1280, 720, "gpu.jpg", True, d)
        assert rc != 0
        assert "no CPU fallback" in err
        assert not os.path.exists(os.path.join(d, "gpu.jpg"))
