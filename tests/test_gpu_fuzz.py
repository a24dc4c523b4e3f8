"""The randomised parity sweep (tests/fuzz_parity.py) as a collected -m gpu test with a bounded budget, so the driver's
round-end GPU run executes it and its summary is kept (gpurun_out/fuzz_parity.log on the GPU box; copied to profiles/)."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_parity_bounded(hip_ctx):
    import fuzz_parity as FZ

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seconds = float(os.environ.get("UHDR_FUZZ_SECONDS", "40"))
    summary, bad = FZ.run(seconds, seed=int(os.environ.get("UHDR_FUZZ_SEED", "2")), context=hip_ctx,
                          log=os.path.join(root, "gpurun_out", "fuzz_parity.log"))
    assert bad == 0, summary
    assert sum(v[0] for v in summary.values()) > 50, summary
