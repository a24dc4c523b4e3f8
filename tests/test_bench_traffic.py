"""CPU: bench.py reports roofline.traffic only for the library it was measured with (profiles/traffic.json carries the SHA-256
of that libuhdr_hip.so; the build is deterministic, so the tree that was profiled rebuilds to the same hash)."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEY = "apply_quad_kernel<F16,RGBA8888,scale1>|1x7680x4320"


def _entry():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        d = json.load(f)
    if KEY not in d:
        pytest.skip("profiles/traffic.json predates the 8K roofline kernel: re-run tools/profile_bench.sh + tools/update_traffic.py")
    return d[KEY]


def test_committed_traffic_belongs_to_this_trees_library():
    if not os.path.exists(os.path.join(ROOT, "libultrahdr_amd", "lib", "libuhdr_hip.so")):
        pytest.skip("library not built")
    e = _entry()
    traffic, source = bench.measured_traffic(KEY)
    if e["library_sha256"] != bench.library_sha256():  # a kernel changed since the last profile run: bench.py reports null until
        pytest.skip("profiles/traffic.json is stale for this library: re-run tools/profile_bench.sh + tools/update_traffic.py")
    assert traffic == e["traffic_bytes_per_launch"] and "FETCH_SIZE" in source
    # HBM bytes of the roofline kernel's launch: within 2 % of the algorithmic bytes (7680 x 4320 x 13.5 B)
    algorithmic = 7680 * 4320 * 13.5
    assert 0.98 * algorithmic < traffic < 1.02 * algorithmic
    assert e["read_bytes"] + e["write_bytes"] == traffic


def test_another_binary_gets_null_and_a_reason(monkeypatch):
    _entry()  # (skips while the committed file predates the key)
    monkeypatch.setattr(bench, "library_sha256", lambda: "0" * 64)
    traffic, why = bench.measured_traffic(KEY)
    assert traffic is None and "another libuhdr_hip.so" in why
    assert bench.measured_traffic("no such kernel") == (None, None)
