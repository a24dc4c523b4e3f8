"""CPU: bench.py reports roofline.traffic only for the library it was measured with (profiles/traffic.json carries the SHA-256
of that libuhdr_hip.so; the build is deterministic, so the tree that was profiled rebuilds to the same hash)."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEY = "apply_quad_kernel<F16,RGBA8888,scale1>|16x3840x2160"


def _entry():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        return json.load(f)[KEY]


def test_committed_traffic_belongs_to_this_trees_library():
    if not os.path.exists(os.path.join(ROOT, "libultrahdr_amd", "lib", "libuhdr_hip.so")):
        pytest.skip("library not built")
    e = _entry()
    traffic, source = bench.measured_traffic(KEY)
    if e["library_sha256"] != bench.library_sha256():  # a kernel changed since the last profile run: bench.py reports null until
        pytest.skip("profiles/traffic.json is stale for this library: re-run tools/profile_bench.sh + tools/update_traffic.py")
    assert traffic == e["traffic_bytes_per_launch"] and "FETCH_SIZE" in source
    # HBM bytes of the headline launch: within 2 % of the algorithmic bytes (16 frames x 3840 x 2160 x 13.5 B)
    algorithmic = 16 * 3840 * 2160 * 13.5
    assert 0.98 * algorithmic < traffic < 1.02 * algorithmic
    assert e["read_bytes"] + e["write_bytes"] == traffic


def test_another_binary_gets_null_and_a_reason(monkeypatch):
    monkeypatch.setattr(bench, "library_sha256", lambda: "0" * 64)
    traffic, why = bench.measured_traffic(KEY)
    assert traffic is None and "another libuhdr_hip.so" in why
    assert bench.measured_traffic("no such kernel") == (None, None)
