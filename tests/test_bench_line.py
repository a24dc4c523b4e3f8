"""CPU: the line bench.py prints LAST is compact.  The driver keeps a 16 KB tail of stdout; round 5's line had grown to 24 KB and
could not be parsed (BENCH_r05.json: parsed = null).  The emitter is run here on a canned full record with every section
present and absurdly long strings, and on a record where every section failed."""
import json
import os
import sys

import bench

LONG = "x" * 5000


def _full():
    roof = {"bound": "hbm", "kernel": "apply_quad_kernel<F16,RGBA8888,scale1> 7680x4320, one frame per launch", "achieved": 5540.1, "peak": 8000.0,
            "unit": "GB/s", "frac": 0.6925, "traffic": 449000000, "traffic_source": LONG, "algorithmic_bytes": 447897600, "avg_launch_us": 80.85,
            "launches_timed": 60, "launch_us": {"min": 1, "p10": 1, "median": 1, "p90": 1, "max": 1, "n": 60},
            "north_star_8k": {"mapC": {"note": LONG}}, "ns8k_mapC_frac": 0.69, "ns8k_mapB_frac": 0.66, "ns8k_mapA_cold_frac": 0.6,
            "ns8k_mapA_hot_frac": 0.74, "onbox_copy_note": LONG}
    return {
        "metric": bench.METRIC, "value": 6870.4, "unit": "Mpixels/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.2073,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": LONG, "frames_per_rank_per_step": 1, "clock_ramp": LONG, "sharding": LONG},
        "step": {"kernel_families_us_per_step": {f: {"us": 1.0, "launches": 3} for f in "abcdef"}, "note": LONG},
        "roofline": roof,
        "api1_roundtrip": {"api1_4k_enc_us": 518.3, "api1_4k_dec_us": 689.0, "api1_8k_enc_us": 1425.0, "api1_8k_dec_us": 1553.0,
                           "api1_8k_roundtrip_Mpxs": 11139.0, "api1_note": LONG},
        "api1_concurrent": {"frames_in_flight_2_Mpxs": 15000.0, "frames_in_flight_4_Mpxs": 24000.0, "workload": LONG},
        "config5": {"frac_of_8TBs": 0.367, "workload": LONG}, "headline_16x4k": {"frac": 0.67, "workload": LONG},
        "config4": {"ms_per_image": 1.9, "all_reduce_us_back_to_back": 21.0, "full_16k_x_16k_one_gpu": {"ms_per_image": 14.2}, "workload": LONG},
        "encode": {"blob": [LONG] * 4}, "extra": {"blob": [LONG] * 8}, "api_level": {"blob": LONG, "uhdr_encode_api0_8k_hip": {"ms": 31.0}},
        "cpu_baseline": {"value": 9.6, "unit": "Mpixels/s", "cores": 4, "kind": "reference", "sample": LONG, "stages": {"blob": LONG}},
    }


def test_compact_line_is_small_and_complete():
    line = bench.compact_line(_full(), "bench_detail.json")
    assert len(line) <= bench.COMPACT_LIMIT <= 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("Mpixels/s encode+decode (API-1 P010+YUV420")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    extras = [k for k in r if k not in bench.ROOFLINE_KEYS]
    assert set(extras) == {n for n, _ in bench.ROOFLINE_SCALARS} and all(isinstance(r[k], (int, float)) for k in extras)
    assert set(d["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert all(not isinstance(v, (dict, list)) for v in r.values())  # scalars only: nothing a reader has to dig for


def test_compact_line_survives_failed_sections():
    full = _full()
    for k in ("api1_roundtrip", "api1_concurrent", "config5", "headline_16x4k", "config4", "encode", "extra", "api_level"):
        full[k] = {"error": "RuntimeError: " + LONG}
    full["roofline"] = {"bound": "hbm", "kernel": None, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None, "error": LONG}
    full["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: " + LONG}
    line = bench.compact_line(full, None)
    assert len(line) <= bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d["value"] == 6870.4 and d["roofline"]["frac"] is None and "config4" in d["sections_with_errors"]


def test_emit_prints_the_compact_line_last(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_full())
    out = capsys.readouterr().out.strip().splitlines()
    assert out[-2].startswith("bench detail: ") and json.loads(out[-1])["detail"] == bench.DETAIL_NAME
    detail = json.load(open(os.path.join(str(tmp_path), bench.DETAIL_NAME)))
    assert detail["extra"]["blob"][0] == LONG  # the full record keeps everything


def test_bench_never_switches_the_stderr_trace_on():
    src = open(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "bench.py")).read()
    assert "UHDR_HIP_SEAM_TRACE" not in src
