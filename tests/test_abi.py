"""The C-ABI shared library loads without a GPU and exports exactly what include/uhdr_hip.h
declares; pure-host entry points work; device entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from libultrahdr_amd import capi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "uhdr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uhdr_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = A.load()
    names = header_symbols()
    assert len(names) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", A.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(names) <= exported, sorted(set(names) - exported)
    assert exported == set(names), f"exports beyond the header: {sorted(exported - set(names))}"
    assert set(names) == set(A.ABI_SYMBOLS), "ctypes binding and header disagree"
    assert lib.uhdr_hip_version().startswith(b"libuhdr_hip")


def test_no_kernel_of_the_product_uses_scratch_memory():
    """Round 6: experiments that spill stay out of libuhdr_hip.so (the round-5 binary carried sixteen spilling A/B instantiations)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("spill_check", os.path.join(ROOT, "tools", "spill_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ks = mod.kernels(A.LIB_PATH)
    assert len(ks) > 100  # the metadata was really parsed
    spill = [(k["name"], k["private_segment_fixed_size"]) for k in ks if k["private_segment_fixed_size"] > 0]
    assert not spill, spill


def test_seam_stage_table_and_context_recycling_without_a_gpu():
    """Round 6: the stage tallies the facade reports into (uhdr_hip_seam_note / _stats / _stats_reset) are plain host code -- they work without
    a device --, rows come back in first-seen order, and the context-pool helpers answer for a null context."""
    lib = A.load()
    lib.uhdr_hip_seam_stats_reset()
    assert A.seam_stats() == {}
    lib.uhdr_hip_seam_note(b"jpeg_decode_scan", 1, 1.5)
    lib.uhdr_hip_seam_note(b"apply_gainmap", 1, 2.0)
    lib.uhdr_hip_seam_note(b"jpeg_decode_scan", 1, 0.5)
    lib.uhdr_hip_seam_note(b"tone_map", 0, 0.0)
    st = A.seam_stats()
    assert list(st) == ["jpeg_decode_scan", "apply_gainmap", "tone_map"]
    assert st["jpeg_decode_scan"] == {"device": 2, "reference": 0, "device_ms": 2.0, "last_ms": 0.5}
    assert st["tone_map"]["reference"] == 1 and st["tone_map"]["device"] == 0
    assert C.sizeof(A.SeamStage) == 40 + 3 * 8 + 2 * 8
    assert A.seam_stats(reset=True) == st and A.seam_stats() == {}
    assert lib.uhdr_hip_recycle(None, 0) == -1


def test_seam_stage_table_is_written_at_exit_when_asked(tmp_path):
    """UHDR_HIP_SEAM_STATS_FILE: how tests read the table out of a process they cannot call into (the reference's ultrahdr_app)."""
    import json
    import sys

    out = tmp_path / "stages.json"
    code = ("import ctypes as C\n"
            f"lib = C.CDLL({A.LIB_PATH!r})\n"
            "lib.uhdr_hip_seam_note.argtypes = [C.c_char_p, C.c_int, C.c_double]\n"
            "lib.uhdr_hip_seam_note(b'encode_api1_fused', 1, 3.25)\n"
            "lib.uhdr_hip_seam_note(b'uhdr_call', 1, 4.0)\n")
    env = dict(os.environ, UHDR_HIP_SEAM_STATS_FILE=str(out))
    subprocess.check_call([sys.executable, "-c", code], env=env)
    d = json.loads(out.read_text())
    assert d["encode_api1_fused"]["device"] == 1 and d["encode_api1_fused"]["device_ms"] == 3.25 and d["uhdr_call"]["first_seq"] == 1


def test_struct_layouts_match_reference_abi():
    # ultrahdr_api.h:220-283 on LP64: error info 264 B, raw image 64 B, metadata 72 B
    assert C.sizeof(A.ErrorInfo) == 264
    assert C.sizeof(A.RawImage) == 64
    assert A.RawImage.planes.offset == 24 and A.RawImage.stride.offset == 48
    assert C.sizeof(A.GainmapMetadata) == 72
    assert C.sizeof(A.EncodeCfg) == 36


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = A.load()
    err = A.ErrorInfo()
    h = lib.uhdr_hip_create(0, C.byref(err))
    assert not h and err.error_code == A.UHDR_CODEC_ERROR and b"no CPU fallback" in err.detail
    from libultrahdr_amd.ultrahdr import Context

    with pytest.raises(A.UhdrError):
        Context(0)
    st = lib.uhdr_hip_apply_gainmap(None, None, None, None, 0, 4, 1.0, None)
    assert st.error_code == A.UHDR_CODEC_INVALID_PARAM


def test_product_never_touches_the_oracle():
    """No file of the product package may import / load / link anything under oracle/."""
    pkg = os.path.join(ROOT, "libultrahdr_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"oracle[/.]", r"libuhdr_oracle", r"uhdr_ref", r"\buo_[a-z]", r"dlopen"):
                    if pat == "dlopen" and fn in ("rccl_bind.cpp", "rccl_bind.h"):
                        # the one place that loads a library at run time, and only RCCL: every quoted .so name is librccl
                        names = re.findall(r'"([^"]*\.so[^"]*)"', txt)
                        assert all("librccl" in n for n in names), names
                        continue
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dp, fn), pat)
    ldd = subprocess.check_output(["ldd", A.LIB_PATH], text=True)
    assert "uhdr_oracle" not in ldd and "uhdr_ref" not in ldd and "jpeg" not in ldd


def test_quant_table_host_entry_point():
    from oracle import loader as L

    lib = A.load()
    for q in (1, 10, 50, 75, 85, 95, 100):
        for chroma in (0, 1):
            t = (C.c_uint16 * 64)()
            lib.uhdr_hip_jpeg_quant_table(q, chroma, t)
            assert np.array_equal(np.frombuffer(t, dtype=np.uint16), L.quant_table_port(q, bool(chroma)))


def test_finalize_host_entry_point():
    """uhdr_hip_generate_gainmap_finalize (jpegr.cpp:969-986, 1031-1048) is pure host code."""
    from libultrahdr_amd.stripes import finalize_minmax

    cfg = A.default_encode_cfg()
    mm, md = finalize_minmax(cfg, A.UHDR_CT_HLG, 0, [-1.5, -20.0, 0.25, 2.0, 3.0, 0.25])
    f = np.float32
    assert mm[0] == f(-1.5) and mm[1] == f(-14.3) and mm[3] == f(2.0)
    assert mm[2] == f(0.25) and mm[5] == f(0.25) + f(0.1)  # epsilon guard
    assert md.max_content_boost[0] == pytest.approx(4.0) and md.min_content_boost[1] == pytest.approx(2.0 ** -14.3, rel=1e-6)
    assert md.hdr_capacity_max == pytest.approx(1000.0 / 203.0) and md.use_base_cg == 0
    assert md.offset_sdr[0] == f(1e-7) and md.gamma[2] == 1.0
    cfg = A.default_encode_cfg(use_multi_channel_gainmap=0, max_content_boost=3.0, min_content_boost=1.0, target_disp_peak_nits=812.0, gamma=1.3)
    mm, md = finalize_minmax(cfg, A.UHDR_CT_PQ, 1, [-1.0, 9, 9, 5.0, 9, 9])
    assert mm[0] == 0.0 and mm[3] == f(np.log2(f(3.0)))
    assert list(md.max_content_boost) == [pytest.approx(3.0, rel=1e-6)] * 3 and md.hdr_capacity_max == pytest.approx(4.0)
    assert md.gamma[0] == f(1.3)


@pytest.mark.parametrize("ct", [A.UHDR_CT_HLG, A.UHDR_CT_PQ])
def test_oetf_code_thresholds_describe_the_reference_composite(ct):
    """The HLG / PQ decode tail runs on a 1023-entry threshold table built by the host layer.  It must
    reproduce the reference composite (oracle: powf + 65536-node LUT + 10-bit quantisation) exactly:
    at every threshold, just below every threshold, and on 4M random inputs (monotonicity)."""
    from oracle import loader as L

    lib = A.load()
    t = (C.c_float * 1024)()
    assert lib.uhdr_hip_oetf_code_thresholds(ct, t) == 0
    T = np.frombuffer(t, dtype=np.float32).copy()
    assert T[0] == 0.0 and np.all(np.diff(T[1:]) >= 0)

    def composite(v):
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.zeros(v.size, dtype=np.uint32)
        L.port().uo_oetf_code(ct, v.ctypes.data, out.ctypes.data, v.size)
        return out

    def by_table(v):
        return (np.searchsorted(T[1:], v, side="right")).astype(np.uint32)

    reach = T[1:][T[1:] <= 1.0]
    below = np.nextafter(reach, np.float32(-1.0)).astype(np.float32)
    rng = np.random.default_rng(0)
    dense = np.concatenate([reach, below[below >= 0], rng.random(2_000_000, dtype=np.float32),
                            (rng.random(2_000_000, dtype=np.float32) ** 8).astype(np.float32),  # emphasise near-black
                            np.array([0.0, 1.0], dtype=np.float32)])
    assert np.array_equal(composite(dense), by_table(dense))
    assert lib.uhdr_hip_oetf_code_thresholds(A.UHDR_CT_LINEAR, t) == -1


def test_public_headers_compile_as_plain_c_and_cpp(tmp_path):
    """include/uhdr_hip.h is C99 (the ABI a cgo / JNI / ctypes binding would consume); include/uhdr_hip.hpp is
    C++14.  Neither needs HIP, torch or the reference's headers."""
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "c99.c"
    c.write_text('#include "uhdr_hip.h"\nint main(void) { return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(c)], check=True)
    cpp = tmp_path / "cpp.cpp"
    cpp.write_text('#include "uhdr_hip.hpp"\nint main() { uhdr_hip::UltraHdr* p = nullptr; (void)p; return 0; }\n')
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(cpp)], check=True)


def test_jpeg_assemble_host_entry_point():
    """uhdr_hip_jpeg_assemble is pure host code: the file it writes around entropy-coded data equals the oracle's, byte
    for byte (4:2:0 with a restart interval, one component without), and bad descriptions are refused."""
    from oracle import loader as L

    lib = A.load()
    rng = np.random.default_rng(3)
    for (w, h, sampling, ri) in ((72, 40, [(2, 2), (1, 1), (1, 1)], 3), (37, 19, [(1, 1)], 0), (64, 32, [(1, 1)] * 3, 7)):
        hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
        coefs = []
        for hs, vs in sampling:
            cw, chh = -(-w * hs // hmax), -(-h * vs // vmax)
            a = (rng.normal(0, 20, (-(-chh // 8), -(-cw // 8), 64)) * (rng.random((-(-chh // 8), -(-cw // 8), 64)) < 0.3)).astype(np.int16)
            coefs.append(np.ascontiguousarray(a))
        ql, qc = L.quant_table_port(80, False), L.quant_table_port(80, True)
        scan = L.huffman_encode_port(coefs, w, h, sampling, ri)
        want = L.jpeg_assemble_port(coefs, w, h, sampling, ri, ql, qc, scan)
        sc = A.JpegScan()
        sc.num_components = len(coefs)
        for i, cf in enumerate(coefs):
            sc.blocks_h[i], sc.blocks_w[i] = cf.shape[0], cf.shape[1]
            sc.h_samp[i], sc.v_samp[i] = sampling[i]
        sc.w, sc.h, sc.restart_interval = w, h, ri
        src = np.frombuffer(scan, dtype=np.uint8)
        out = np.zeros(src.size + 2048, dtype=np.uint8)
        n = lib.uhdr_hip_jpeg_assemble(C.byref(sc), (C.c_uint16 * 64)(*ql.tolist()), (C.c_uint16 * 64)(*qc.tolist()), src.ctypes.data, src.size,
                                       out.ctypes.data, out.size)
        assert n == len(want) and out[:n].tobytes() == want
        assert lib.uhdr_hip_jpeg_assemble(C.byref(sc), (C.c_uint16 * 64)(*ql.tolist()), (C.c_uint16 * 64)(*qc.tolist()), src.ctypes.data, src.size,
                                          out.ctypes.data, 100) == 0  # capacity
        sc.blocks_w[0] += 5
        assert lib.uhdr_hip_jpeg_assemble(C.byref(sc), (C.c_uint16 * 64)(*ql.tolist()), (C.c_uint16 * 64)(*qc.tolist()), src.ctypes.data, src.size,
                                          out.ctypes.data, out.size) == 0  # block grid of another image


def test_jpeg_parse_survives_mutated_files():
    """Untrusted bytes: random mutations / truncations of a valid file never crash the parser, and whatever it accepts
    stays inside the buffer."""
    from oracle import loader as L

    lib = A.load()
    rng = np.random.default_rng(1)
    w, h, ri, sampling = 50, 30, 3, [(2, 2), (1, 1), (1, 1)]
    coefs = []
    for hs, vs in sampling:
        cw, ch = -(-w * hs // 2), -(-h * vs // 2)
        coefs.append(np.ascontiguousarray((rng.normal(0, 20, (-(-ch // 8), -(-cw // 8), 64)) * (rng.random((-(-ch // 8), -(-cw // 8), 64)) < 0.3)).astype(np.int16)))
    ql, qc = L.quant_table_port(75, False), L.quant_table_port(75, True)
    jpeg = L.jpeg_assemble_port(coefs, w, h, sampling, ri, ql, qc, L.huffman_encode_port(coefs, w, h, sampling, ri))
    base = np.frombuffer(jpeg, dtype=np.uint8)
    hdr = A.JpegHeader()
    accepted = 0
    for _ in range(20000):
        b = base.copy()
        k = int(rng.integers(1, 6))
        b[rng.integers(0, min(len(b), 700), k)] = rng.integers(0, 256, k)
        n = int(rng.integers(4, len(b) + 1)) if rng.random() < 0.3 else len(b)
        buf = (C.c_uint8 * n).from_buffer_copy(b[:n].tobytes())
        if lib.uhdr_hip_jpeg_parse(buf, n, C.byref(hdr)) == 0:
            accepted += 1
            assert hdr.scan_offset + hdr.scan_bytes <= n and hdr.scan.num_components in (1, 3)
            assert all(1 <= hdr.scan.h_samp[c] <= 2 and 1 <= hdr.scan.v_samp[c] <= 2 for c in range(hdr.scan.num_components))
    assert accepted > 0
