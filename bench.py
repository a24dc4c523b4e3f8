#!/usr/bin/env python
"""bench.py -- Mpixels/s of the gain-map hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): decode of 4K (3840x2160) UltraHDR frames -- applyGainMap from
a YCbCr 4:2:0 base image + 8-bit gain map to linear RGBA_F16 -- with all inputs resident in HBM.
One "step" = one pass of the hot path over a batch of BATCH distinct frames per rank (distinct
buffers so the 256 MiB Infinity Cache cannot hold the working set).  Frames are independent, so N
ranks shard by frame with no data-path collective (weak scaling); `value` = pixels all ranks
processed / max-over-ranks time.

Extra, rank 0 at N=1 only, outside the timed region: the same kernel at 8K (the north-star
roofline target), the HLG/PQ outputs, the encode-side kernels, and the CPU baseline (the real
reference from oracle/_ref when it loads, else the C port) on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="4K frames per rank per step")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--map", choices=["A", "B", "C"], default="C",
                    help="gain map of the decoded stream.  C: RGBA8888 full resolution (13.5 B/px) -- what the reference's "
                         "C-API default encode (3 channels, scale 1) decodes to with its pinned libjpeg-turbo; "
                         "B: the same as RGB888 (IJG libjpeg layout, 12.5 B/px); A: Y400 at scale 4 (Android default, 9.5625 B/px)")
    ap.add_argument("--launch", choices=["batch", "single"], default="batch",
                    help="batch: one kernel launch per step for the whole frame batch (uhdr_hip_apply_gainmap_batch_dev); "
                         "single: one launch per frame")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="skip the row-striped two-pass encode leg (the RCCL collective)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


MAP_DESC = {"A": "Y400 gain map (scale 4)", "B": "RGB888 gain map (scale 1)", "C": "RGBA8888 gain map (scale 1)"}


def algo_bytes_per_px(map_kind, out_bytes=8):
    # every input read once, every output written once
    return 1.5 + {"A": 1.0 / 16.0, "B": 3.0, "C": 4.0}[map_kind] + out_bytes


def make_frames(n, w, h, map_kind, device, out_fmt, seed0=1234):
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from libultrahdr_amd.images import Image

    frames = []
    for i in range(n):
        sdr = synth.make_sdr_yuv420(w, h, seed=seed0 + i)
        if map_kind == "A":
            gm = synth.make_gainmap(w // 4, h // 4, 1, seed=seed0 + 100 + i)
        else:
            gm = synth.make_gainmap(w, h, 3, alpha=(map_kind == "C"), seed=seed0 + 100 + i)
        dest = Image(out_fmt, w, h, align=64, device=device)
        frames.append((sdr.to(device), gm.to(device), dest))
    return frames


CLOCK_RAMP_S = float(os.environ.get("UHDR_BENCH_CLOCK_RAMP_S", "0.7"))  # tools/profile_bench.sh shortens it under the profiler
CLOCK_RAMP_MAX_S = float(os.environ.get("UHDR_BENCH_CLOCK_RAMP_MAX_S", "2.0"))  # the fixed cap of the adaptive ramp (round 4: was 4 s)


def clock_ramp(ctx, fn, seconds=CLOCK_RAMP_S, adaptive=False):
    """Untimed: keep the device busy with fn() for `seconds` before a measurement.  The part idles at 1.4 GHz and needs about
    half a second of continuous load to reach its 2.4 GHz (profiles/r03_clock_ramp.txt: sysfs clocks every 20 ms); every
    section of this bench starts after host-side work (frame synthesis, allocation), i.e. from an idle device, and a
    30-launch region lasts 2.5 ms -- without the ramp it reads 25-35 % slow (8K map A: 77-89 us against 59-61 us once the clock
    is up, tools/kbench with KB_N = 30 / 3000).  A decode service under load sits at the ramped clock.
    The ramp does not always begin at once (the SMU may sit in a lower state for a few hundred ms first; a default run on a
    freshly acquired box read 7 % slow throughout its 20 timed steps while the same box gave the usual figure minutes later),
    so with adaptive=True (the headline and the 8K north-star sections) the loop goes on after `seconds` in windows of 0.1 s
    until the time per call has stopped falling -- two windows in a row within 0.3 % of their predecessor -- or
    CLOCK_RAMP_MAX_S have passed."""
    t0 = time.perf_counter()
    n = 0
    win_t0, win_n, prev, stable = t0, 0, None, 0
    while True:
        for _ in range(20):
            fn()
        ctx.synchronize()
        n += 20
        win_n += 20
        now = time.perf_counter()
        if now - win_t0 >= 0.1:
            per = (now - win_t0) / win_n
            stable = stable + 1 if (prev is not None and abs(per - prev) <= 0.003 * prev) else 0
            prev, win_t0, win_n = per, now, 0
        if not adaptive or seconds <= 0.1:  # (the profiler passes ask for a token ramp)
            if now - t0 >= seconds:
                break
        elif (now - t0 >= seconds and stable >= 2) or now - t0 >= CLOCK_RAMP_MAX_S:
            break
    return n


def time_kernel(ctx, fn, iters=10, warm=3):
    """Average wall time per call of fn() measured with HIP events on the context's stream
    (uhdr_hip_profile_* wraps every launch of the family in an event pair)."""
    clock_ramp(ctx, fn)
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    for _ in range(iters):
        fn()
    n, ms = ctx.profile_read(None, reset=True)
    ctx.profile(False)
    return ms / max(n, 1) * (n / iters)  # ms per fn() call (a call may launch >1 kernel)


def time_region(ctx, fn, iters=30, warm=6, reps=3, adaptive_ramp=False):
    """ms per fn() call from ONE pair of HIP events around `iters` back-to-back calls on the library's stream (median of
    `reps` regions): the sustained per-launch rate, boundaries between consecutive launches included, with the part at
    its full clock (clock_ramp: a 30-launch region after an idle gap reads 25-35 % slow, which is what rounds 1 and 2 took
    for a "sustained vs burst" effect)."""
    import torch

    _, ext = ctx._streams()
    clock_ramp(ctx, fn, adaptive=adaptive_ramp)
    for _ in range(warm):
        fn()
    ctx.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ext)
        for _ in range(iters):
            fn()
        e1.record(ext)
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / iters)
    out.sort()
    return out[len(out) // 2]


METRIC = "Mpixels/s encode+decode (API-1 P010+YUV420, 4K), device-resident round trip"
DETAIL_NAME = "bench_detail.json"
COMPACT_LIMIT = 4096  # bytes: the driver keeps a 16 KB tail of stdout; the round-5 line (24 KB) was cut and could not be parsed


def make_roundtrip(ctx, u, device, w, h, seed=1234):
    """One API-1 round trip on device-resident images, as two closures (bytes <-> pixels, what BASELINE.json's metric names):
      enc(): uhdr_hip_encode_api1_scans_dev -- the fused chain (two-pass 3-channel gain map at scale 1, convertYuv, every FDCT) + the
             file's two scans Huffman-coded without restart markers (the reference's bytes) -> two entropy-coded scans in HBM
             (JpegR::encodeJPEGR API-1, jpegr.cpp:253-316, without the container's host byte shuffling);
      dec(): uhdr_hip_decode_api1_scans_dev -- the two scans Huffman-decoded + the map's dequant / IDCT / ycc->rgb + applyGainMap with
             the base image's dequant + IDCT inside the kernel -> RGBA_F16 linear (JpegR::decodeJPEGR, jpegr.cpp:1469-1531, after parsing).
    `two=False`: the round-5 form -- one C call per stage, the scans coded one after the other."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from libultrahdr_amd.images import Image
    from libultrahdr_amd.ultrahdr import UltraHdr
    import torch

    px = w * h
    qy, qc = u.quant_table(95, False), u.quant_table(95, True)
    enc1 = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY)
    f16, rgba = A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA8888
    S420, S444 = [(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3
    sdr = synth.make_sdr_yuv420(w, h, seed=seed).to(device)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG, seed=seed).to(device)
    out_b = torch.empty(px * 2, dtype=torch.uint8, device=device)
    out_m = torch.empty(px * 4, dtype=torch.uint8, device=device)
    box = {"keep": (sdr, hdr, out_b, out_m)}

    hb = u.jpeg_header(w, h, S420, [qy, qc, qc])
    hm = u.jpeg_header(w, h, S444, [qy, qc, qc])

    qts_enc = (qy, qc)

    # (arguments marshalled once: UltraHdr.bindEncodeApi1Scans / bindDecodeApi1Scans -- a per-frame caller's steady state)
    enc_run = enc1.bindEncodeApi1Scans(sdr, hdr, A.UHDR_CG_DISPLAY_P3, qts_enc, qts_enc, out_b, out_m)

    def enc(two=True):
        if two:  # ONE C call: the fused chain + both scans coded concurrently (uhdr_hip_encode_api1_scans_dev)
            box["nb"], box["nm"], box["md"] = enc_run()
            return
        cb, cm, md_, _ = enc1.encodeApi1Fused(sdr, hdr, A.UHDR_CG_DISPLAY_P3, (qy, qc), (qy, qc), want_map=False)
        box["nb"] = int(u.huffman_encode(cb, w, h, S420, 0, out=out_b).numel())
        box["nm"] = int(u.huffman_encode(cm, w, h, S444, 0, out=out_m).numel())
        box["md"] = md_

    enc()
    ctx.synchronize()
    box["sb"], box["sm"] = out_b[: box["nb"]].clone(), out_m[: box["nm"]].clone()
    box["shp_b"] = [(h // 8, w // 8), (h // 16, w // 16), (h // 16, w // 16)]
    box["shp_m"] = [(h // 8, w // 8)] * 3
    gm3 = Image(rgba, w, h, A.UHDR_CG_BT_2100, align=64, device=device)
    dst = Image(f16, w, h, align=64, device=device)
    box["dst"] = dst
    qts = [qy, qc, qc]

    dec_run = u.bindDecodeApi1Scans(hb, box["sb"], A.UHDR_CG_BT_709, hm, box["sm"], A.UHDR_CG_BT_2100, box["md"], A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst)

    def dec(two=True):
        sb, sm = box["sb"], box["sm"]
        if two:  # ONE C call: both scans decoded concurrently + the map's IDCT + applyGainMap from coefficients (uhdr_hip_decode_api1_scans_dev)
            dec_run()
            return
        cb = u.huffman_decode(sb, box["shp_b"], w, h, S420, 0)
        cm = u.huffman_decode(sm, box["shp_m"], w, h, S444, 0)
        u.idct_dequant_rgb(cm, qy, qc, w, h, rgba, 0, dst=gm3)
        u.applyGainMapFromCoefficients(cb, qts, w, h, A.UHDR_CG_BT_709, gm3, box["md"], A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst)

    return enc, dec, box


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # UHDR_BENCH_DIST_BACKEND=gloo lets the N>1 code path be smoke-tested on a box with fewer GPUs than
    # ranks (ranks then share devices; timing of such a run is meaningless).  Default: RCCL, one GPU per rank.
    backend = os.environ.get("UHDR_BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    device = f"cuda:{dev_index}"

    from libultrahdr_amd import capi as A
    from libultrahdr_amd.ultrahdr import Context, UltraHdr

    ctx = Context(dev_index)
    u = UltraHdr(ctx=ctx)
    w, h = args.width, args.height
    headline_only = bool(os.environ.get("UHDR_BENCH_HEADLINE_ONLY"))

    # ---- the headline: BASELINE.json's metric as worded.  One step = one API-1 round trip of this rank's 4K frame, inputs
    # resident in HBM: P010 + YCbCr 4:2:0 -> two entropy-coded JPEG scans -> RGBA_F16 linear.  Frames are independent: N ranks
    # shard by frame with no data-path collective (weak scaling); value = pixels of all ranks / max-over-ranks time.
    enc, dec, box = make_roundtrip(ctx, u, device, w, h, seed=1234 + 1000 * rank)

    def step():
        enc()
        dec()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    # the timed loops run on the library's stream alone (every input was made and synchronised above): no torch <-> library event pair
    # around each call (Context's stream contract: "stream_safe=False ... for timing loops that must not record extra events")
    ctx.stream_safe = False
    st0 = A.Stats()
    ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st0))
    ramp_t0 = time.perf_counter()
    ramp_steps = clock_ramp(ctx, step, adaptive=True)  # untimed, before the W warm-up steps: see clock_ramp
    ramp_seconds = time.perf_counter() - ramp_t0
    for _ in range(args.warmup):
        step()
    barrier()
    # timed region: exactly K steps between two barriers, nothing else in it
    fams = ["generate_gainmap", "fdct_quant", "huffman_encode", "huffman_decode", "idct_dequant", "apply_gainmap"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    # ... and, OUTSIDE it, the same steps once more with the library's per-family HIP-event pairs on (two hipEventCreate + two hipEventRecord per
    # family scope, ~16 scopes per round trip: through round 6's first lines this rode along INSIDE the timed region and cost it ~70 us per step)
    fam_steps = max(1, min(args.steps, 10))
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    for _ in range(fam_steps):
        step()
    barrier()
    fam_us = {}
    for f in fams:
        n_f, ms_f = ctx.profile_read(f, reset=True)
        if n_f:
            fam_us[f] = {"us": round(ms_f / fam_steps * 1e3, 2), "launches": n_f // fam_steps}
    ctx.profile_read(None, reset=True)
    ctx.profile(False)
    st1 = A.Stats()
    ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st1))
    ctx.stream_safe = True
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = w * h * world * args.steps / elapsed / 1e6

    full = {
        "metric": METRIC,
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"API-1 {w}x{h} round trip per step: encode (P010 HLG + YCbCr420 -> two-pass 3-ch gain map at scale 1, FDCT q95, "
                        "Huffman: two JPEG scans) + decode (Huffman, IDCT, applyGainMap -> RGBA_F16 linear); frames resident in HBM",
            "frames_per_rank_per_step": 1,
            "clock_ramp": f"{ramp_steps} untimed steps ({ramp_seconds:.1f} s) before the {args.warmup} warm-up steps (profiles/r03_clock_ramp.txt)",
            "sharding": f"frames x{world} ranks, no data-path collective",
        },
        "step": {"kernel_families_us_per_step": fam_us, "kernel_families_note": f"{fam_steps} more steps after the timed region, the library's per-family event pairs on",
                 "scan_bytes_base": box["nb"], "scan_bytes_map": box["nm"],
                 "decode_route": {"parallel": int(st1.entropy_decode_parallel - st0.entropy_decode_parallel),
                                  "single_lane": int(st1.entropy_decode_single_lane - st0.entropy_decode_single_lane),
                                  "declined": int(st1.entropy_decode_declined - st0.entropy_decode_declined)}},
        "roofline": {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
    }
    del enc, dec, box, step
    torch.cuda.empty_cache()

    # ---- roofline: the dominant HBM kernel of the path -- applyGainMap -> RGBA_F16 -- at the north-star size (ONE 7680x4320
    # frame per launch, RGBA8888 map at scale 1), HIP events around every launch on the library's stream, live in this run (rank 0)
    if rank == 0:
        try:
            full["roofline"] = roofline_section(ctx, device, w, h, light=headline_only)
        except Exception as e:  # noqa: BLE001
            full["roofline"]["error"] = f"{type(e).__name__}: {e}"

    # BASELINE configs[3]: the row-striped API-1 two-pass encode, the one place where the path has a collective.  Every
    # rank runs it (also at N = 1: a one-rank communicator), after the headline's timed region.
    if not args.no_config4 and not headline_only:
        # The headline is measured; nothing after it may cost the line.  An exception is caught below, but a collective that
        # never returns (a communicator that cannot form on some node) cannot be: a watchdog prints the line and leaves.
        dog = arm_watchdog(180.0 if world > 1 else 600.0, full, rank, "config4")
        try:
            c4 = config4_section(ctx, u, device, rank, world, backend)
        except Exception as e:  # noqa: BLE001
            c4 = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
        if rank == 0:
            full["config4"] = c4

    # the stage measurements and the CPU baseline run AFTER the timed region; a failure there (e.g. an
    # out-of-memory on a smaller device) must not cost the headline line
    if rank == 0 and world == 1 and not args.no_extra and not headline_only:
        for key, fn in (("api1_roundtrip", lambda: api1_roundtrip_section(ctx, u, device)),
                        ("api1_concurrent", lambda: concurrent_section(device, dev_index)),
                        ("headline_16x4k", lambda: batch16_section(ctx, u, device, args)),
                        ("encode", lambda: encode_section(ctx, u, device)), ("config5", lambda: config5_section(ctx, u, device)),
                        ("extra", lambda: extras(ctx, u, device)), ("api_level", lambda: api_level_section())):
            try:
                full[key] = fn()
            except Exception as e:  # noqa: BLE001  (a failure in one section must not cost the headline line)
                full[key] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            full["cpu_baseline"] = cpu_baseline(w, h, args.cpu_seconds, stages=not args.no_extra)
        except Exception as e:  # noqa: BLE001
            full["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        emit(full)
    if world > 1:  # the line is out: a teardown that hangs must not keep the launcher waiting
        arm_watchdog(60.0, None, rank, "teardown")
    ctx.close()
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    if world > 1:
        dist.destroy_process_group()


def dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


# the scalars the compact line carries inside `roofline` beyond the contract's keys: (name on the line, path in the full record)
ROOFLINE_SCALARS = (
    ("api1_4k_enc_us", ("api1_roundtrip", "api1_4k_enc_us")), ("api1_4k_dec_us", ("api1_roundtrip", "api1_4k_dec_us")),
    ("api1_8k_enc_us", ("api1_roundtrip", "api1_8k_enc_us")), ("api1_8k_dec_us", ("api1_roundtrip", "api1_8k_dec_us")),
    ("api1_8k_roundtrip_Mpxs", ("api1_roundtrip", "api1_8k_roundtrip_Mpxs")),
    ("api1_4k_x4_in_flight_Mpxs", ("api1_concurrent", "frames_in_flight_4_Mpxs")),
    ("ns8k_mapA_cold_frac", ("roofline", "ns8k_mapA_cold_frac")), ("ns8k_mapB_frac", ("roofline", "ns8k_mapB_frac")),
    ("ns8k_mapC_frac", ("roofline", "ns8k_mapC_frac")),
    ("config5_frac", ("config5", "frac_of_8TBs")), ("headline_16x4k_frac", ("headline_16x4k", "frac")),
    ("config4_ms_per_image", ("config4", "ms_per_image")), ("config4_all_reduce_us", ("config4", "all_reduce_us_back_to_back")),
    ("config4_full_16k_ms", ("config4", "full_16k_x_16k_one_gpu", "ms_per_image")),
    ("uhdr_encode_api0_8k_ms", ("api_level", "uhdr_encode_api0_8k_hip", "ms")),
)
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_us", "launches_timed")


def compact_line(full, detail_path=None):
    """The ONE line the driver parses (<= COMPACT_LIMIT bytes): the contract's keys, `roofline` for the dominant HBM kernel with a
    dozen scalars of the other sections, `cpu_baseline`.  Everything else is in the detail file."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in full}
    cfg = full.get("config") or {}
    out["config"] = {k: (v if not isinstance(v, str) else v[:400]) for k, v in cfg.items() if k in ("workload", "frames_per_rank_per_step", "sharding", "clock_ramp")}
    r = full.get("roofline") or {}
    roof = {k: r.get(k) for k in ROOFLINE_KEYS}
    for name, path in ROOFLINE_SCALARS:
        v = dig(full, *path)
        if isinstance(v, (int, float)):
            roof[name] = v
    if isinstance(r.get("error"), str):
        roof["error"] = r["error"][:200]
    out["roofline"] = roof
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (cb.get(k) if not isinstance(cb.get(k), str) else cb.get(k)[:300]) for k in ("value", "unit", "cores", "kind", "sample")}
    errs = [k for k, v in full.items() if isinstance(v, dict) and isinstance(v.get("error"), str)]
    if errs:
        out["sections_with_errors"] = errs[:8]
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:  # cannot happen with the bounded fields above; never let a long string cost the record
        for k in ("clock_ramp", "sharding"):
            out["config"].pop(k, None)
        out.pop("sections_with_errors", None)
        if "cpu_baseline" in out and isinstance(out["cpu_baseline"].get("sample"), str):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:120]
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def emit(full):
    """Write the full record next to the script (and into gpurun_out/ when that exists, so that a gpurun call brings it back),
    say where it is on an EARLIER line, then print the compact line as the LAST line of stdout."""
    paths = [os.path.join(ROOT, DETAIL_NAME)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", DETAIL_NAME))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            written = written or p
        except OSError:
            pass
    # RCCL reports its version through C stdio; drain that first so that the JSON line is the LAST line on stdout
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    if written:
        print(f"bench detail: {written}", flush=True)
    print(compact_line(full, DETAIL_NAME if written else None), flush=True)


def roofline_section(ctx, device, w, h, light=False):
    """`roofline` of the bench line: applyGainMap of ONE 7680x4320 frame per launch -> RGBA_F16 with the RGBA8888 map at scale 1 (the
    north-star configuration; 13.5 algorithmic B/px), per-launch HIP events on the library's stream over 60
    launches after the clock ramp, three rotating buffer sets.  The other map layouts and the on-box copy ceiling ride along."""
    ns = north_star_8k(ctx, device, kinds=(("mapC", "C", 3),) if light else None)
    c = ns["mapC"]
    each = c.pop("launch_ms_each")
    # the launch duration: HIP events on the library's stream around 30 back-to-back launches, median of 5 such regions (150 launches) -- an
    # event pair around EVERY launch adds ~3 us to an 80 us kernel (launch_us below keeps those per-launch figures for their spread)
    avg_s = c["sustained_us"] * 1e-6
    algo_b = c["algorithmic_bytes_per_launch"]
    kernel_name = "apply_quad_kernel<F16,RGBA8888,scale1>"
    traffic, traffic_src = measured_traffic(f"{kernel_name}|1x7680x4320")
    r = {"bound": "hbm", "kernel": kernel_name + " 7680x4320, one frame per launch", "achieved": round(algo_b / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(algo_b / avg_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
         "algorithmic_bytes": int(algo_b), "avg_launch_us": round(avg_s * 1e6, 3), "launches_timed": 150,
         "launch_us": launch_stats(each), "per_launch_event_pairs_avg_us": round(sum(each) / len(each) * 1e3, 2), "north_star_8k": ns}
    for key, short in (("mapC", "mapC"), ("mapB", "mapB"), ("mapA", "mapA_hot"), ("mapA_cold_inputs", "mapA_cold")):
        e = ns.get(key)
        if isinstance(e, dict) and "frac" in e:
            e.pop("launch_ms_each", None)
            r[f"ns8k_{short}_frac"] = e["frac"]
            r[f"ns8k_{short}_us"] = e["sustained_us"]
    if not light:
        r.update(onbox_ceiling(device))
    return r


def batch16_section(ctx, u, device, args):
    """Rounds 1-5's headline, kept as a section: applyGainMap of 16 distinct resident 4K frames per launch -> RGBA_F16
    (uhdr_hip_apply_gainmap_batch_dev), map C; per-launch HIP events; plus the same launch from an idle clock."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    import torch

    w, h, nb = 3840, 2160, args.batch
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    md = synth.default_metadata(use_base_cg=0)  # BT.709 base, BT.2100 gain-map space: the SDR-side 3x3 is active
    frames = make_frames(nb, w, h, args.map, device, f16, seed0=1234)
    for s_, g_, _ in frames:
        s_.raw.cg, g_.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    arr_s = (A.RawImage * nb)(*[s.raw for s, _, _ in frames])
    arr_g = (A.RawImage * nb)(*[g.raw for _, g, _ in frames])
    arr_d = (A.RawImage * nb)(*[d.raw for _, _, d in frames])

    def step():
        st = ctx.lib.uhdr_hip_apply_gainmap_batch_dev(ctx.handle, nb, arr_s, arr_g, C.byref(md), A.UHDR_CT_LINEAR, f16, A.FLT_MAX, arr_d)
        if st.error_code != 0:
            raise RuntimeError(st.detail)

    step()
    ctx.synchronize()
    time.sleep(2.0)  # the OTHER clock regime: the first 20 launches after 2 s of idle, no ramp, no warm-up
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    for _ in range(20):
        step()
    ctx.synchronize()
    cold = ctx.profile_read_list("apply_gainmap", reset=True)
    ctx.profile(False)
    clock_ramp(ctx, step, adaptive=True)
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    for _ in range(40):
        step()
    each = ctx.profile_read_list("apply_gainmap", reset=True)
    ctx.profile(False)
    fpl = (40 * nb) // max(len(each), 1)
    algo_b = algo_bytes_per_px(args.map) * w * h * fpl
    avg_s = sum(each) / len(each) / 1e3
    kernel_name = "apply_quad_kernel<F16," + {"A": "Y400,scale4>", "B": "RGB888,scale1>", "C": "RGBA8888,scale1>"}[args.map]
    traffic, traffic_src = measured_traffic(f"{kernel_name}|{fpl}x{w}x{h}")
    res = {"workload": f"{nb} x {w}x{h} YCbCr420 + " + MAP_DESC[args.map] + " -> RGBA_F16, one batched launch", "frames_per_launch": fpl,
           "avg_launch_us": round(avg_s * 1e6, 2), "frac": round(algo_b / avg_s / 1e9 / HBM_PEAK_GBS, 4), "GB/s": round(algo_b / avg_s / 1e9, 1),
           "Mpx/s": round(fpl * w * h / avg_s / 1e6, 1), "algorithmic_bytes_per_launch": int(algo_b), "traffic": traffic, "traffic_source": traffic_src,
           "launch_us": launch_stats(each)}
    if cold:
        cold_s = sum(cold) / len(cold) / 1e3
        res["cold_start_frac"] = round(algo_b / cold_s / 1e9 / HBM_PEAK_GBS, 4)
        res["cold_start_avg_launch_us"] = round(cold_s * 1e6, 2)
        res["cold_start_note"] = "the first 20 launches after 2 s of idle, no clock ramp, no warm-up (sclk starts at its 1.4 GHz idle state)"
    del frames
    torch.cuda.empty_cache()
    return res


def onbox_ceiling(device):
    """What this box's HBM delivers to a plain device-to-device copy right now (torch's copy kernel, 1 GiB read + 1 GiB written
    per call, beyond the 256 MiB infinity cache): the practical ceiling the roofline fraction should be read against."""
    import torch

    n = 1 << 30
    a = torch.empty(n // 4, dtype=torch.float32, device=device)  # float32: torch's copy kernel moves 16 bytes per lane
    b = torch.empty(n // 4, dtype=torch.float32, device=device)
    a.fill_(7)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        best = ms if best is None or ms < best else best
    del a, b
    torch.cuda.empty_cache()
    gbs = 2 * n / (best / 1e3) / 1e9
    return {"onbox_copy_GBs": round(gbs, 1), "onbox_copy_frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
            "onbox_copy_note": "torch device-to-device copy of 1 GiB (bytes read + bytes written) / time, best of 3 x 10 calls, in this run"}


def north_star_8k(ctx, device, kinds=None):
    """BASELINE north star: applyGainMap of ONE 7680x4320 frame per launch -> RGBA_F16, frames rotating through enough buffer
    sets that neither inputs nor outputs stay in the 256 MiB infinity cache.  Per map kind: the sustained time per launch
    (one HIP-event pair around 30 back-to-back launches, median of 5 regions) and the spread of the individual launches."""
    import torch

    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    res = {}
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    md = synth.default_metadata(use_base_cg=0)
    w, h = 7680, 4320
    for key, kind, nsets in (kinds or (("mapC", "C", 3), ("mapB", "B", 3), ("mapA", "A", 3), ("mapA_cold_inputs", "A", 6))):
        sets = make_frames(nsets, w, h, kind, device, f16, seed0=4242)
        for s_, g_, _ in sets:
            s_.raw.cg, g_.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
        argv = [(C.byref(s_.raw), C.byref(g_.raw), C.byref(md), C.byref(d_.raw)) for s_, g_, d_ in sets]
        k = [0]

        def fn():
            a = argv[k[0] % nsets]
            k[0] += 1
            st = ctx.lib.uhdr_hip_apply_gainmap_dev(ctx.handle, a[0], a[1], a[2], A.UHDR_CT_LINEAR, f16, A.FLT_MAX, a[3], 0, 0)
            if st.error_code != 0:
                raise RuntimeError(st.detail)

        ms = time_region(ctx, fn, iters=30, warm=10, reps=5, adaptive_ramp=True)
        ctx.profile(True)
        ctx.profile_read(None, reset=True)
        for _ in range(60):
            fn()
        each = ctx.profile_read_list("apply_gainmap", reset=True)
        ctx.profile(False)
        b = algo_bytes_per_px(kind) * w * h
        st = launch_stats(each)
        res[key] = {"map": MAP_DESC[kind], "algorithmic_bytes_per_launch": int(b), "sustained_us": round(ms * 1e3, 2),
                             "frac": round(b / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4), "GB/s": round(b / (ms / 1e3) / 1e9, 1),
                             "launch_us": st, "p90_over_p10": round(st["p90"] / st["p10"], 3) if st else None, "buffer_sets": nsets,
                             "launch_ms_each": each}
        del sets, argv
        torch.cuda.empty_cache()
    res["timing"] = ("sustained_us: one HIP-event pair around 30 back-to-back launches, median of 5 regions; launch_us: per-launch HIP events "
                     "over 60 more launches")
    res["buffer_sets_note"] = ("three rotating buffer sets (the rotation of rounds 1 and 2): 546 MB of map-C inputs per rotation never stay in the 256 MiB "
                               "infinity cache, but map A's 51 MB per set (153 MB per rotation) do -- mapA_cold_inputs rotates six sets so that they come "
                               "from HBM as well; its 17-23 us of extra time are the access pattern's (narrow reads inside a saturated write stream, "
                               "tools/ubench8, profiles/r03_factor_study.txt), the arithmetic is the same")
    return res


def arm_watchdog(seconds, out, rank, section):
    """Daemon timer: if it fires, rank 0 prints the bench line as it stands (with an error note for `section`) and every
    rank leaves the process.  Cancel it when the section returns."""
    import threading

    def fire():
        if rank == 0 and out is not None:
            out.setdefault(section, {"error": f"section did not return within {seconds:.0f} s (watchdog); the lines above it are complete"})
            try:
                C.CDLL(None).fflush(None)
            except Exception:  # noqa: BLE001
                pass
            emit(out)
        os._exit(0)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def launch_stats(ms_list):
    """min / median / max (and the 10th / 90th percentile) of a list of per-launch durations in ms -> us."""
    if not ms_list:
        return None
    v = sorted(ms_list)
    q = lambda f: v[min(len(v) - 1, int(f * (len(v) - 1) + 0.5))]
    return {"min": round(v[0] * 1e3, 2), "p10": round(q(0.1) * 1e3, 2), "median": round(q(0.5) * 1e3, 2),
            "p90": round(q(0.9) * 1e3, 2), "max": round(v[-1] * 1e3, 2), "n": len(v)}


def family_times(ctx, fn, families, iters=5, warm=2):
    """Run fn() `iters` times with the library's per-launch HIP events on; -> {family: us per fn() call}."""
    clock_ramp(ctx, fn)
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    for _ in range(iters):
        fn()
    out = {}
    for f in families:
        n, ms = ctx.profile_read(f, reset=True)
        if n:
            out[f] = {"us": round(ms / iters * 1e3, 2), "launches": n // iters}
    ctx.profile_read(None, reset=True)
    ctx.profile(False)
    return out


def library_sha256():
    import hashlib

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libultrahdr_amd", "lib", "libuhdr_hip.so")
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def measured_traffic(key):
    """HBM bytes per launch of the dominant kernel as measured with rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE (their own passes, MI355X_MICROARCH.md's gfx950 correction applied) on this exact bench
    configuration; committed next to the rocprof summaries in profiles/traffic.json (tools/update_traffic.py writes it from a
    tools/profile_bench.sh run).  The entry carries the SHA-256 of the libuhdr_hip.so it was measured with (round 4; the
    version string did not change when a kernel did): any other binary gets null instead of a stale figure."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    try:
        with open(path) as f:
            e = json.load(f).get(key)
    except (OSError, ValueError):
        e = None
    if not e:
        return None, None
    sha = library_sha256()
    if e.get("library_sha256") != sha:
        return None, f"profiles/traffic.json was measured with another libuhdr_hip.so ({str(e.get('library_sha256'))[:12]}..., this is {str(sha)[:12]}...): re-run tools/profile_bench.sh + tools/update_traffic.py"
    return e["traffic_bytes_per_launch"], e["source"]


def encode_section(ctx, u, device):
    """The ENCODE half of BASELINE's metric, with its own roofline: per-kernel HIP-event durations inside the stage
    chains and the algorithmic bytes of SURVEY.md 8(d) for the kernels actually launched.
      config3: API-0 encode of an 8K RGBA1010102 PQ frame (tone map, one-pass max-RGB 3-channel map at scale 1,
               RGB -> YCbCr 4:4:4, FDCT + quantize of the base's and the map's three planes)
      api1_4k: API-1 encode of a 4K P010 + YCbCr 4:2:0 pair (two-pass 3-channel map, convertYuv, FDCTs)
    Entropy coding is a separate stage (extra.huffman_*)."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from libultrahdr_amd.images import Image
    from libultrahdr_amd.ultrahdr import UltraHdr
    import numpy as np
    import torch

    res = {}
    fams = ["tone_map", "generate_gainmap", "convert_raw_input_to_ycbcr", "convert_yuv", "fdct_quant", "jpeg_color", "encode_api0_fused"]
    qy, qc = u.quant_table(95, False), u.quant_table(95, True)

    def fdct_planes(img, planes, tables):
        for c in planes:
            rows, stride, wv = img.layout[c]
            u.fdct_quant(img.plane_tensor(c), stride, wv // 8, rows // 8, tables[c])

    def roof(px, kernels, bytes_per_px, traffic_key=None):
        tot_us = sum(k["us"] for k in kernels.values())
        b = sum(bytes_per_px.values()) * px
        # HBM bytes of the chain's kernels from the counter passes of a profiled run of THIS library (profiles/traffic.json, round 5)
        tr, tr_src = measured_traffic(traffic_key) if traffic_key else (None, None)
        for name, k in kernels.items():
            k["bytes_per_px"] = bytes_per_px.get(name)
            if bytes_per_px.get(name):
                k["GB/s"] = round(bytes_per_px[name] * px / (k["us"] * 1e-6) / 1e9, 1)
                k["frac"] = round(k["GB/s"] / HBM_PEAK_GBS, 4)
        return {"bound": "hbm", "achieved": round(b / (tot_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(b / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(b), "chain_us": round(tot_us, 1),
                "traffic": tr, "traffic_source": tr_src, "kernels": kernels}

    # ---- BASELINE config 3: API-0, 8K RGBA1010102 PQ ------------------------------------------------------------------
    w8, h8 = 7680, 4320
    px8 = w8 * h8
    hdr8 = synth.make_hdr_rgba1010102(w8, h8, ct=A.UHDR_CT_PQ).to(device)
    sdr8 = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w8, h8, align=64, device=device)
    enc0 = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_REALTIME)

    def api0():
        u.toneMap(hdr8, sdr8)
        _, gm_ = enc0.generateGainMap(sdr8, hdr8, False, False)
        fdct_planes(u.convert_raw_input_to_ycbcr(sdr8, False), (0, 1, 2), (qy, qc, qc))
        fdct_planes(u.jpeg_rgb_to_ycc(gm_), (0, 1, 2), (qy, qc, qc))

    k = family_times(ctx, api0, fams, iters=3, warm=1)
    r = roof(px8, k, {"tone_map": 8.0, "generate_gainmap": 11.0, "convert_raw_input_to_ycbcr": 7.0, "fdct_quant": 18.0, "jpeg_color": 6.0})
    res["config3_api0_8k_reference_operators"] = {
        "workload": "configs[2]: 7680x4320 RGBA1010102 PQ -> toneMap + generateGainMap (1 pass, max-RGB, 3 ch, scale 1) + "
                    "convert_raw_input_to_ycbcr(4:4:4) + jpeg rgb->ycc of the map + 6 x fdct_quant; entropy coding excluded",
        "us": r["chain_us"], "Mpx/s": round(px8 / r["chain_us"], 1), "roofline": r}

    def api0_fused():
        _, ycc_, _, gm_ = enc0.encodeApi0Fused(hdr8, want_sdr_rgba=False, use_luminance=False)
        fdct_planes(ycc_, (0, 1, 2), (qy, qc, qc))
        u.fdct_quant_rgb(gm_, qy, qc)

    k = family_times(ctx, api0_fused, fams, iters=3, warm=1)
    # fused front end: 4 in, 3 + 3 out; base FDCT 3 + 6; fused map rgb->ycc + FDCT 3 + 6 (both under "fdct_quant")
    r = roof(px8, k, {"encode_api0_fused": 10.0, "fdct_quant": 18.0}, "encode.config3_api0_8k")
    if r.get("traffic") is not None:
        r["traffic_note"] = "the fused front-end kernel only (10 B/px of the 28): the six FDCTs behind it were not in the profiled case"
    res["config3_api0_8k"] = {
        "workload": "configs[2] with the MI355X-first fusion: one front-end kernel (tone map + gain map + YCbCr 4:4:4) + "
                    "3 x fdct_quant + fused (rgb->ycc + 3 x fdct_quant) of the map; same bytes out as the reference operators",
        "us": r["chain_us"], "Mpx/s": round(px8 / r["chain_us"], 1), "fused_floor_16B_per_px_us": round(16.0 * px8 / (HBM_PEAK_GBS * 1e3), 1),
        "roofline": r}
    del hdr8, sdr8
    torch.cuda.empty_cache()

    # ---- API-1, 4K P010 + 4:2:0 (BASELINE configs[0] / [3] shape on one GPU) ---------------------------------------------
    w, h = 3840, 2160
    px = w * h
    sdr = synth.make_sdr_yuv420(w, h).to(device)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG).to(device)
    enc1 = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY)
    base = sdr.clone()

    def api1():
        _, gm_ = enc1.generateGainMap(sdr, hdr)
        u.convertYuv(base, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3)
        fdct_planes(base, (0, 1, 2), (qy, qc, qc))
        u.fdct_quant_rgb(gm_, qy, qc)

    k = family_times(ctx, api1, fams, iters=5, warm=2)
    r = roof(px, k, {"generate_gainmap": 31.5, "convert_yuv": 3.0, "fdct_quant": 4.5 + 9.0})
    res["api1_4k_reference_operators"] = {
        "workload": "3840x2160 P010 (BT.2100 HLG) + YCbCr 4:2:0 -> generateGainMap (2 pass, 3 ch, scale 1: pass 1 4.5 in + 12 out, "
                    "pass 2 12 in + 3 out) + convertYuv + 3 x fdct_quant (base) + fused (rgb->ycc + 3 x fdct_quant) of the map: 8 launches",
        "us": r["chain_us"], "wall_us_per_chain": round(time_region(ctx, api1, iters=10, warm=2, reps=3) * 1e3, 1),
        "Mpx/s": round(px / r["chain_us"], 1), "roofline": r}

    # round 4: the same coefficients in four launches (uhdr_hip_encode_api1_fused_dev): pass 1, range + step tables, pass 2 fused
    # with the map's rgb->ycc + FDCTs (no 8-bit map round trip), convertYuv fused with the base image's three FDCTs
    def fused_chain(enc, s_, h_):
        def fn():
            enc.encodeApi1Fused(s_, h_, A.UHDR_CG_DISPLAY_P3, (qy, qc), (qy, qc), want_map=False)
        return fn

    fn4 = fused_chain(enc1, sdr, hdr)
    k = family_times(ctx, fn4, fams + ["encode_api1_chain"], iters=5, warm=2)
    chain4 = k.pop("encode_api1_chain", None)
    r = roof(px, k, {"generate_gainmap": 16.5, "fdct_quant": 18.0 + 4.5}, "encode.api1_4k")
    st = A.Stats()
    ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st))
    res["api1_4k"] = {
        "workload": "the API-1 4K chain fused (uhdr_hip_encode_api1_fused_dev, bit-identical coefficients): pass 1 (4.5 in + 12 out) + range / "
                    "step-table kernel + [pass 2 + rgb->ycc + 3 x fdct_quant] of the map (12 in + 6 out) + [convertYuv + 3 x fdct_quant] of the base "
                    "(1.5 in + 3 out): 4 launches, 39 B/px",
        "us": r["chain_us"], "chain_us_one_event_pair": chain4["us"] if chain4 else None,
        "chain_frac_one_event_pair": round(39.0 * px / (chain4["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if chain4 else None,
        "wall_us_per_chain": round(time_region(ctx, fn4, iters=10, warm=2, reps=3) * 1e3, 1),
        "Mpx/s": round(px / r["chain_us"], 1), "launches": sum(v["launches"] for v in k.values()), "roofline": r,
        "timing_note": "us = sum of the per-stage HIP-event pairs (each pair adds ~3 us to a launch); chain_us_one_event_pair = ONE pair around the four launches, "
                       "gaps included; wall_us_per_chain = host wall clock per call incl. the metadata synchronisation and the Python wrapper",
        "two_pass_channels_through_step_tables": int(st.generate_channels_tabled), "two_pass_channels_per_sample": int(st.generate_channels_per_sample)}
    del sdr, hdr, base
    torch.cuda.empty_cache()
    w, h = 7680, 4320
    px = w * h
    sdr = synth.make_sdr_yuv420(w, h).to(device)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG).to(device)
    fn8 = fused_chain(enc1, sdr, hdr)
    k = family_times(ctx, fn8, fams + ["encode_api1_chain"], iters=3, warm=1)
    chain8 = k.pop("encode_api1_chain", None)
    r = roof(px, k, {"generate_gainmap": 16.5, "fdct_quant": 18.0 + 4.5}, "encode.api1_8k")
    res["api1_8k"] = {"workload": "the fused API-1 chain at 7680x4320", "us": r["chain_us"], "chain_us_one_event_pair": chain8["us"] if chain8 else None,
                      "wall_us_per_chain": round(time_region(ctx, fn8, iters=6, warm=1, reps=3) * 1e3, 1), "Mpx/s": round(px / r["chain_us"], 1), "roofline": r}
    return res


def api1_roundtrip_section(ctx, u, device):
    """BASELINE.json's own metric -- "Mpixels/s encode+decode (API-1 P010+YUV420, 4K/8K)" -- device resident, bytes <-> pixels, each
    direction on its own (make_roundtrip's two closures) at 4K and at 8K.
    ONE pair of HIP events on the library's stream around `iters` back-to-back calls per direction (the entropy entry points are
    synchronous -- they read a byte count / status word back --, so host gaps are inside the pair), plus the per-family kernel sums."""
    from libultrahdr_amd import capi as A
    import torch

    res = {}
    fams = ["generate_gainmap", "fdct_quant", "huffman_encode", "huffman_decode", "idct_dequant", "apply_gainmap"]
    for tag, w, h, it in (("4k", 3840, 2160, 8), ("8k", 7680, 4320, 4)):
        px = w * h
        enc, dec, box = make_roundtrip(ctx, u, device, w, h)
        enc(False)
        ms_enc_seq = time_region(ctx, lambda: enc(False), iters=it, warm=2, reps=3)
        seq_sizes = (box["nb"], box["nm"])
        enc()
        assert seq_sizes == (box["nb"], box["nm"])
        ms_enc = time_region(ctx, enc, iters=it, warm=2, reps=3)
        k_enc = family_times(ctx, enc, fams, iters=3, warm=1)
        st0 = A.Stats()
        ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st0))
        dec()
        ctx.synchronize()
        t0 = time.perf_counter()
        dec()
        ctx.synchronize()
        once = time.perf_counter() - t0
        if once > 0.1:  # a scan the parallel decoder did not settle fell back to one lane (seconds): report that, do not repeat it
            ms_dec, k_dec = once * 1e3, {}
        else:
            ms_dec = time_region(ctx, dec, iters=it, warm=2, reps=3)
            k_dec = family_times(ctx, dec, fams, iters=3, warm=1)
            res[f"api1_{tag}_dec_one_scan_at_a_time_us"] = round(time_region(ctx, lambda: dec(False), iters=it, warm=2, reps=3) * 1e3, 1)
        st1 = A.Stats()
        ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st1))
        res[f"api1_{tag}_enc_us"] = round(ms_enc * 1e3, 1)
        res[f"api1_{tag}_enc_one_scan_at_a_time_us"] = round(ms_enc_seq * 1e3, 1)
        res[f"api1_{tag}_dec_us"] = round(ms_dec * 1e3, 1)
        res[f"api1_{tag}_roundtrip_Mpxs"] = round(px / ((ms_enc + ms_dec) * 1e-3) / 1e6, 1)
        res[f"api1_{tag}_enc_kernels_us"] = round(sum(v["us"] for v in k_enc.values()), 1)
        res[f"api1_{tag}_dec_kernels_us"] = round(sum(v["us"] for v in k_dec.values()), 1)
        res[f"api1_{tag}_detail"] = {
            "scan_bytes_base": box["nb"], "scan_bytes_map": box["nm"], "encode_kernels": k_enc, "decode_kernels": k_dec,
            "decode_route": {"parallel": int(st1.entropy_decode_parallel - st0.entropy_decode_parallel),
                             "single_lane": int(st1.entropy_decode_single_lane - st0.entropy_decode_single_lane),
                             "declined": int(st1.entropy_decode_declined - st0.entropy_decode_declined)},
            # 39 B/px of the fused chain (encode_section) + coefficients read once by each Huffman pass (3 + 6) + the bytes written
            "encode_algorithmic_bytes": int(48.0 * px + box["nb"] + box["nm"]),
            # bytes read + coefficients written (3 + 6) and read again (3 + 6), the map's RGBA written + read (4 + 4), F16 out (8)
            "decode_algorithmic_bytes": int(34.0 * px + box["nb"] + box["nm"])}
        res[f"api1_{tag}_enc_frac"] = round(res[f"api1_{tag}_detail"]["encode_algorithmic_bytes"] / (ms_enc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res[f"api1_{tag}_dec_frac"] = round(res[f"api1_{tag}_detail"]["decode_algorithmic_bytes"] / (ms_dec * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        del enc, dec, box
        torch.cuda.empty_cache()
    res["api1_note"] = ("device-resident API-1 round trip, synthetic noisy frames at q95 (dense streams: ~3.3 MB base scan per 4K frame); enc_us / dec_us: "
                        "one HIP-event pair around back-to-back calls, host gaps of the synchronous entropy entry points included, the file's two scans "
                        "entropy-coded concurrently (uhdr_hip_huffman_{encode,decode}2_dev; *_one_scan_at_a_time_us = the same with two single-scan "
                        "calls); *_kernels_us: sum of the per-launch HIP events of the same calls (overlapping launches counted in full)")
    return res


def concurrent_section(device, dev_index, nctx=4, steps=12, warmup=3):
    """The same 4K API-1 round trip with SEVERAL frames in flight: `nctx` contexts (own streams, scratch, worker thread), one host thread each,
    every thread loops encode + decode of its own frame.  One round trip alone is a chain of latency-bound launches that leaves most of the device
    idle; a service that transcodes independent frames fills it this way.  Whole-job rate = frames x pixels / wall clock between two
    barriers (all threads synchronised their contexts before each).  Not the line's `value` (one frame per step): a second figure beside it."""
    import threading

    from libultrahdr_amd.ultrahdr import Context, UltraHdr

    w, h = 3840, 2160
    res = {}
    for n in (2, nctx):
        ctxs = [Context(dev_index) for _ in range(n)]
        work = []
        for i, c in enumerate(ctxs):
            enc, dec, box = make_roundtrip(c, UltraHdr(ctx=c), device, w, h, seed=4321 + i)
            c.stream_safe = False
            work.append((enc, dec, box))
        bar = threading.Barrier(n + 1)
        errs = []

        def loop(i):
            enc, dec, _ = work[i]
            try:
                for _ in range(warmup):
                    enc()
                    dec()
                ctxs[i].synchronize()
                bar.wait()
                for _ in range(steps):
                    enc()
                    dec()
                ctxs[i].synchronize()
                bar.wait()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
                bar.abort()

        th = [threading.Thread(target=loop, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        try:
            bar.wait()
            t0 = time.perf_counter()
            bar.wait()
            dt = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            dt = None
        for t in th:
            t.join()
        if errs or dt is None:
            for c in ctxs:
                c.close()
            raise RuntimeError("; ".join(errs) or "barrier broken")
        res[f"frames_in_flight_{n}_Mpxs"] = round(n * steps * w * h / dt / 1e6, 1)
        res[f"frames_in_flight_{n}_ms_per_round_trip"] = round(dt / steps * 1e3, 4)
        del work
        for c in ctxs:  # (explicitly: a context owns a stream, an auxiliary context and its worker thread)
            c.close()
        del ctxs
        import torch

        torch.cuda.empty_cache()
    res["workload"] = (f"4K API-1 round trip (the line's step), 2 and {nctx} independent frames in flight: one context + one host thread per frame, {steps} round trips each "
                       "after a warm-up, wall clock between two barriers; Mpixels/s of all frames together")
    return res


def config4_section(ctx, u, device, rank, world, backend):
    """BASELINE configs[3]: API-1 encode of a 16384-wide P010 + YCbCr 4:2:0 image sharded by row stripe, 2048 rows per rank
    (at 8 ranks: 16K x 16K), the WHOLE per-stripe chain on every rank:
      two-pass 3-channel generateGainMap -- pass 1 -> ONE all-reduce(min) of {min0..2, -max0..2}, issued by the C++ host layer on
      its own stream (RCCL over xGMI) -> range finalised on the device -> pass 2;
      convertYuv of the base stripe; FDCT + quantize of the base planes and (fused with rgb -> ycc) of the map stripe;
      Huffman coding of the stripe's MCU rows (restart intervals: stripes are independent, RST7 joins them);
      device gather of the two entropy-coded streams to rank 0 (one all-gather of the sizes, one group of send / recv).
    Weak scaling: the image grows with the rank count.  value = pixels of all ranks / max-over-ranks time.
    UHDR_BENCH_DIST_BACKEND=gloo (ranks share devices) swaps RCCL for the host-relay transport: a dry run of the N > 1 code path."""
    import torch
    import torch.distributed as dist

    from libultrahdr_amd import capi as A
    from libultrahdr_amd import stripes, synth
    from libultrahdr_amd.images import Image
    from libultrahdr_amd.ultrahdr import UltraHdr

    ws, hs, iters = 16384, 2048, 10
    relay = world > 1 and backend != "nccl"
    nranks = stripes.init_comm_relay(ctx) if relay else stripes.init_comm(ctx)
    enc = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY)
    cfg = enc.encode_cfg()
    sdr_s = synth.make_sdr_yuv420(ws, hs, noise=0.0, seed=1234 + rank).to(device)
    hdr_s = synth.make_hdr_p010(ws, hs, ct=A.UHDR_CT_HLG, noise=0.0, seed=1234 + rank).to(device)
    qy, qc = u.quant_table(95, False), u.quant_table(95, True)
    s420, s444 = [(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3
    ri420, ri444 = 10, 21  # the longest restart interval one wavefront holds (64 blocks): 10 MCUs at 4:2:0, 21 at 4:4:4

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    out_bytes = [0, 0]

    def one_image():
        # round 4: the fused chain on the striped context (uhdr_hip_encode_api1_fused_dev: pass 1 -> reduce -> ONE all-reduce ->
        # finalize + tables -> [pass 2 + rgb->ycc + FDCT] of the map stripe, [convertYuv + FDCT] of the base stripe)
        base, mapc, md_, _ = enc.encodeApi1Fused(sdr_s, hdr_s, A.UHDR_CG_DISPLAY_P3, (qy, qc), (qy, qc), want_map=False)
        e_base = u.huffman_encode(base, ws, hs, s420, ri420)
        e_map = u.huffman_encode(list(mapc), ws, hs, s444, ri444)
        pb = stripes.gather_streams_to_root(ctx, e_base, root=0)
        pm = stripes.gather_streams_to_root(ctx, e_map, root=0)
        if pb is not None:
            out_bytes[0] = sum(int(t.numel()) for t in pb) + 2 * (len(pb) - 1)
            out_bytes[1] = sum(int(t.numel()) for t in pm) + 2 * (len(pm) - 1)
        return md_

    md = None
    for _ in range(12):  # warm-up + clock ramp (count based: every rank takes part in the same collectives)
        md = one_image()
    sync_all()
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    t0 = time.perf_counter()
    for _ in range(iters):
        md = one_image()
    sync_all()
    el = time.perf_counter() - t0
    fam = {}
    for f in ("stripe_exchange", "generate_gainmap", "convert_yuv", "fdct_quant", "huffman_encode"):
        n_f, ms_f = ctx.profile_read(f, reset=True)
        fam[f] = round(ms_f / iters * 1e3, 1)
    ctx.profile_read(None, reset=True)
    ctx.profile(False)
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    px = ws * hs * world

    # ---- the decode counterpart (round 4): ONE 16384 x (2048 * world) frame decoded by row stripes, the gain map replicated ----
    # JpegR::applyGainMap hands row ranges to its job queue the same way (jpegr.cpp:1714-1812); no exchange step at all:
    # uhdr_hip_apply_gainmap_dev(..., y0, full_height) on every rank's rows.  Map: the Android-style Y400 map at scale 4 of
    # the WHOLE frame (every rank holds all of it: 2 MB per 2048 rows).
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    full_h = hs * world
    md_d = synth.default_metadata(use_base_cg=0)
    sdr_d = synth.make_sdr_yuv420(ws, hs, seed=4321 + rank).to(device)
    gm_d = synth.make_gainmap(ws // 4, full_h // 4, 1, seed=777).to(device)  # the same map on every rank
    sdr_d.raw.cg, gm_d.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    dst_d = [Image(f16, ws, hs, align=64, device=device) for _ in range(3)]
    kd = [0]

    def decode_stripe():
        u.applyGainMap(sdr_d, gm_d, md_d, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst_d[kd[0] % 3], y0=rank * hs, full_height=full_h)
        kd[0] += 1

    for _ in range(40):
        decode_stripe()
    sync_all()
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    t0 = time.perf_counter()
    for _ in range(30):
        decode_stripe()
    sync_all()
    el_d = time.perf_counter() - t0
    each_d = ctx.profile_read_list("apply_gainmap", reset=True)
    ctx.profile(False)
    kern_us_d = sum(each_d) / max(len(each_d), 1) * 1e3
    if world > 1:
        tt = torch.tensor([el_d], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el_d = float(tt.item())
    striped_decode = {"workload": f"decode of ONE {ws}x{full_h} frame (YCbCr 4:2:0 + Y400 map at scale 4 -> RGBA_F16) by row stripes, {hs} rows per rank, the map "
                                  "replicated: uhdr_hip_apply_gainmap_dev(y0, full_height), no collective",
                      "wall_us_per_frame": round(el_d / 30 * 1e6, 1), "Mpx/s": round(px * 30 / el_d / 1e6, 1),
                      "rank0_kernel_us_per_stripe": round(kern_us_d, 1),
                      "rank0_kernel_frac_of_8TBs": round(algo_bytes_per_px("A") * ws * hs / (kern_us_d * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if kern_us_d else None,
                      "note": "wall clock = max over ranks around 30 frames, launched from Python (the host loop, not the GPU, bounds it at one rank); the kernel figure is "
                              "rank 0's HIP-event time per stripe launch"}
    del sdr_d, gm_d, dst_d
    torch.cuda.empty_cache()
    # the exchange on its own: 50 back-to-back min-all-reduces of the six extrema on the library's stream
    t0 = time.perf_counter()
    for _ in range(50):
        stripes.all_reduce_probe(ctx)
    sync_all()
    allreduce_us = (time.perf_counter() - t0) / 50 * 1e6
    # N = 1 only: the WHOLE 16384 x 16384 image of configs[3] on one GPU (it fits: 1.2 GB of intents, 5.6 GB of coefficients and
    # gain ratios) -- the denominator of the strong-scaling point an 8-GPU run of this section gives (8 ranks x 2048 rows)
    full = None
    if world == 1 and not os.environ.get("UHDR_BENCH_NO_FULL_16K"):
        try:
            nrep = 16384 // hs

            def tiled(img):  # the stripe's planes stacked nrep times, on the device (no 268-Mpx synthesis on the host)
                big = Image(img.fmt, ws, hs * nrep, img.raw.cg, img.raw.ct, img.raw.range, align=64, device=device)
                for pl, lay in enumerate(img.layout):
                    if lay is None:
                        continue
                    src = img.plane_tensor(pl)
                    dstp = big.plane_tensor(pl)
                    for r in range(nrep):
                        dstp[r * src.shape[0]:(r + 1) * src.shape[0]].copy_(src)
                torch.cuda.synchronize()
                return big

            sdr_f, hdr_f = tiled(sdr_s), tiled(hdr_s)
            hf = hs * nrep

            def full_image():
                base, mapc, md_, _ = enc.encodeApi1Fused(sdr_f, hdr_f, A.UHDR_CG_DISPLAY_P3, (qy, qc), (qy, qc), want_map=False)
                eb_ = u.huffman_encode(base, ws, hf, s420, ri420)
                em_ = u.huffman_encode(list(mapc), ws, hf, s444, ri444)
                return int(eb_.numel()), int(em_.numel())

            for _ in range(2):
                nbm = full_image()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                nbm = full_image()
            ctx.synchronize()
            el_f = (time.perf_counter() - t0) / 3
            full = {"workload": f"the whole {ws}x{hf} image on one GPU: fused API-1 chain + Huffman coding (restart intervals), no stripes, no exchange",
                    "ms_per_image": round(el_f * 1e3, 3), "Mpx/s": round(ws * hf / el_f / 1e6, 1), "jpeg_scan_bytes_base_and_map": list(nbm),
                    "ratio_to_one_2048_row_stripe": round(el_f / (el / iters), 2)}
            del sdr_f, hdr_f
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            full = {"error": f"{type(e).__name__}: {e}"}
    sync_all()
    ctx.lib.uhdr_hip_comm_destroy(ctx.handle)  # the context goes back to whole images (the sections behind this one)
    return {"workload": f"configs[3]: API-1 encode of a {ws}x{hs * world} P010 + YCbCr 4:2:0 image, {hs} rows per rank, {world} rank(s): the fused chain "
                        "(two-pass 3-channel generateGainMap whose pass 2 feeds the map's rgb->ycc + FDCT directly, convertYuv inside the base image's FDCT: "
                        "uhdr_hip_encode_api1_fused_dev) + Huffman coding (restart intervals) + gather of the entropy-coded streams to rank 0",
            "collective": "one all-reduce(min, 6 x float32 {min0..2, -max0..2}) per image, issued by libuhdr_hip.so on its own stream between pass 1 "
                          "and pass 2; range finalised on the device; stream sizes: one 8-byte all-gather, stream bytes: one send / recv group to rank 0",
            "transport": "host relay over torch.distributed (dry run: the ranks share devices)" if relay else "RCCL (ncclAllReduce / ncclAllGather / ncclSend + ncclRecv)",
            "ncclCommCount": nranks, "images": iters, "ms_per_image": round(el / iters * 1e3, 3), "Mpx/s": round(px * iters / el / 1e6, 1),
            "rank0_us_per_image": fam, "all_reduce_us_back_to_back": round(allreduce_us, 1), "jpeg_scan_bytes_base_and_map": out_bytes,
            "striped_decode": striped_decode, "full_16k_x_16k_one_gpu": full,
            "max_content_boost": [round(float(v), 6) for v in md.max_content_boost]}


def config5_section(ctx, u, device):
    """BASELINE configs[4] on one GPU: a batch of 32 4K frames decoded to HLG RGBA1010102 (Y400 map, scale 4),
    captured into a HIP graph once and replayed."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    import torch

    nb, w, h = 32, 3840, 2160
    u32 = A.UHDR_IMG_FMT_32bppRGBA1010102
    md = synth.default_metadata(use_base_cg=0)
    sets = make_frames(nb, w, h, "A", device, u32, seed0=555)
    for s5, g5, _ in sets:
        s5.raw.cg, g5.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    args5 = ([f[0] for f in sets], [f[1] for f in sets], md, A.UHDR_CT_HLG, u32, A.FLT_MAX, [f[2] for f in sets])
    u.applyGainMapBatch(*args5)  # warm: tables, occupancy queries
    ctx.synchronize()
    ms = time_kernel(ctx, lambda: u.applyGainMapBatch(*args5), iters=10, warm=2)  # kernel time by HIP events, eager launches
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    res = {}
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            u.applyGainMapBatch(*args5)
        torch.cuda.synchronize()
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < CLOCK_RAMP_S:  # see clock_ramp
            for _ in range(10):
                graph.replay()
            torch.cuda.synchronize()
        reps, walls = 20, []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) / reps)
        walls.sort()
        wall = walls[len(walls) // 2]
        b = algo_bytes_per_px("A", 4) * w * h * nb
        res = {"workload": "configs[4] per-GPU share scaled to one GPU: 32 x 3840x2160 YCbCr420 + Y400 map (scale 4) -> HLG RGBA1010102, "
                           "uhdr_hip_apply_gainmap_batch_dev captured in a HIP graph, replayed",
               "graph_replay_us_per_batch": round(wall * 1e6, 1), "us_per_frame": round(wall * 1e6 / nb, 2),
               "Mpx/s": round(nb * w * h / wall / 1e6, 1), "GB/s": round(b / wall / 1e9, 1), "frac_of_8TBs": round(b / wall / 1e9 / HBM_PEAK_GBS, 4),
               "kernel_us_per_batch_eager_hip_events": round(ms * 1e3, 1), "timing": "wall clock around 20 replays, median of 5"}
    finally:
        ctx.set_stream(None)
    del sets
    torch.cuda.empty_cache()
    return res


def seam_split(fn, n):
    """Per-call medians are not available from tallies: run fn() n times between a reset and a read of the facade's stage table
    (uhdr_hip_seam_stats) -> (mean ms of the device stages per call, mean ms of the whole accelerated call, {stage: calls})."""
    from libultrahdr_amd import capi as A

    A.seam_stats(reset=True)
    for _ in range(n):
        fn()
    st = A.seam_stats(reset=True)
    call = st.pop("uhdr_call", None)
    stages = sum(v["device_ms"] for k, v in st.items() if k not in ("gainmap_copy_deferred", "gainmap_image_asked_for"))
    return stages / n, (call["device_ms"] / n if call else None), {k: v["device"] for k, v in st.items()}


def api_level_section():
    """SURVEY.md 8(d) configs[1] at the API level: the reference's own uhdr_decode / uhdr_encode through the drop-in
    libuhdr.so (facade/), with uhdr_enable_gpu_acceleration(codec, 1) -- host buffers in and out, PCIe and the CPU-side
    JPEG entropy coding included.  The CPU-only numbers of the same calls are in cpu_baseline.stages."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import facade as FA
    from libultrahdr_amd import synth

    if not FA.available():
        return {"error": "libuhdr.so facade not built on this machine"}
    w, h = 3840, 2160
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
    sdr = synth.make_sdr_yuv420(w, h)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat

    def med(fn, n):
        # the uhdr_encode / uhdr_decode call itself (facade.last_call_seconds): creating the codec object, handing it the
        # inputs and copying the 66 MB result into a numpy array are the Python harness's, not the library's
        ts = []
        for _ in range(n):
            r = fn()
            ts.append(FA.last_call_seconds)
        return r, sorted(ts)[len(ts) // 2]

    def with_env(name, value, fn):
        os.environ[name] = value
        try:
            return fn()
        finally:
            del os.environ[name]

    FA.encode(hdr, sdr, gpu=True)  # warm-up: context creation, tables
    # default since round 3: the whole compressImage on the device, marker-less Huffman coding included -> the reference's bytes
    jpg, t_enc = med(lambda: FA.encode(hdr, sdr, gpu=True), 5)
    FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True)  # warm-up: the decoder's table forms and scratch on the pooled context
    _, t_dec = med(lambda: FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True), 5)
    # round 4: the decoded images stay on the device and the gain-map image is downloaded when uhdr_get_decoded_gainmap_image
    # asks for it.  The cost of asking, and the round-3 behaviour (UHDR_HIP_SEAM_EAGER_DOWNLOADS=1) for comparison
    t_gm = []
    for _ in range(3):
        FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True, want_gainmap=True)
        t_gm.append(FA.last_gainmap_seconds)
    t_gm = sorted(t_gm)[1]
    _, t_dec_eager = with_env("UHDR_HIP_SEAM_EAGER_DOWNLOADS", "1", lambda: med(lambda: FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True), 5))
    # the round-2 default, kept as an option: device FDCT, libjpeg's Huffman pass on one CPU core (same bytes)
    jpg_cpu, t_enc_cpu = with_env("UHDR_HIP_SEAM_CPU_ENTROPY", "1", lambda: med(lambda: FA.encode(hdr, sdr, gpu=True), 2))
    # the round-4 route: four per-stage seams instead of the one at encodeJPEGR (same bytes)
    jpg_ps, t_enc_ps = with_env("UHDR_HIP_SEAM_NO_FUSED_ENCODE", "1", lambda: med(lambda: FA.encode(hdr, sdr, gpu=True), 5))
    jpg2, t_enc2 = med(lambda: FA.encode(hdr, sdr, gpu=True), 5)  # ... and the fused seam once more, later in the process
    # where the time of those two calls goes: the seam's own stage table (uhdr_hip_seam_stats) over five more calls
    # each -- the device stages against the reference's own host code around them (two ICC profiles, container parsing and writing,
    # copies; until round 5 also 50-110 MB of value-initialised buffers per call, which the facade's blocks now get from calloc)
    split = {}
    try:
        se, ce, ne = seam_split(lambda: FA.encode(hdr, sdr, gpu=True), 5)
        sd, cd, nd = seam_split(lambda: FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True), 5)
        split = {"uhdr_encode_4k_device_stages_ms": round(se, 2), "uhdr_encode_4k_scope_ms": round(ce, 2) if ce is not None else None,
                 "uhdr_encode_4k_stage_calls": ne,
                 "uhdr_decode_4k_device_stages_ms": round(sd, 2), "uhdr_decode_4k_scope_ms": round(cd, 2) if cd is not None else None,
                 "uhdr_decode_4k_stage_calls": nd, "note": "means over 5 calls, from the library's stage table (uhdr_hip_seam_stats)"}
    except Exception as e:  # noqa: BLE001
        split = {"error": f"{type(e).__name__}: {e}"}
    # opt-in: restart intervals, one per wavefront
    with_env("UHDR_HIP_SEAM_RESTART_INTERVAL", "max", lambda: FA.encode(hdr, sdr, gpu=True))
    jpg_ri, t_enc_ri = with_env("UHDR_HIP_SEAM_RESTART_INTERVAL", "max", lambda: med(lambda: FA.encode(hdr, sdr, gpu=True), 5))
    _, t_dec_ri = med(lambda: FA.decode(jpg_ri, A.UHDR_CT_LINEAR, f16, gpu=True), 5)

    def row(t, **kw):
        return dict({"ms": round(t * 1e3, 1), "Mpx/s": round(w * h / t / 1e6, 1)}, **kw)

    # the same two calls at 8K (BASELINE's metric names 4K / 8K)
    rows8 = {}
    try:
        w8, h8 = 7680, 4320
        hdr8, sdr8 = synth.make_hdr_p010(w8, h8, ct=A.UHDR_CT_HLG), synth.make_sdr_yuv420(w8, h8)
        FA.encode(hdr8, sdr8, gpu=True)
        jpg8, t_enc8 = med(lambda: FA.encode(hdr8, sdr8, gpu=True), 3)
        FA.decode(jpg8, A.UHDR_CT_LINEAR, f16, gpu=True)
        _, t_dec8 = med(lambda: FA.decode(jpg8, A.UHDR_CT_LINEAR, f16, gpu=True), 3)
        rows8 = {"uhdr_encode_api1_8k_hip": {"ms": round(t_enc8 * 1e3, 1), "Mpx/s": round(w8 * h8 / t_enc8 / 1e6, 1), "jpeg_bytes": len(jpg8)},
                 "uhdr_decode_8k_f16_hip": {"ms": round(t_dec8 * 1e3, 1), "Mpx/s": round(w8 * h8 / t_dec8 / 1e6, 1)}}
        del hdr8, sdr8, jpg8
    except Exception as e:  # noqa: BLE001
        rows8 = {"uhdr_8k": {"error": f"{type(e).__name__}: {e}"}}

    # BASELINE config 3 through the drop-in (round 6): uhdr_encode API-0 of an 8K RGBA1010102 PQ intent -- one device sequence behind the seam at
    # JpegR::encodeJPEGR API-0 (uhdr_hip_encode_api0_scans) -- and the five per-stage seams for comparison
    rows0 = {}
    try:
        w8, h8 = 7680, 4320
        hdr0 = synth.make_hdr_rgba1010102(w8, h8, ct=A.UHDR_CT_PQ)
        FA.encode(hdr0, None, gpu=True)
        A.seam_stats(reset=True)
        jpg0, t0_ = med(lambda: FA.encode(hdr0, None, gpu=True), 3)
        st0 = A.seam_stats(reset=True)
        _, t0_ps = with_env("UHDR_HIP_SEAM_NO_FUSED_ENCODE", "1", lambda: med(lambda: FA.encode(hdr0, None, gpu=True), 2))
        rows0 = {"uhdr_encode_api0_8k_hip": {"ms": round(t0_ * 1e3, 1), "Mpx/s": round(w8 * h8 / t0_ / 1e6, 1), "jpeg_bytes": len(jpg0),
                                             "device_stages": {k: v["device"] for k, v in st0.items() if k != "uhdr_call"}},
                 "uhdr_encode_api0_8k_hip_per_stage_seams": {"ms": round(t0_ps * 1e3, 1), "Mpx/s": round(w8 * h8 / t0_ps / 1e6, 1)}}
        del hdr0, jpg0
    except Exception as e:  # noqa: BLE001
        rows0 = {"uhdr_encode_api0_8k_hip": {"error": f"{type(e).__name__}: {e}"}}

    return {**rows8, **rows0, "seam_trace_split": split, "uhdr_encode_api1_4k_hip": row(t_enc, jpeg_bytes=len(jpg), same_bytes_as_the_libjpeg_entropy_route=bool(jpg == jpg_cpu),
                                           entropy_coding="device, no restart markers (the default): FDCT + quantize + Huffman coding in three passes, "
                                                          "the file is the reference's byte for byte"),
            "uhdr_decode_4k_f16_hip": row(t_dec, entropy_decoding="device (self-synchronising decoder: the file has no restart markers)",
                                          gainmap_image="stays on the device; uhdr_get_decoded_gainmap_image downloads it when called",
                                          get_decoded_gainmap_image_ms=round(t_gm * 1e3, 2)),
            "uhdr_decode_4k_f16_hip_eager_downloads": row(t_dec_eager, note="UHDR_HIP_SEAM_EAGER_DOWNLOADS=1: both decoded images written to the JpegDecoderHelper "
                                                          "buffers and the gain-map image copied inside uhdr_decode (the round-3 behaviour)"),
            "uhdr_encode_api1_4k_hip_per_stage_seams": row(t_enc_ps, same_bytes=bool(jpg_ps == jpg), note="UHDR_HIP_SEAM_NO_FUSED_ENCODE=1: generate_gainmap, convert_yuv and 2 x jpeg_encode_scan seams (round 4)"),
            "uhdr_encode_api1_4k_hip_again": row(t_enc2, same_bytes=bool(jpg2 == jpg)),
            "uhdr_encode_api1_4k_hip_libjpeg_entropy": row(t_enc_cpu, jpeg_bytes=len(jpg_cpu), entropy_coding="UHDR_HIP_SEAM_CPU_ENTROPY=1: device FDCT, libjpeg's Huffman pass on one CPU core (the round-2 default)"),
            "uhdr_encode_api1_4k_hip_restart_intervals": row(t_enc_ri, jpeg_bytes=len(jpg_ri), entropy_coding="UHDR_HIP_SEAM_RESTART_INTERVAL=max: device, one restart interval per wavefront: DRI + RSTn markers added, decoded pixels identical"),
            "uhdr_decode_4k_f16_hip_of_that_file": row(t_dec_ri, entropy_decoding="device (restart-interval file)"),
            "note": "libuhdr.so facade, uhdr_enable_gpu_acceleration(1): host buffers in and out (pageable), PCIe included; container / "
                    "metadata handling is the reference's CPU code.  The CPU-only numbers of the same calls are in cpu_baseline.stages"}


def extras(ctx, u, device):
    """Stage-level kernel timings (HIP events), rank 0, N=1.  GB/s = algorithmic bytes / time."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from libultrahdr_amd.images import Image
    from libultrahdr_amd.ultrahdr import UltraHdr
    import numpy as np
    import torch

    res = {}
    f16, u32 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA1010102
    md = synth.default_metadata(use_base_cg=0)

    def apply_case(name, w, h, map_kind, ct, extra_modes=False):
        fmt = f16 if ct == A.UHDR_CT_LINEAR else u32
        sets = make_frames(3, w, h, map_kind, device, fmt, seed0=77)  # rotate 3 sets: > L3 at 8K, mostly at 4K
        for s, g, _ in sets:
            s.raw.cg, g.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
        k = [0]
        # direct C-ABI calls (no Python wrapper between launches): the launches queue back to back, as in a decode loop
        lib_, hnd_ = ctx.lib, ctx.handle
        argv = [(C.byref(s.raw), C.byref(g.raw), C.byref(md), C.byref(d.raw)) for s, g, d in sets]

        def fn():
            s_, g_, m_, d_ = argv[k[0] % 3]
            k[0] += 1
            st = lib_.uhdr_hip_apply_gainmap_dev(hnd_, s_, g_, m_, ct, fmt, A.FLT_MAX, d_, 0, 0)
            if st.error_code != 0:
                raise RuntimeError(st.detail)

        ms = time_region(ctx, fn, iters=30, warm=6)
        b = algo_bytes_per_px(map_kind, 8 if ct == A.UHDR_CT_LINEAR else 4) * w * h
        res[name] = {"us": round(ms * 1e3, 2), "timing": "one HIP-event pair around 30 back-to-back launches, median of 3 regions", "GB/s": round(b / (ms / 1e3) / 1e9, 1), "frac_of_8TBs": round(b / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                     "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1)}
        if extra_modes:
            # (a) a burst: 10 launches after the device sat idle (what a single decode request sees)
            bursts = []
            for _ in range(4):
                ctx.synchronize()
                time.sleep(0.03)
                bursts.append(time_region(ctx, fn, iters=10, warm=0, reps=1))
            mb = min(bursts)
            # (b) two contexts = two HIP streams, launches alternating: the tail of one launch overlaps the ramp of the next
            from libultrahdr_amd.ultrahdr import Context
            ctx2 = Context(ctx._device, stream_safe=False)
            h2 = ctx2.handle
            j = [0]

            def both():
                s_, g_, m_, d_ = argv[j[0] % 3]
                hh = hnd_ if j[0] % 2 == 0 else h2
                j[0] += 1
                st = lib_.uhdr_hip_apply_gainmap_dev(hh, s_, g_, m_, ct, fmt, A.FLT_MAX, d_, 0, 0)
                if st.error_code != 0:
                    raise RuntimeError(st.detail)

            for _ in range(6):
                both()
            ctx.synchronize(); ctx2.synchronize()
            walls = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(60):
                    both()
                ctx.synchronize(); ctx2.synchronize()
                walls.append((time.perf_counter() - t0) / 60)
            m2 = sorted(walls)[1] * 1e3
            res[name].update({"burst_us": round(mb * 1e3, 2), "burst_frac_of_8TBs": round(b / (mb / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                              "two_streams_us": round(m2 * 1e3, 2), "two_streams_frac_of_8TBs": round(b / (m2 / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                              "two_streams_timing": "wall clock around 60 launches alternating between two contexts, median of 3"})
            del ctx2
        del sets
        torch.cuda.empty_cache()

    apply_case("apply_8k_f16_mapC", 7680, 4320, "C", A.UHDR_CT_LINEAR, extra_modes=True)
    apply_case("apply_8k_f16_mapB", 7680, 4320, "B", A.UHDR_CT_LINEAR)
    apply_case("apply_8k_f16_mapA", 7680, 4320, "A", A.UHDR_CT_LINEAR)
    apply_case("apply_4k_f16_mapC_single_launch", 3840, 2160, "C", A.UHDR_CT_LINEAR)
    apply_case("apply_4k_f16_mapA_single_launch", 3840, 2160, "A", A.UHDR_CT_LINEAR)
    apply_case("apply_4k_hlg_mapA", 3840, 2160, "A", A.UHDR_CT_HLG)
    apply_case("apply_4k_pq_mapA", 3840, 2160, "A", A.UHDR_CT_PQ)

    # BASELINE config 5: a batch of 4K frames decoded to HLG RGBA1010102 (map A), one launch
    nb5 = 16
    sets5 = make_frames(nb5, 3840, 2160, "A", device, u32, seed0=555)
    for s5, g5, _ in sets5:
        s5.raw.cg, g5.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    ms = time_kernel(ctx, lambda: u.applyGainMapBatch([f[0] for f in sets5], [f[1] for f in sets5], md, A.UHDR_CT_HLG, u32, A.FLT_MAX,
                                                      [f[2] for f in sets5]), iters=5, warm=2)
    b5 = algo_bytes_per_px("A", 4) * 3840 * 2160 * nb5
    res["apply_4k_hlg_mapA_batch16_one_launch"] = {"us": round(ms * 1e3, 1), "us_per_frame": round(ms * 1e3 / nb5, 2),
                                                    "GB/s": round(b5 / (ms / 1e3) / 1e9, 1), "Mpx/s": round(nb5 * 3840 * 2160 / (ms / 1e3) / 1e6, 1)}
    del sets5
    torch.cuda.empty_cache()

    # 4:4:4 (what an API-0 stream decodes to) and RGBA8888 bases: the quad kernel's BASE 1 / 2 / 3 variants
    def generic_case(name, base_fmt, map_kind):
        w_, h_ = 3840, 2160
        base = Image(base_fmt, w_, h_, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=64, device=device)
        base.buf.random_(0, 256)
        torch.cuda.synchronize()
        g = (synth.make_gainmap(w_ // 4, h_ // 4, 1, seed=9) if map_kind == "A" else synth.make_gainmap(w_, h_, 3, alpha=True, seed=9)).to(device)
        g.raw.cg = A.UHDR_CG_BT_2100
        d = Image(f16, w_, h_, align=64, device=device)
        ms = time_kernel(ctx, lambda: u.applyGainMap(base, g, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, d), iters=6, warm=2)
        bpp_in = {A.UHDR_IMG_FMT_24bppYCbCr444: 3, A.UHDR_IMG_FMT_32bppRGBA8888: 4, A.UHDR_IMG_FMT_16bppYCbCr422: 2}[base_fmt]
        b = (bpp_in + (1 / 16 if map_kind == "A" else 4) + 8) * w_ * h_
        res[name] = {"us": round(ms * 1e3, 2), "GB/s": round(b / (ms / 1e3) / 1e9, 1), "Mpx/s": round(w_ * h_ / (ms / 1e3) / 1e6, 1)}

    generic_case("apply_4k_f16_444base_mapC_quad_kernel", A.UHDR_IMG_FMT_24bppYCbCr444, "C")
    generic_case("apply_4k_f16_444base_mapA_quad_kernel", A.UHDR_IMG_FMT_24bppYCbCr444, "A")
    generic_case("apply_4k_f16_rgba8888base_mapC_quad_kernel", A.UHDR_IMG_FMT_32bppRGBA8888, "C")
    generic_case("apply_4k_f16_422base_mapC_quad_kernel", A.UHDR_IMG_FMT_16bppYCbCr422, "C")

    # the drop-in boundary with HOST buffers (H2D + kernel + D2H, pageable memory): PCIe-inclusive rate
    hw, hh = 3840, 2160
    hs = synth.make_sdr_yuv420(hw, hh, seed=5)
    hg = synth.make_gainmap(hw, hh, 3, alpha=True, seed=6)
    hs.raw.cg, hg.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    hd = Image(f16, hw, hh, align=1)
    u.applyGainMap(hs, hg, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, hd)
    t0 = time.perf_counter()
    for _ in range(3):
        u.applyGainMap(hs, hg, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, hd)
    el = (time.perf_counter() - t0) / 3
    res["apply_4k_f16_mapC_host_buffers_pcie_inclusive"] = {"ms": round(el * 1e3, 2), "Mpx/s": round(hw * hh / el / 1e6, 1)}

    # encode side, 4K: API-1 defaults (two-pass, 3-channel, scale 1) and the realtime preset
    w, h = 3840, 2160
    sdr = synth.make_sdr_yuv420(w, h).to(device)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG).to(device)
    enc = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY)
    ms = time_kernel(ctx, lambda: enc.generateGainMap(sdr, hdr), iters=5, warm=2)
    res["generate_4k_2pass_3ch_s1"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                       "GB/s": round(31.5 * w * h / (ms / 1e3) / 1e9, 1)}
    rt = UltraHdr(ctx=ctx, mapDimensionScaleFactor=4, useMultiChannelGainMap=False, preset=A.UHDR_USAGE_REALTIME)
    ms = time_kernel(ctx, lambda: rt.generateGainMap(sdr, hdr), iters=5, warm=2)
    res["generate_4k_1pass_1ch_s4"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                       "GB/s": round((4.5 + 1 / 16) * w * h / (ms / 1e3) / 1e9, 1)}
    tm_out = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, align=64, device=device)
    ms = time_kernel(ctx, lambda: u.toneMap(hdr, tm_out), iters=5, warm=2)
    res["tonemap_4k_p010"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "GB/s": round(4.5 * w * h / (ms / 1e3) / 1e9, 1)}
    cv = sdr.clone()
    ms = time_kernel(ctx, lambda: u.convertYuv(cv, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3), iters=5, warm=2)
    res["convert_yuv_4k_420"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "GB/s": round(3.0 * w * h / (ms / 1e3) / 1e9, 1)}
    qt = u.quant_table(95, False)
    plane = sdr.buf[: sdr.layout[0][0] * sdr.layout[0][1]]
    coef = torch.empty((h // 8, w // 8, 64), dtype=torch.int16, device=device)
    ms = time_kernel(ctx, lambda: u.fdct_quant(plane, sdr.layout[0][1], w // 8, h // 8, qt, coef), iters=5, warm=2)
    res["fdct_quant_4k_luma"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "GB/s": round(3.0 * w * h / (ms / 1e3) / 1e9, 1)}
    dec_plane = torch.empty((h, w), dtype=torch.uint8, device=device)
    ms = time_kernel(ctx, lambda: u.idct_dequant(coef, qt, plane=dec_plane, stride=w), iters=5, warm=2)
    res["idct_dequant_4k_luma"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "GB/s": round(3.0 * w * h / (ms / 1e3) / 1e9, 1)}

    # entropy stage: the 4K base frame's quantized coefficients (q 95, 4:2:0) -> entropy-coded bytes
    hq = [u.quant_table(95, False), u.quant_table(95, True), u.quant_table(95, True)]
    hco = []
    for c in range(3):
        rows, stride, wv = sdr.layout[c]
        hco.append(u.fdct_quant(sdr.plane_tensor(c), stride, wv // 8, rows // 8, hq[c]))
    hout = torch.empty(w * h * 2, dtype=torch.uint8, device=device)
    nbytes = [0]

    def huff():
        nbytes[0] = int(u.huffman_encode(hco, w, h, [(2, 2), (1, 1), (1, 1)], 10, out=hout).numel())

    ms = time_kernel(ctx, huff, iters=5, warm=2)
    res["huffman_encode_4k_420_q95"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "jpeg_scan_bytes": nbytes[0],
                                        "GB/s_coef_in": round(3.0 * w * h / (ms / 1e3) / 1e9, 1),
                                        "stages": "one wavefront per restart interval (10 MCUs; two launches = two LDS size classes) -> interval sizes -> offsets -> gather with RSTn markers"}
    # ... the stream the reference itself writes: no restart markers (three passes over all blocks)
    try:
        def huff0():
            nbytes[0] = int(u.huffman_encode(hco, w, h, [(2, 2), (1, 1), (1, 1)], 0, out=hout).numel())

        ms = time_kernel(ctx, huff0, iters=5, warm=2)
        res["huffman_encode_4k_420_q95_no_restart_markers"] = {
            "us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "jpeg_scan_bytes": nbytes[0],
            "stages": "code lengths per block (DC difference from the neighbour's DC) -> scan of the bit lengths -> emit at bit offsets -> 0xFF count, scan, stuff"}
    except Exception as e:  # noqa: BLE001
        res["huffman_encode_4k_420_q95_no_restart_markers"] = {"error": f"{type(e).__name__}: {e}"}
    # ... and back: the same frame with restart markers -> coefficients (long intervals: the parallel decoder with the markers
    # spliced out; short ones: one lane per interval)
    try:
        for ri_ in (10, 2):
            stream = u.huffman_encode(hco, w, h, [(2, 2), (1, 1), (1, 1)], ri_, out=hout).clone()
            shp = [tuple(c.shape[:2]) for c in hco]
            ms = time_kernel(ctx, lambda: u.huffman_decode(stream, shp, w, h, [(2, 2), (1, 1), (1, 1)], ri_), iters=3, warm=1)
            res[f"huffman_decode_4k_420_q95_ri{ri_}"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                                          "jpeg_scan_bytes": int(stream.numel()),
                                                          "stages": ("unstuff + RSTn markers spliced out -> the self-synchronising decoder of the marker-less case, hopping over "
                                                                     "the padding bits at the flagged interval starts -> DC scan per interval (DESIGN.md 5.5)") if ri_ == 10 else
                                                                    "marker count -> scan -> interval table -> decode (one lane per restart interval: short intervals)"}
    except Exception as e:  # noqa: BLE001  (a failure here must not cost the other stage measurements)
        res["huffman_decode_4k_420_q95"] = {"error": f"{type(e).__name__}: {e}"}
    # ... and a scan WITHOUT restart markers, as every file of the reference has it: the primary image of an UltraHDR file
    # written through the drop-in facade (the reference's own libjpeg Huffman pass), parsed by uhdr_hip_jpeg_parse
    try:
        from libultrahdr_amd import facade as FA
        if FA.available():
            jpg = FA.encode(synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG), synth.make_sdr_yuv420(w, h), gpu=True)
            hd = u.jpeg_parse(jpg)
            sc = hd.scan
            data = torch.from_numpy(np.frombuffer(jpg, dtype=np.uint8)[hd.scan_offset: hd.scan_offset + hd.scan_bytes].copy()).to(device)
            bits = np.frombuffer(hd.tables.bits, dtype=np.uint8).reshape(4, 17)
            vals = np.frombuffer(hd.tables.vals, dtype=np.uint8).reshape(4, 256)
            shp0 = [(sc.blocks_h[c], sc.blocks_w[c]) for c in range(3)]
            ms = time_kernel(ctx, lambda: u.huffman_decode(data, shp0, sc.w, sc.h, [(2, 2), (1, 1), (1, 1)], 0, tables=(bits, vals)), iters=5, warm=2)
            res["huffman_decode_4k_420_q95_no_restart_markers"] = {
                "us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "jpeg_scan_bytes": int(hd.scan_bytes),
                "stages": "unstuff -> one decode per possible block position (6 hypotheses x 1024-bit subsequences) -> overflow until the paths merge "
                          "-> true path by a scan over map composition -> write pass -> DC scan (DESIGN.md 5.5)"}
            # ... the same file's gain-map JPEG: three channels, 4:4:4, full resolution (the C API's default encode) -- sparse blocks,
            # long synchronisation distances (DESIGN.md 5.5 "Overflow depth")
            cut = jpg.rfind(b"\xff\xd8\xff")
            gmj = jpg[cut:]
            hm = u.jpeg_parse(gmj)
            scm = hm.scan
            datam = torch.from_numpy(np.frombuffer(gmj, dtype=np.uint8)[hm.scan_offset: hm.scan_offset + hm.scan_bytes].copy()).to(device)
            bitsm = np.frombuffer(hm.tables.bits, dtype=np.uint8).reshape(4, 17)
            valsm = np.frombuffer(hm.tables.vals, dtype=np.uint8).reshape(4, 256)
            shpm = [(scm.blocks_h[c], scm.blocks_w[c]) for c in range(scm.num_components)]
            smp = [(scm.h_samp[c], scm.v_samp[c]) for c in range(scm.num_components)]
            ms = time_kernel(ctx, lambda: u.huffman_decode(datam, shpm, scm.w, scm.h, smp, 0, tables=(bitsm, valsm)), iters=5, warm=2)
            res["huffman_decode_4k_444_map_no_restart_markers"] = {
                "us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1), "jpeg_scan_bytes": int(hm.scan_bytes),
                "bits_per_block": round(hm.scan_bytes * 8.0 / sum(a_ * b_ for a_, b_ in shpm), 1)}
            # ... and uhdr_hip_jpeg_decode_scan as the facade calls it (host bytes in, lazy download: samples stay on the device):
            # wall time of the whole call against the kernel time inside it -- the shim's own host overhead
            for nm_, file_, rgb_ in (("base", jpg[:cut], 0), ("map3ch", gmj, 4)):
                u.lib.uhdr_hip_resident_begin(ctx.handle)
                u.lib.uhdr_hip_resident_lazy(ctx.handle, 1)
                outs_ = u.jpeg_decode(file_, rgb_)
                outs_ = outs_ if isinstance(outs_, list) else [outs_]
                clock_ramp(ctx, lambda: u.jpeg_decode(file_, rgb_, outs=outs_), seconds=0.3)
                walls_, cwalls_ = [], []
                stx_ = A.Stats()
                for _ in range(7):  # (no per-launch events here: they cost the call host time)
                    t0_ = time.perf_counter()
                    u.jpeg_decode(file_, rgb_, outs=outs_)
                    walls_.append(time.perf_counter() - t0_)
                    u.lib.uhdr_hip_get_stats(ctx.handle, C.byref(stx_))
                    cwalls_.append(stx_.last_jpeg_decode_scan_ns * 1e-9)
                ctx.profile(True)
                ctx.profile_read(None, reset=True)
                for _ in range(3):
                    u.jpeg_decode(file_, rgb_, outs=outs_)
                n_, kms_ = ctx.profile_read(None, reset=True)
                ctx.profile(False)
                u.lib.uhdr_hip_resident_end(ctx.handle)
                walls_.sort()
                cwalls_.sort()
                res[f"jpeg_decode_scan_4k_{nm_}_lazy"] = {"c_call_wall_us": round(cwalls_[len(cwalls_) // 2] * 1e6, 1), "kernels_us": round(kms_ / 3 * 1e3, 1),
                                                          "python_wrapper_wall_us": round(walls_[len(walls_) // 2] * 1e6, 1), "launches": n_ // 3, "file_bytes": len(file_),
                                                          "note": "c_call_wall_us: steady_clock inside uhdr_hip_jpeg_decode_scan (uhdr_hip_stats_t::last_jpeg_decode_scan_ns), what the facade pays; "
                                                                  "python_wrapper_wall_us adds ctypes marshalling, the header parse and the stream ordering of the Python binding"}
    except Exception as e:  # noqa: BLE001
        res["huffman_decode_4k_420_q95_no_restart_markers"] = {"error": f"{type(e).__name__}: {e}"}
    del hco, hout

    # ---- whole stage chains, device resident (sum of the kernels' HIP-event durations per pass) ----------------
    def blocks(n):
        return (n + 7) // 8

    def fdct_planes(img, planes, tables):
        for c in planes:
            rows, stride, wv = img.layout[c]
            u.fdct_quant(img.plane_tensor(c), stride, wv // 8, rows // 8, tables[c])

    qy, qc = u.quant_table(95, False), u.quant_table(95, True)
    # (the API-1 4K and API-0 8K encode chains live in encode_section: they carry their own roofline objects)
    base = sdr.clone()
    # (3) decode chain, 4K: coefficient blocks -> IDCT (Y, Cb, Cr, Y400 map s=4) -> applyGainMap -> F16
    dsdr = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=64, device=device)
    mw, mh = w // 4, (h // 4 + 7) // 8 * 8  # 960 x 544: the block grid of the 960 x 540 map
    dgm = Image(A.UHDR_IMG_FMT_8bppYCbCr400, mw, mh, A.UHDR_CG_BT_2100, align=64, device=device)
    cf = [torch.zeros((blocks(r), blocks(wv), 64), dtype=torch.int16, device=device) for (r, _, wv) in dsdr.layout]
    cfm = torch.zeros((mh // 8, mw // 8, 64), dtype=torch.int16, device=device)
    for t in cf + [cfm]:
        t[..., 0] = 37  # a DC-only image is as expensive as any other for this kernel
    dst4 = Image(f16, w, h, align=64, device=device)
    dgm_view = Image(A.UHDR_IMG_FMT_8bppYCbCr400, mw, h // 4, A.UHDR_CG_BT_2100, align=64, device=device)

    def decode_chain():
        for c in range(3):
            rows, stride, wv = dsdr.layout[c]
            u.idct_dequant(cf[c], qy if c == 0 else qc, plane=dsdr.plane_tensor(c), stride=stride)
        u.idct_dequant(cfm, qy, plane=dgm.plane_tensor(0), stride=dgm.layout[0][1])
        dgm_view.raw.planes[0] = dgm.raw.planes[0]
        u.applyGainMap(dsdr, dgm_view, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst4)

    ms = time_kernel(ctx, decode_chain, iters=5, warm=2)
    res["decode_chain_4k_idct_plus_apply_mapA"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                                    "stages": "idct_dequant(Y, Cb, Cr, Y400 map) + applyGainMap -> RGBA_F16; Huffman decode not included"}
    # (3b) the same with a full-resolution 3-channel map (what the default encoder settings produce): the map's
    #      three coefficient planes -> RGBA8888 either in one pass or as three IDCTs + ycc_rgb_convert
    rgba = A.UHDR_IMG_FMT_32bppRGBA8888
    cf3 = [torch.zeros((h // 8, w // 8, 64), dtype=torch.int16, device=device) for _ in range(3)]
    for t in cf3:
        t[..., 0] = 37
    gm3 = Image(rgba, w, h, A.UHDR_CG_BT_2100, align=64, device=device)
    ycc3 = Image(A.UHDR_IMG_FMT_24bppYCbCr444, w, h, align=64, device=device)
    ms_f = time_kernel(ctx, lambda: u.idct_dequant_rgb(cf3, qy, qc, w, h, rgba, 0, dst=gm3), iters=5, warm=2)
    res["idct_dequant_rgb_4k_fused"] = {"us": round(ms_f * 1e3, 1), "Mpx/s": round(w * h / (ms_f / 1e3) / 1e6, 1),
                                        "GB/s_10B_per_px": round(10.0 * w * h / (ms_f / 1e3) / 1e9, 1)}

    def map_four_step():
        for c in range(3):
            u.idct_dequant(cf3[c], qy if c == 0 else qc, plane=ycc3.plane_tensor(c), stride=ycc3.layout[c][1])
        A.check(u.lib.uhdr_hip_jpeg_ycc_to_rgb_dev(ctx.handle, C.byref(ycc3.raw), 0, C.byref(gm3.raw)))

    ms_u = time_kernel(ctx, map_four_step, iters=5, warm=2)
    res["idct_dequant_rgb_4k_four_step"] = {"us": round(ms_u * 1e3, 1), "Mpx/s": round(w * h / (ms_u / 1e3) / 1e6, 1),
                                            "GB/s_16B_per_px": round(16.0 * w * h / (ms_u / 1e3) / 1e9, 1)}

    def decode_chain_c():
        for c in range(3):
            rows, stride, wv = dsdr.layout[c]
            u.idct_dequant(cf[c], qy if c == 0 else qc, plane=dsdr.plane_tensor(c), stride=stride)
        u.idct_dequant_rgb(cf3, qy, qc, w, h, rgba, 0, dst=gm3)
        u.applyGainMap(dsdr, gm3, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst4)

    # (3c) SURVEY 8f-1 as worded: the base image's dequant + IDCT inside the applyGainMap kernel
    qts = [qy, qc, qc]

    def fused_chain_a():
        u.idct_dequant(cfm, qy, plane=dgm.plane_tensor(0), stride=dgm.layout[0][1])
        dgm_view.raw.planes[0] = dgm.raw.planes[0]
        u.applyGainMapFromCoefficients(cf, qts, w, h, A.UHDR_CG_BT_709, dgm_view, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst4)

    ms = time_kernel(ctx, fused_chain_a, iters=5, warm=2)
    res["decode_chain_4k_apply_from_coefficients_mapA"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                                            "stages": "idct_dequant(Y400 map) + applyGainMap with the base image's dequant + IDCT inside the kernel"}

    def fused_chain_c():
        u.idct_dequant_rgb(cf3, qy, qc, w, h, rgba, 0, dst=gm3)
        u.applyGainMapFromCoefficients(cf, qts, w, h, A.UHDR_CG_BT_709, gm3, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst4)

    ms = time_kernel(ctx, fused_chain_c, iters=5, warm=2)
    res["decode_chain_4k_apply_from_coefficients_mapC"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                                            "stages": "idct_dequant_rgb(3-ch map) + applyGainMap with the base image's dequant + IDCT inside the kernel"}
    ms = time_kernel(ctx, decode_chain_c, iters=5, warm=2)
    res["decode_chain_4k_idct_plus_apply_mapC"] = {"us": round(ms * 1e3, 1), "Mpx/s": round(w * h / (ms / 1e3) / 1e6, 1),
                                                    "stages": "idct_dequant(Y, Cb, Cr) + idct_dequant_rgb(3-ch map) + applyGainMap -> RGBA_F16; Huffman decode not included"}
    return res


def cpu_baseline(w, h, budget_s, stages=True):
    """The reference's CPU path for the SAME workload on this box's host cores, bounded sample, rank 0 only: whole uhdr_encode
    (API-1, ultrahdr_api.cpp:1200) + uhdr_decode (-> RGBA_F16 linear, ultrahdr_api.cpp:1918) calls of the real reference
    (oracle/_ref: libultrahdr built from its own sources) on host buffers, timed around the C calls.  Without oracle/_ref the C
    port has no codec layer: it then times generateGainMap + applyGainMap only and says so."""
    import ctypes
    import numpy as np

    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from oracle import loader as L

    sdr = synth.make_sdr_yuv420(w, h, seed=1234)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG, seed=1234)
    px = w * h
    if L.ref() is not None:
        lib = L.ref()
        cap = px * 6
        buf = (ctypes.c_uint8 * cap)()
        dest = np.empty((h, w, 8), dtype=np.uint8)
        ow, oh = ctypes.c_int(0), ctypes.c_int(0)
        f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat

        def trip():
            t0 = time.perf_counter()
            n = lib.ref_uhdr_encode(ctypes.byref(hdr.raw), ctypes.byref(sdr.raw), 95, A.UHDR_USAGE_BEST_QUALITY, buf, cap)
            t1 = time.perf_counter()
            if n <= 0:
                raise RuntimeError(f"ref_uhdr_encode failed: {n}")
            rc = lib.ref_uhdr_decode(buf, n, A.UHDR_CT_LINEAR, f16, dest.ctypes.data, dest.nbytes, ctypes.byref(ow), ctypes.byref(oh))
            t2 = time.perf_counter()
            if rc != 0:
                raise RuntimeError(f"ref_uhdr_decode failed: {rc}")
            return t1 - t0, t2 - t1, n

        trip()  # warm-up: the reference builds its static LUTs on first use
        te, td, n, nbytes, t_start = 0.0, 0.0, 0, 0, time.perf_counter()
        while True:
            e_, d_, nbytes = trip()
            te, td, n = te + e_, td + d_, n + 1
            if time.perf_counter() - t_start >= budget_s or n >= 32:
                break
        cores = min(os.cpu_count() or 1, 4)
        res = {"value": round(n * px / (te + td) / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "reference",
               "sample": f"{n} x (uhdr_encode API-1 q95 + uhdr_decode -> RGBA_F16) of the same {w}x{h} frame on host buffers: "
                         f"encode {te / n * 1e3:.0f} ms + decode {td / n * 1e3:.0f} ms per frame, {nbytes} byte file; libultrahdr built from /root/reference "
                         f"(oracle/_ref), its own min(hw,4)-thread job queue, libjpeg on one thread; host has {os.cpu_count()} logical cores",
               "encode_ms": round(te / n * 1e3, 1), "decode_ms": round(td / n * 1e3, 1)}
        if stages:
            try:
                res["stages"] = cpu_stage_baselines(w, h)
            except Exception as e:  # noqa: BLE001
                res["stages"] = {"error": f"{type(e).__name__}: {e}"}
        return res
    cfg = A.default_encode_cfg()
    md_, gm = L.generate_gainmap("port", sdr, hdr, cfg)
    t0 = time.perf_counter()
    md_, gm = L.generate_gainmap("port", sdr, hdr, cfg)
    L.apply_gainmap("port", sdr, gm, md_, A.UHDR_CT_LINEAR)
    el = time.perf_counter() - t0
    return {"value": round(px / el / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"1 x (generateGainMap + applyGainMap) {w}x{h}, single-threaded C restatement (oracle/uhdr_oracle.c), JPEG stages NOT included "
                      f"(oracle/_ref absent): {el:.1f} s"}


def cpu_stage_baselines(w, h):
    """the reference's stage calls and its whole uhdr_encode / uhdr_decode on identical host buffers,
    on this box's host cores (the reference uses min(N, 4) threads for the pixel stages and one thread for libjpeg,
    convertYuv and the converters, jpegr.cpp:739).  Bounded: one warm-up + 1-3 timed calls per stage, median."""
    import numpy as np

    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from oracle import loader as L

    px = w * h
    sdr = synth.make_sdr_yuv420(w, h)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
    out = {}

    def timed(name, fn, reps, note=None, warm=False):
        if warm:  # the reference builds its static LUTs on first use
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        out[name] = {"ms": round(t * 1e3, 1), "Mpx/s": round(px / t / 1e6, 1)}
        if note:
            out[name]["note"] = note
        return r

    cfg2 = A.default_encode_cfg()  # C-API defaults: two passes, 3 channels, scale 1
    timed("toneMap_p010_to_420", lambda: L.tone_map("ref", hdr), 2, warm=True)
    _, gm = timed("generateGainMap_2pass_3ch_s1", lambda: L.generate_gainmap("ref", sdr, hdr, cfg2), 1, warm=True)
    timed("convertYuv_709_to_p3", lambda: L.convert_yuv("ref", sdr, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3), 3, "single-threaded in the reference")
    base_jpg = timed("compressImage_base_420_q95", lambda: L.ref_jpeg_compress(sdr, 95), 2, "libjpeg, one thread")
    map_jpg = timed("compressImage_map_rgb888_q95", lambda: L.ref_jpeg_compress(gm, 95), 1, "libjpeg, one thread")
    timed("decompressImage_base", lambda: L.ref_jpeg_decompress(base_jpg, 0), 2, "libjpeg, one thread")
    timed("decompressImage_map", lambda: L.ref_jpeg_decompress(map_jpg, 1), 1, "libjpeg, one thread")
    jpg = timed("uhdr_encode_api1", lambda: L.ref_uhdr_encode(hdr, sdr), 1, "whole C API call: ultrahdr_api.cpp:1200")
    dest = np.empty((h, w, 8), dtype=np.uint8)
    timed("uhdr_decode_to_f16", lambda: L.ref_uhdr_decode(jpg, A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, dest), 2,
          "whole C API call: ultrahdr_api.cpp:1918")
    out["image"] = f"{w}x{h} P010 (BT.2100 HLG) + YCbCr 4:2:0 (BT.709), synthetic seed 1234"
    return out


if __name__ == "__main__":
    main()
