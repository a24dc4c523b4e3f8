"""Row-stripe sharding of one image across the ranks of a node.

Every stage of the hot path computes an output pixel / block from a bounded input neighbourhood,
so an image splits into independent horizontal stripes -- one per rank / GPU -- with NO data-path
collective.  The single exception is two-pass gain-map generation, whose only cross-stripe
dependency is the global per-channel min/max of the log2 gain (the mutex-guarded merge at
/root/reference/lib/src/jpegr.cpp:932-938).  That becomes ONE tiny all-reduce (MIN over the 6 floats
{min0..2, -max0..2}; RCCL over xGMI on GPUs, gloo in the CPU tests): latency-bound, never bandwidth-bound.

Stripe boundaries must be multiples of lcm(2 (4:2:0 chroma rows), scale (box sampling), 16 (4:2:0
MCU rows of the JPEG stage)); ``partition_rows`` takes that granule.
"""
from __future__ import annotations

import ctypes as C
import math

from . import capi as A


def stripe_granule(scale: int = 1, mcu_rows: int = 16) -> int:
    return math.lcm(2, max(1, scale), mcu_rows)


def partition_rows(height: int, world_size: int, granule: int = 16):
    """[(row0, rows)] per rank: contiguous stripes, sizes multiples of ``granule`` except possibly
    the last; earlier ranks get the extra granules.  Ranks beyond the available granules get 0 rows."""
    if height <= 0 or world_size <= 0:
        raise ValueError("height and world_size must be positive")
    units = (height + granule - 1) // granule
    base, extra = divmod(units, world_size)
    out, row = [], 0
    for r in range(world_size):
        n = (base + (1 if r < extra else 0)) * granule
        n = max(0, min(n, height - row))
        out.append((row, n))
        row += n
    assert row == height
    return out


def merge_minmax(parts):
    """Sequential merge of per-stripe {min0,min1,min2,max0,max1,max2} lists -- what the reference's mutex-guarded
    merge does (jpegr.cpp:932-938) and what the all-reduce below computes across ranks."""
    out = [127.0, 127.0, 127.0, -128.0, -128.0, -128.0]
    for p in parts:
        for i in range(3):
            out[i] = min(out[i], float(p[i]))
            out[3 + i] = max(out[3 + i], float(p[3 + i]))
    return out


def allreduce_minmax(minmax, group=None):
    """In-place all-reduce of a 6-element float32 tensor {min0,min1,min2,max0,max1,max2} with ONE collective:
    MIN over {min0, min1, min2, -max0, -max1, -max2} (negation is exact, so max == -min(-x) bit for bit).  Works on
    CPU tensors (gloo) and CUDA tensors (backend "nccl" == RCCL on ROCm).  Float min/max is order-independent, so
    the result is bit-identical to the reference's sequential merge.  (The C++ host layer does the same on the
    context's own stream without leaving the device: generate_gainmap_striped below.)"""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return minmax
    t = torch.cat([minmax[:3], -minmax[3:]])
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    minmax[:3] = t[:3]
    minmax[3:] = -t[3:]
    return minmax


def finalize_minmax(cfg: A.EncodeCfg, hdr_ct: int, use_base_cg: int, minmax6):
    """Host half of two-pass generation (clamp, user hints, epsilon guard, metadata fill:
    jpegr.cpp:969-986, 1031-1048) through the C ABI; pure host code, needs no GPU.
    Returns (finalized [6] list, GainmapMetadata)."""
    lib = A.load()
    mm = (C.c_float * 6)(*[float(v) for v in minmax6])
    md = A.GainmapMetadata()
    A.check(lib.uhdr_hip_generate_gainmap_finalize(C.byref(cfg), hdr_ct, use_base_cg, mm, C.byref(md)))
    return list(mm), md


def generate_gainmap_two_pass_striped(uhdr, sdr_stripe, hdr_stripe, cfg: A.EncodeCfg, gm_stripe, group=None):
    """Two-pass generateGainMap for THIS rank's stripe (device images).  ``gm_stripe`` is the
    destination stripe of the gain map (device Image, rows = stripe rows / scale).
    pass 1 (kernel) -> all-reduce(min/max) -> finalize (host) -> pass 2 (kernel)."""
    import torch

    dev = sdr_stripe.buf.device
    nch = 3 if cfg.use_multi_channel_gainmap else 1
    s = cfg.map_dimension_scale_factor
    mw, mh = sdr_stripe.w // s, sdr_stripe.h // s
    gains = torch.empty(max(mw * mh * nch, 1), dtype=torch.float32, device=dev)
    mm = torch.tensor([127.0] * 3 + [-128.0] * 3, dtype=torch.float32, device=dev)  # the identity of the merge
    torch.cuda.current_stream(dev).synchronize()
    # a rank without a map row never runs pass 1: derive use_base_cg from the gamuts with the rule pass 1 uses
    # (jpegr.cpp:605-638), so that the metadata is the same on every rank
    s_cg, h_cg = sdr_stripe.raw.cg, hdr_stripe.raw.cg
    ubc = C.c_int(1 if (s_cg == h_cg or not (h_cg == A.UHDR_CG_BT_2100 or (h_cg == A.UHDR_CG_DISPLAY_P3 and s_cg != A.UHDR_CG_BT_2100))) else 0)
    lib, h = uhdr.lib, uhdr.ctx.handle
    # a last stripe with fewer rows than the scale factor holds no map row (the whole image's map has H // s rows): it
    # launches nothing and contributes the identity; uhdr_hip_generate_gainmap_pass1_dev refuses such a stripe
    if mh > 0:
        A.check(lib.uhdr_hip_generate_gainmap_pass1_dev(h, C.byref(sdr_stripe.raw), C.byref(hdr_stripe.raw), C.byref(cfg),
                                                        C.c_void_p(gains.data_ptr()), C.c_void_p(mm.data_ptr()), C.byref(ubc)))
    uhdr.ctx.synchronize()  # mm was produced on the context's stream; the collective runs on torch's
    allreduce_minmax(mm, group)
    fin, md = finalize_minmax(cfg, hdr_stripe.raw.ct, ubc.value, mm.cpu().tolist())
    gm_stripe.raw.w, gm_stripe.raw.h = mw, mh
    if mh > 0:
        A.check(lib.uhdr_hip_generate_gainmap_pass2_dev(h, C.c_void_p(gains.data_ptr()), (C.c_float * 6)(*fin), C.byref(cfg),
                                                        C.byref(gm_stripe.raw)))
    return md


# ---- the same through the C++ host layer: RCCL on the library's own stream, no host round trip -----------------------
def init_comm(ctx, group=None):
    """Give ``ctx`` (libultrahdr_amd.ultrahdr.Context) an RCCL communicator spanning the torch.distributed group
    (one process per GPU): rank 0 draws the NCCL unique id, torch.distributed carries the 128 bytes to the other
    ranks (any bootstrap would do), every rank calls uhdr_hip_comm_init.  World size 1 (no process group): a
    one-rank communicator, so the collective still executes.  Returns ncclCommCount."""
    import torch
    import torch.distributed as dist

    lib = ctx.lib
    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    ident = (C.c_ubyte * 128)()
    if rank == 0 and lib.uhdr_hip_comm_unique_id(ident) != 0:
        raise RuntimeError("RCCL is not available in this process")
    if world > 1:
        backend = dist.get_backend(group)
        t = torch.tensor(list(ident), dtype=torch.uint8, device="cuda" if backend == "nccl" else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = (C.c_ubyte * 128)(*t.cpu().tolist())
    A.check(lib.uhdr_hip_comm_init(ctx.handle, ident, rank, world))
    return int(lib.uhdr_hip_comm_size(ctx.handle))


def all_reduce_probe(ctx):
    """One min-all-reduce of six floats on the library's stream (uhdr_hip_comm_all_reduce_min_dev): the exchange step of the
    striped encode by itself, for latency measurements."""
    import torch

    buf = getattr(ctx, "_probe_buf", None)
    if buf is None:
        buf = ctx._probe_buf = torch.zeros(6, dtype=torch.float32, device=f"cuda:{torch.cuda.current_device()}")
        torch.cuda.synchronize()
    A.check(ctx.lib.uhdr_hip_comm_all_reduce_min_dev(ctx.handle, C.c_void_p(buf.data_ptr()), 6))


def init_comm_relay(ctx, group=None):
    """Give ``ctx`` a HOST-RELAY transport over the torch.distributed group (any backend, gloo included) through
    uhdr_hip_comm_init_custom: every exchange step copies its device buffer to the host on the library's stream, runs the
    torch.distributed collective there and copies the result back.  For set-ups where RCCL cannot form the communicator
    -- several ranks sharing one GPU (tests, the bench's one-GPU dry run of the N > 1 path) -- and as the reference
    implementation of the transport interface; RCCL over xGMI (init_comm) is the product path.  Returns the world size.
    The callbacks are kept alive on ``ctx``."""
    import torch
    import torch.distributed as dist

    lib = ctx.lib
    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1

    def _sync(stream):
        torch.cuda.synchronize()  # the library's stream is a non-blocking HIP stream of this device: drain everything

    def _view(ptr, nbytes):
        # a uint8 CUDA tensor over raw device memory (no copy): torch's __cuda_array_interface__ protocol
        class _Raw:
            __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_Raw(), device="cuda")

    def all_reduce_min(_user, buf, n, stream):
        try:
            _sync(stream)
            dev = _view(buf, n * 4).view(torch.float32)
            t = dev.cpu()
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            dev.copy_(t)
            _sync(stream)
            return 0
        except Exception:  # never let an exception cross the C boundary
            return 1

    def all_gather(_user, send, recv, nbytes, stream):
        try:
            _sync(stream)
            t = _view(send, nbytes).cpu()
            parts = [torch.empty_like(t) for _ in range(world)]
            if world > 1:
                dist.all_gather(parts, t, group=group)
            else:
                parts = [t]
            _view(recv, nbytes * world).copy_(torch.cat(parts))
            _sync(stream)
            return 0
        except Exception:
            return 1

    def gather_v(_user, send, send_bytes, recv, counts, root, stream):
        try:
            _sync(stream)
            cnt = [int(counts[i]) for i in range(world)]
            mx = max(max(cnt), 1)
            t = torch.zeros(mx, dtype=torch.uint8)
            if send_bytes:
                t[:send_bytes] = _view(send, send_bytes).cpu()
            parts = [torch.empty_like(t) for _ in range(world)]
            if world > 1:
                dist.all_gather(parts, t, group=group)  # gloo has no gather-v; the relay is not the performance path
            else:
                parts = [t]
            if rank == root and sum(cnt):
                _view(recv, sum(cnt)).copy_(torch.cat([p_[:n] for p_, n in zip(parts, cnt)]))
            _sync(stream)
            return 0
        except Exception:
            return 1

    ops = A.CommOps(None, A.ALL_REDUCE_MIN_FN(all_reduce_min), A.ALL_GATHER_FN(all_gather), A.GATHER_V_FN(gather_v))
    ctx._comm_ops = ops  # keep the ctypes callbacks alive as long as the context
    A.check(lib.uhdr_hip_comm_init_custom(ctx.handle, C.byref(ops), rank, world))
    return int(lib.uhdr_hip_comm_size(ctx.handle))


def gather_rows_to_root(ctx, stripe, counts_rows, row_bytes: int, root: int = 0):
    """Device gather of row stripes of unequal height (uhdr_hip_comm_gather_dev: one group of ncclSend / ncclRecv over
    xGMI, no host staging): ``stripe`` = this rank's rows as a contiguous uint8 CUDA tensor [rows, row_bytes] (rows may be 0),
    ``counts_rows`` = rows per rank.  Returns the whole [sum(rows), row_bytes] tensor on ``root``, None elsewhere."""
    import torch

    lib = ctx.lib
    world = max(1, int(lib.uhdr_hip_comm_size(ctx.handle)))
    rank = int(lib.uhdr_hip_comm_rank(ctx.handle)) if world > 1 else 0
    assert len(counts_rows) == world and stripe.is_contiguous()
    counts = (C.c_size_t * world)(*[int(r) * row_bytes for r in counts_rows])
    out = torch.empty((sum(counts_rows), row_bytes), dtype=torch.uint8, device=stripe.device) if rank == root else None
    with ctx.ordered():
        A.check(lib.uhdr_hip_comm_gather_dev(ctx.handle, C.c_void_p(stripe.data_ptr() if stripe.numel() else 0), stripe.numel(),
                                             C.c_void_p(out.data_ptr() if out is not None else 0), counts, root))
    return out


def gather_streams_to_root(ctx, stream_bytes, root: int = 0):
    """Per-stripe entropy-coded streams (uint8 CUDA tensors of different lengths) to ``root``: the sizes travel in one
    uhdr_hip_comm_all_gather_dev (8 bytes per rank), the bytes in one uhdr_hip_comm_gather_dev.  Returns the list of the
    ranks' streams (CUDA tensors) on ``root``, None elsewhere -- stitch_entropy_streams joins them."""
    import torch

    lib = ctx.lib
    world = max(1, int(lib.uhdr_hip_comm_size(ctx.handle)))
    rank = int(lib.uhdr_hip_comm_rank(ctx.handle)) if world > 1 else 0
    dev = stream_bytes.device
    n_mine = torch.tensor([stream_bytes.numel()], dtype=torch.int64, device=dev)
    sizes = torch.empty(world, dtype=torch.int64, device=dev)
    with ctx.ordered():
        A.check(lib.uhdr_hip_comm_all_gather_dev(ctx.handle, C.c_void_p(n_mine.data_ptr()), C.c_void_p(sizes.data_ptr()), 8))
    ctx.synchronize()
    cnt = [int(v) for v in sizes.cpu().tolist()]
    counts = (C.c_size_t * world)(*cnt)
    out = torch.empty(max(sum(cnt), 1), dtype=torch.uint8, device=dev) if rank == root else None
    with ctx.ordered():
        A.check(lib.uhdr_hip_comm_gather_dev(ctx.handle, C.c_void_p(stream_bytes.data_ptr() if cnt[rank] else 0), cnt[rank],
                                             C.c_void_p(out.data_ptr() if out is not None else 0), counts, root))
    ctx.synchronize()
    if rank != root:
        return None
    parts, off = [], 0
    for n in cnt:
        parts.append(out[off: off + n])
        off += n
    return parts


def generate_gainmap_striped(uhdr, sdr_stripe, hdr_stripe, cfg: A.EncodeCfg, gm_stripe):
    """Two-pass generateGainMap for THIS rank's stripe, entirely inside the C++ host layer
    (uhdr_hip_generate_gainmap_striped_dev): pass 1 -> ncclAllReduce(min) over {min0..2, -max0..2} on the context's
    stream -> finalisation on the device -> pass 2; one host synchronisation at the end.  Device images.  Returns the
    metadata (identical on every rank)."""
    md = A.GainmapMetadata()
    with uhdr.ctx.ordered():
        A.check(uhdr.lib.uhdr_hip_generate_gainmap_striped_dev(uhdr.ctx.handle, C.byref(sdr_stripe.raw), C.byref(hdr_stripe.raw),
                                                               C.byref(cfg), C.byref(md), C.byref(gm_stripe.raw)))
    return md


# ---- entropy stage across ranks (last bullet) ---------------------------------------------
# With restart intervals every interval is independent, so a rank can entropy-code the MCU rows of its own stripe and
# the compressed streams only have to be concatenated.  Two conditions make the concatenation EQUAL to the single-rank
# stream: a stripe holds a whole number of intervals, and (because RSTn markers are numbered modulo 8 from the start of
# the scan, T.81 E.1.4) every stripe but the last holds a multiple of 8 of them -- then each rank's own numbering,
# starting at RST0, is already the global one and the marker that joins two stripes is always RST7.
def entropy_stripe_plan(mcu_rows: int, mcus_per_row: int, restart_interval: int, world_size: int):
    """[(mcu_row0, n_mcu_rows)] per rank for rank-local Huffman coding, or ValueError when the geometry does not allow
    it.  Stripes are multiples of the smallest row count g with (g * mcus_per_row) % (8 * restart_interval) == 0."""
    if min(mcu_rows, mcus_per_row, restart_interval, world_size) <= 0:
        raise ValueError("all arguments must be positive")
    g = (8 * restart_interval) // math.gcd(8 * restart_interval, mcus_per_row)
    units = mcu_rows // g  # whole granules; the remainder rows go to the last non-empty stripe
    if units == 0:
        return [(0, mcu_rows)] + [(mcu_rows, 0)] * (world_size - 1)
    base, extra = divmod(units, world_size)
    plan, row = [], 0
    for r in range(world_size):
        n = (base + (1 if r < extra else 0)) * g
        plan.append([row, n])
        row += n
    last = max(i for i, (_, n) in enumerate(plan) if n > 0)
    plan[last][1] += mcu_rows - row
    for i in range(last + 1, world_size):
        plan[i][0] = mcu_rows
    return [tuple(p) for p in plan]


def stitch_entropy_streams(streams) -> bytes:
    """Concatenates per-stripe entropy-coded streams (in stripe order, empty ones skipped) with the RST7 markers that
    sit between them in the single-rank stream (see entropy_stripe_plan for why it is always RST7)."""
    parts = [bytes(s) for s in streams if len(s)]
    return b"\xff\xd7".join(parts)


def stripe_scan_slices(shapes, sampling, w: int, h: int, mcu_row0: int, n_mcu_rows: int):
    """Geometry of one stripe as its own scan: per component (block_row0, block_rows) into the [blocks_h, blocks_w, 64]
    arrays (clipped to the real grid: what lies below it are libjpeg's dummy blocks, produced by the encoder), and
    the pixel height to put in the stripe's scan description."""
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    if len(shapes) == 1:  # non-interleaved: an MCU is a block
        vmax = 1
    px0 = mcu_row0 * 8 * vmax
    px_h = min(n_mcu_rows * 8 * vmax, h - px0)
    out = []
    for (bh, _bw), (_hs, vs) in zip(shapes, sampling):
        v = vs if len(shapes) > 1 else 1
        b0 = min(mcu_row0 * v, bh)
        out.append((b0, min(n_mcu_rows * v, bh - b0)))
    return out, px_h


def huffman_encode_striped(encode, coefs, w: int, h: int, sampling, restart_interval: int, rank: int, world_size: int):
    """This rank's share of the entropy coding of one image.  ``encode(coef_slices, w, stripe_h, sampling, ri)`` is the
    stripe encoder -- ``UltraHdr.huffman_encode`` on a GPU rank -- and ``coefs`` the whole image's coefficient arrays
    (or any object sliceable along block rows; only this rank's rows are touched).  Returns the rank's stream (bytes-like,
    possibly empty); ``stitch_entropy_streams`` of all ranks' results in rank order is the single-rank stream."""
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    ncomp = len(coefs)
    mcus_per_row = int(coefs[0].shape[1]) if ncomp == 1 else (w + 8 * hmax - 1) // (8 * hmax)
    mcu_rows = int(coefs[0].shape[0]) if ncomp == 1 else (h + 8 * vmax - 1) // (8 * vmax)
    row0, rows = entropy_stripe_plan(mcu_rows, mcus_per_row, restart_interval, world_size)[rank]
    if rows == 0:
        return b""
    slices, px_h = stripe_scan_slices([tuple(c.shape[:2]) for c in coefs], sampling, w, h, row0, rows)
    part = [c[b0: b0 + n] for c, (b0, n) in zip(coefs, slices)]
    return encode(part, w, px_h, sampling, restart_interval)
