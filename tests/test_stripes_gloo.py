"""The N > 1 path on CPU: two gloo ranks shard one image by row stripe and run two-pass gain-map
generation around the product's one exchange step (libultrahdr_amd.stripes.allreduce_minmax +
the C ABI's host-side finalize).  Per-stripe pixel math comes from the CPU oracle here (test
infrastructure -- there is no GPU in this container); the result must equal the whole-image run
bit for bit, metadata included (float min/max is order independent, jpegr.cpp:932-938)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image, stripe_view
from libultrahdr_amd.stripes import allreduce_minmax, finalize_minmax, partition_rows, stripe_granule


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg_kw, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import loader as L

    cfg = A.default_encode_cfg(**cfg_kw)
    s = cfg.map_dimension_scale_factor
    nch = 3 if cfg.use_multi_channel_gainmap else 1
    sdr = synth.make_sdr_yuv420(w, h, noise=0.05)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG, noise=0.05)
    row0, rows = partition_rows(h, world, stripe_granule(s))[rank]
    sv, hv = stripe_view(sdr, row0, rows), stripe_view(hdr, row0, rows)
    mw, mh = w // s, rows // s
    gains = np.zeros(mw * mh * nch, dtype=np.float32)
    mm = (C.c_float * 6)()
    ubc = C.c_int(0)
    assert L.port().uo_generate_gainmap_pass1(C.byref(sv), C.byref(hv), C.byref(cfg), gains.ctypes.data, mm, C.byref(ubc)) == 0
    t = torch.tensor(list(mm), dtype=torch.float32)
    allreduce_minmax(t)                                    # <- the product's exchange step (gloo here, RCCL on GPUs)
    fin, md = finalize_minmax(cfg, hdr.raw.ct, ubc.value, t.tolist())  # <- product host logic via the C ABI
    out = np.zeros((mh, mw * nch), dtype=np.uint8)
    L.port().uo_generate_gainmap_pass2(gains.ctypes.data, (C.c_float * 6)(*fin), cfg.gamma, nch, mw, mh, out.ctypes.data, mw)
    gathered = [None] * world
    dist.all_gather_object(gathered, (row0 // s, out, bytes(md)))
    if rank == 0:
        full = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])], axis=0)
        assert all(g[2] == gathered[0][2] for g in gathered)  # identical metadata on every rank
        np.save(out_path, full)
        open(out_path + ".md", "wb").write(gathered[0][2])
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg_kw", [dict(), dict(map_dimension_scale_factor=4, use_multi_channel_gainmap=0),
                                    dict(max_content_boost=3.0, gamma=1.2)])
def test_two_rank_striped_two_pass_equals_whole_image(tmp_path, cfg_kw):
    from oracle import loader as L

    w, h = 128, 96
    out = str(tmp_path / "gm.npy")
    mp.spawn(_worker, args=(2, _free_port(), cfg_kw, w, h, out), nprocs=2, join=True)
    cfg = A.default_encode_cfg(**cfg_kw)
    md_w, gm_w = L.generate_gainmap("port", synth.make_sdr_yuv420(w, h, noise=0.05),
                                    synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG, noise=0.05), cfg)
    assert np.array_equal(np.load(out), gm_w.valid(0))
    assert open(out + ".md", "rb").read() == bytes(md_w)


def test_allreduce_minmax_single_process_is_identity():
    t = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    assert allreduce_minmax(t.clone()).tolist() == t.tolist()


# ---- entropy stage across ranks ------------------------------------------------------------------------------------
def _entropy_worker(rank, world, port, w, h, sampling, ri, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libultrahdr_amd.stripes import huffman_encode_striped, stitch_entropy_streams
    from oracle import loader as L

    rng = np.random.default_rng(7)  # every rank builds the same image
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    coefs = []
    for hs, vs in sampling:
        cw, ch = -(-w * hs // hmax), -(-h * vs // vmax)
        a = (rng.normal(0, 25, (-(-ch // 8), -(-cw // 8), 64)) * (rng.random((-(-ch // 8), -(-cw // 8), 64)) < 0.25)).astype(np.int16)
        a[..., 0] = rng.integers(-1000, 1000, a.shape[:2])
        coefs.append(np.ascontiguousarray(a))
    # the product's sharding logic; the oracle stands in for the kernel (no GPU here)
    mine = huffman_encode_striped(lambda part, w_, h_, s_, ri_: L.huffman_encode_port(part, w_, h_, s_, ri_), coefs, w, h, sampling, ri, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, bytes(mine))
    if rank == 0:
        whole = L.huffman_encode_port(coefs, w, h, sampling, ri)
        ok = stitch_entropy_streams(gathered) == whole
        with open(out_path, "w") as f:
            f.write("ok" if ok else f"mismatch: {len(stitch_entropy_streams(gathered))} vs {len(whole)} bytes; parts {[len(g) for g in gathered]}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(640, 400, [(2, 2), (1, 1), (1, 1)], 5), (200, 330, [(1, 1)] * 3, 25), (333, 203, [(2, 2), (1, 1), (1, 1)], 3),
                                  (256, 136, [(1, 1)], 16)])
def test_entropy_coding_shards_by_stripe_without_a_collective(tmp_path, case):
    """Two ranks Huffman-code their own MCU rows; the concatenation (RST7 between stripes) is byte-identical to the
    single-rank stream -- odd sizes (dummy blocks in the last stripe) and all three scan layouts."""
    w, h, sampling, ri = case
    out = tmp_path / "result.txt"
    mp.spawn(_entropy_worker, args=(2, _free_port(), w, h, sampling, ri, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def test_entropy_stripe_plan_properties():
    from libultrahdr_amd.stripes import entropy_stripe_plan

    for mcu_rows, mpr, ri, world in ((135, 240, 10, 8), (1024, 1024, 8, 8), (25, 40, 5, 2), (3, 7, 2, 4), (17, 13, 64, 3)):
        plan = entropy_stripe_plan(mcu_rows, mpr, ri, world)
        assert len(plan) == world and sum(n for _, n in plan) == mcu_rows
        row = 0
        nonempty = [p for p in plan if p[1]]
        for r0, n in nonempty:
            assert r0 == row
            row += n
        for r0, n in nonempty[:-1]:  # whole intervals, a multiple of 8 of them
            assert (n * mpr) % (8 * ri) == 0
