"""Host-side mirror of the reference's ``ultrahdr::UltraHdr`` operator interface
(``/root/reference/lib/include/ultrahdr/ultrahdrcommon.h:448-655``): same constructor knobs, same
four methods with the same argument meaning and error behaviour, but every pixel is computed by
the gfx950 kernels behind the C ABI (``include/uhdr_hip.h``).  Images may live on the host
(numpy; the call stages them) or on the GPU (torch CUDA tensors; the call only enqueues kernels).

    u = UltraHdr(device=0, mapDimensionScaleFactor=4, useMultiChannelGainMap=False)
    md, gm = u.generateGainMap(sdr, hdr)
    u.applyGainMap(sdr, gm, md, UHDR_CT_LINEAR, UHDR_IMG_FMT_64bppRGBAHalfFloat, float("inf")...)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi as A
from .images import Image, _align


class Context:
    """One ``uhdr_hip_ctx_t``: a device + a stream.  Raises if the HIP library or a GPU is missing.

    Stream contract of the Python binding.  The library enqueues on the context's own non-blocking stream, torch on
    its current stream.  With ``stream_safe=True`` (the default) every device-buffer method of ``UltraHdr`` is ordered
    against torch by events, never by host synchronisation: before the call the library's stream waits for everything
    queued on torch's current stream (inputs still being produced by torch kernels / copies), after the call torch's
    current stream waits for the library's work (a returned tensor can be used at once: ``u.idct_dequant(c).cpu()``).
    ``stream_safe=False`` leaves ordering to the caller (``ctx.synchronize()`` / ``torch.cuda.synchronize()``), e.g.
    for timing loops that must not record extra events."""

    def __init__(self, device: int = -1, stream=None, stream_safe: bool = True):
        self.lib = A.load()
        err = A.ErrorInfo()
        self.handle = self.lib.uhdr_hip_create(device, C.byref(err))
        if not self.handle:
            raise A.UhdrError(err.error_code, err.detail.decode("utf-8", "replace"))
        self.stream_safe = stream_safe
        self._ext = None
        self._ext_ptr = None
        self._device = device
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        """``stream``: a raw hipStream_t integer or a ``torch.cuda.Stream``."""
        ptr = getattr(stream, "cuda_stream", stream)
        A.check(self.lib.uhdr_hip_set_stream(self.handle, C.c_void_p(ptr)))

    def stream_ptr(self) -> int:
        return int(self.lib.uhdr_hip_get_stream(self.handle) or 0)

    def ordered(self):
        """Context manager that orders the library's stream after torch's current stream on entry and torch's current
        stream after the library's on exit (events only; a no-op when both are the same stream or stream_safe is off)."""
        return _Ordered(self)

    def _streams(self):
        import torch

        ptr = self.stream_ptr()
        if self._ext is None or self._ext_ptr != ptr:
            dev = self._device if self._device >= 0 else torch.cuda.current_device()
            self._ext = torch.cuda.ExternalStream(ptr, device=dev)
            self._ext_ptr = ptr
        return torch.cuda.current_stream(self._ext.device), self._ext

    def synchronize(self):
        A.check(self.lib.uhdr_hip_synchronize(self.handle))

    def profile(self, enable: bool):
        self.lib.uhdr_hip_profile_enable(self.handle, 1 if enable else 0)

    def profile_read(self, family: str | None, reset=True):
        ms = C.c_double(0.0)
        n = self.lib.uhdr_hip_profile_read(self.handle, family.encode() if family else None, C.byref(ms), 1 if reset else 0)
        return n, ms.value

    def profile_read_list(self, family: str | None, reset=True, capacity=65536):
        """Durations (ms) of the recorded launches of a family, in launch order."""
        buf = (C.c_double * capacity)()
        n = self.lib.uhdr_hip_profile_read_list(self.handle, family.encode() if family else None, buf, capacity, 1 if reset else 0)
        return list(buf[: min(n, capacity)])

    def close(self):
        if getattr(self, "handle", None):
            self.lib.uhdr_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Ordered:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.pair = None

    def __enter__(self):
        if self.ctx.stream_safe:
            cur, ext = self.ctx._streams()
            if cur.cuda_stream != ext.cuda_stream:
                ext.wait_stream(cur)
                self.pair = (cur, ext)
        return self

    def __exit__(self, *exc):
        if self.pair is not None:
            self.pair[0].wait_stream(self.pair[1])
        return False


def _is_dev(*imgs) -> bool:
    dev = [im.device is not None for im in imgs]
    if any(dev) and not all(dev):
        raise ValueError("mixing host and device images in one call")
    return all(dev)


class UltraHdr:
    """Mirror of ``ultrahdr::UltraHdr`` (constructor args: ultrahdrcommon.h:450-457)."""

    def __init__(self, ctx: Context | None = None, device: int = -1,
                 mapDimensionScaleFactor: int = 4, mapCompressQuality: int = 85,
                 useMultiChannelGainMap: bool = False, gamma: float = 1.0,
                 preset: int = A.UHDR_USAGE_REALTIME, minContentBoost: float = A.FLT_MIN,
                 maxContentBoost: float = A.FLT_MAX, targetDispPeakBrightness: float = -1.0):
        self.ctx = ctx or Context(device)
        self.lib = self.ctx.lib
        self.mMapDimensionScaleFactor = mapDimensionScaleFactor
        self.mMapCompressQuality = mapCompressQuality
        self.mUseMultiChannelGainMap = useMultiChannelGainMap
        self.mGamma = gamma
        self.mEncPreset = preset
        self.mMinContentBoost = minContentBoost
        self.mMaxContentBoost = maxContentBoost
        self.mTargetDispPeakBrightness = targetDispPeakBrightness

    def _call(self, dev: bool, fn, *args):
        """One C-ABI call; device-buffer calls are ordered against torch's current stream (Context.ordered)."""
        if dev:
            with self.ctx.ordered():
                A.check(fn(*args))
        else:
            A.check(fn(*args))

    def encode_cfg(self, sdr_is_601=False, use_luminance=True) -> A.EncodeCfg:
        return A.EncodeCfg(self.mMapDimensionScaleFactor, int(self.mUseMultiChannelGainMap), self.mGamma,
                           self.mEncPreset, self.mMinContentBoost, self.mMaxContentBoost,
                           self.mTargetDispPeakBrightness, int(sdr_is_601), int(use_luminance))

    # ---- toneMap (ultrahdrcommon.h:482) ----------------------------------------------------
    def toneMap(self, hdr_intent: Image, sdr_intent: Image):
        dev = _is_dev(hdr_intent, sdr_intent)
        fn = self.lib.uhdr_hip_tone_map_dev if dev else self.lib.uhdr_hip_tone_map
        self._call(dev, fn, self.ctx.handle, C.byref(hdr_intent.raw), C.byref(sdr_intent.raw))

    # ---- generateGainMap (ultrahdrcommon.h:507-510) -------------------------------------------
    def gainmap_dims(self, w: int, h: int):
        s = self.mMapDimensionScaleFactor
        mw, mh = w // s, h // s
        if mw == 0 or mh == 0:  # jpegr.cpp:696-706
            s = min(w, h)
            s = s // 8 if s >= 8 else 1
            mw, mh = w // s, h // s
        return mw, mh

    def generateGainMap(self, sdr_intent: Image, hdr_intent: Image, sdr_is_601=False, use_luminance=True):
        """Returns (metadata, gainmap Image) -- the reference fills a fresh 64-aligned image."""
        dev = _is_dev(sdr_intent, hdr_intent)
        mw, mh = self.gainmap_dims(sdr_intent.w, sdr_intent.h)
        fmt = A.UHDR_IMG_FMT_24bppRGB888 if self.mUseMultiChannelGainMap else A.UHDR_IMG_FMT_8bppYCbCr400
        gm = Image(fmt, mw, mh, align=64, device=sdr_intent.device)
        md = A.GainmapMetadata()
        cfg = self.encode_cfg(sdr_is_601, use_luminance)
        fn = self.lib.uhdr_hip_generate_gainmap_dev if dev else self.lib.uhdr_hip_generate_gainmap
        self._call(dev, fn, self.ctx.handle, C.byref(sdr_intent.raw), C.byref(hdr_intent.raw), C.byref(cfg), C.byref(md), C.byref(gm.raw))
        gm.sync_meta_from_raw()
        return md, gm

    def encodeApi0Fused(self, hdr_intent: Image, want_sdr_rgba=True, use_luminance=False):
        """MI355X extension: toneMap + generateGainMap + convert_raw_input_to_ycbcr(4:4:4) of an API-0 encode
        (jpegr.cpp:202-251) in one pass over a device-resident RGBA1010102 / RGBA-F16 image (scale factor 1).
        Returns (sdr_rgba or None, base_ycc444, metadata, gainmap), bit-identical to the three separate calls."""
        assert _is_dev(hdr_intent)
        w, h, dev = hdr_intent.w, hdr_intent.h, hdr_intent.device
        sdr = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w, h, align=64, device=dev) if want_sdr_rgba else None
        ycc = Image(A.UHDR_IMG_FMT_24bppYCbCr444, w, h, align=64, device=dev)
        fmt = A.UHDR_IMG_FMT_24bppRGB888 if self.mUseMultiChannelGainMap else A.UHDR_IMG_FMT_8bppYCbCr400
        gm = Image(fmt, w, h, align=64, device=dev)
        md = A.GainmapMetadata()
        cfg = self.encode_cfg(False, use_luminance)
        self._call(True, self.lib.uhdr_hip_encode_api0_fused_dev, self.ctx.handle, C.byref(hdr_intent.raw), C.byref(cfg),
                   C.byref(sdr.raw) if sdr is not None else None, C.byref(ycc.raw), C.byref(md), C.byref(gm.raw))
        gm.sync_meta_from_raw()
        return sdr, ycc, md, gm

    def encodeApi1Fused(self, sdr_intent: Image, hdr_intent: Image, base_encoding: int, qt_base, qt_map, want_map=True,
                        sdr_is_601=False, use_luminance=True):
        """MI355X extension: the sample -> coefficient part of an API-1 encode (jpegr.cpp:253-316) in four launches on
        device-resident images -- two-pass generateGainMap fused with the map's rgb->ycc + FDCT, convertYuv fused with the base
        image's three FDCTs.  qt_base / qt_map: (luma, chroma) quantization tables.  Returns (base coefficient tensors [3],
        map coefficient tensors [1 or 3], metadata, gainmap Image or None), bit-identical to the separate operators."""
        import torch

        assert _is_dev(sdr_intent, hdr_intent)
        w, h, dev = sdr_intent.w, sdr_intent.h, sdr_intent.buf.device
        mw, mh = self.gainmap_dims(w, h)
        nch = 3 if self.mUseMultiChannelGainMap else 1
        base = [torch.empty((h // 8, w // 8, 64), dtype=torch.int16, device=dev)] + [torch.empty((h // 16, w // 16, 64), dtype=torch.int16, device=dev) for _ in range(2)]
        mapc = [torch.empty((mh // 8, mw // 8, 64), dtype=torch.int16, device=dev) for _ in range(nch)]
        blocks = A.Api1Blocks()
        for i in range(3):
            blocks.base_coef[i] = base[i].data_ptr()
            blocks.map_coef[i] = mapc[i].data_ptr() if i < nch else None
        qb = np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.uint16) for q in qt_base]))
        qm = np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.uint16) for q in qt_map]))
        gm = None
        if want_map:
            gm = Image(A.UHDR_IMG_FMT_24bppRGB888 if nch == 3 else A.UHDR_IMG_FMT_8bppYCbCr400, mw, mh, align=64, device=sdr_intent.device)
        md = A.GainmapMetadata()
        cfg = self.encode_cfg(sdr_is_601, use_luminance)
        self._call(True, self.lib.uhdr_hip_encode_api1_fused_dev, self.ctx.handle, C.byref(sdr_intent.raw), C.byref(hdr_intent.raw), C.byref(cfg),
                   base_encoding, C.c_void_p(qb.ctypes.data), C.c_void_p(qm.ctypes.data), C.byref(blocks), C.byref(md),
                   C.byref(gm.raw) if gm is not None else None)
        if gm is not None:
            gm.sync_meta_from_raw()
        return base, mapc, md, gm

    # ---- one entry point per direction of the API-1 round trip, device resident (round 6) -----------------
    def encodeApi1Scans(self, sdr_intent: Image, hdr_intent: Image, base_encoding: int, qt_base, qt_map, out_base, out_map,
                        sdr_is_601=False, use_luminance=True):
        """JpegR::encodeJPEGR API-1 (jpegr.cpp:253-316) without the container, on device images: the fused sample -> coefficient
        chain and both scans' Huffman coding in ONE C call (uhdr_hip_encode_api1_scans_dev).  out_base / out_map: uint8 CUDA tensors
        that receive the entropy-coded scans.  Returns (bytes of the base scan, bytes of the map scan, metadata)."""
        assert _is_dev(sdr_intent, hdr_intent) and out_base.is_cuda and out_map.is_cuda
        qb, qm = self._qt_pair(qt_base), self._qt_pair(qt_map)
        md = A.GainmapMetadata()
        cfg = self.encode_cfg(sdr_is_601, use_luminance)
        nb, nm = C.c_size_t(0), C.c_size_t(0)
        self._call(True, self.lib.uhdr_hip_encode_api1_scans_dev, self.ctx.handle, C.byref(sdr_intent.raw), C.byref(hdr_intent.raw), C.byref(cfg), base_encoding,
                   C.c_void_p(qb.ctypes.data), C.c_void_p(qm.ctypes.data), C.byref(md), None, C.c_void_p(out_base.data_ptr()), int(out_base.numel()), C.byref(nb),
                   C.c_void_p(out_map.data_ptr()), int(out_map.numel()), C.byref(nm))
        return int(nb.value), int(nm.value), md

    def bindEncodeApi1Scans(self, sdr_intent: Image, hdr_intent: Image, base_encoding: int, qt_base, qt_map, out_base, out_map,
                            sdr_is_601=False, use_luminance=True):
        """encodeApi1Scans with every argument marshalled ONCE: returns run() -> (bytes of the base scan, bytes of the map scan, metadata) for a
        caller that codes frame after frame from and into the same device buffers (a service's steady state; ~5 us of ctypes work per call
        otherwise).  The images, tables and outputs must stay alive and in place; stream ordering against torch as in every device call."""
        assert _is_dev(sdr_intent, hdr_intent) and out_base.is_cuda and out_map.is_cuda
        qb, qm = self._qt_pair(qt_base), self._qt_pair(qt_map)
        md = A.GainmapMetadata()
        cfg = self.encode_cfg(sdr_is_601, use_luminance)
        nb, nm = C.c_size_t(0), C.c_size_t(0)
        fn = self.lib.uhdr_hip_encode_api1_scans_dev
        args = (self.ctx.handle, C.byref(sdr_intent.raw), C.byref(hdr_intent.raw), C.byref(cfg), base_encoding, C.c_void_p(qb.ctypes.data),
                C.c_void_p(qm.ctypes.data), C.byref(md), None, C.c_void_p(out_base.data_ptr()), int(out_base.numel()), C.byref(nb),
                C.c_void_p(out_map.data_ptr()), int(out_map.numel()), C.byref(nm))
        keep = (sdr_intent, hdr_intent, qb, qm, cfg, out_base, out_map)
        ordered, check = self.ctx.ordered, A.check

        def run(_keep=keep):
            with ordered():
                check(fn(*args))
            return nb.value, nm.value, md

        return run

    def bindDecodeApi1Scans(self, base_hdr: "A.JpegHeader", base_data, base_cg: int, map_hdr: "A.JpegHeader", map_data, map_cg: int,
                            gainmap_metadata: A.GainmapMetadata, output_ct: int, output_format: int, max_display_boost: float, dest: Image,
                            libjpeg_variant: int = 0):
        """decodeApi1Scans with the arguments marshalled once (see bindEncodeApi1Scans): returns run()."""
        assert base_data.is_cuda and map_data.is_cuda and _is_dev(dest)
        fn = self.lib.uhdr_hip_decode_api1_scans_dev
        args = (self.ctx.handle, C.byref(base_hdr), C.c_void_p(base_data.data_ptr()), int(base_data.numel()), base_cg, C.byref(map_hdr),
                C.c_void_p(map_data.data_ptr()), int(map_data.numel()), map_cg, libjpeg_variant, C.byref(gainmap_metadata), output_ct, output_format,
                max_display_boost, C.byref(dest.raw))
        keep = (base_hdr, base_data, map_hdr, map_data, gainmap_metadata, dest)
        ordered, check = self.ctx.ordered, A.check

        def run(_keep=keep):
            with ordered():
                check(fn(*args))

        return run

    def _qt_pair(self, qt):
        """(luma, chroma) quantization tables -> one contiguous uint16 [2][64] block; built once per pair of arrays (a per-frame caller hands in
        the same two arrays every time: no numpy work between the frames)."""
        key = (id(qt[0]), id(qt[1]))
        cache = self.__dict__.setdefault("_qt_cache", {})
        hit = cache.get(key)
        if hit is not None and hit[0] is qt[0] and hit[1] is qt[1]:
            return hit[2]
        blk = np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.uint16) for q in qt]))
        if len(cache) > 64:
            cache.clear()
        cache[key] = (qt[0], qt[1], blk)
        return blk

    @staticmethod
    def jpeg_header(w: int, h: int, sampling, qtables, restart_interval: int = 0) -> "A.JpegHeader":
        """What uhdr_hip_jpeg_parse fills for a baseline file with the Annex K Huffman tables: scan geometry of a w x h image with the
        given sampling factors [(h, v)] per component, and its quantization tables (one per component)."""
        hd = A.JpegHeader()
        sc = hd.scan
        sc.num_components = len(sampling)
        hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
        for i, (hs, vs) in enumerate(sampling):
            sc.h_samp[i], sc.v_samp[i] = hs, vs
            pw, ph = (w * hs + hmax - 1) // hmax, (h * vs + vmax - 1) // vmax
            sc.blocks_w[i], sc.blocks_h[i] = (pw + 7) // 8, (ph + 7) // 8
            for k in range(64):
                hd.qtable[i][k] = int(qtables[i][k])
        sc.w, sc.h, sc.restart_interval = w, h, restart_interval
        return hd  # (tables left zero: the Annex K tables)

    def decodeApi1Scans(self, base_hdr: "A.JpegHeader", base_data, base_cg: int, map_hdr: "A.JpegHeader", map_data, map_cg: int,
                        gainmap_metadata: A.GainmapMetadata, output_ct: int, output_format: int, max_display_boost: float, dest: Image,
                        libjpeg_variant: int = 0):
        """JpegR::decodeJPEGR behind its container parsing (jpegr.cpp:1469-1531) on device data in ONE C call
        (uhdr_hip_decode_api1_scans_dev): both scans entropy-decoded, the map's IDCT, applyGainMap with the base image's IDCT inside."""
        assert base_data.is_cuda and map_data.is_cuda and _is_dev(dest)
        self._call(True, self.lib.uhdr_hip_decode_api1_scans_dev, self.ctx.handle, C.byref(base_hdr), C.c_void_p(base_data.data_ptr()), int(base_data.numel()), base_cg,
                   C.byref(map_hdr), C.c_void_p(map_data.data_ptr()), int(map_data.numel()), map_cg, libjpeg_variant, C.byref(gainmap_metadata), output_ct, output_format,
                   max_display_boost, C.byref(dest.raw))

    # ---- applyGainMap (ultrahdrcommon.h:531-534) -----------------------------------------------
    def applyGainMap(self, sdr_intent: Image, gainmap_img: Image, gainmap_metadata: A.GainmapMetadata,
                     output_ct: int, output_format: int, max_display_boost: float, dest: Image,
                     y0: int = 0, full_height: int = 0):
        if _is_dev(sdr_intent, gainmap_img, dest):
            self._call(True, self.lib.uhdr_hip_apply_gainmap_dev,
                       self.ctx.handle, C.byref(sdr_intent.raw), C.byref(gainmap_img.raw), C.byref(gainmap_metadata),
                       output_ct, output_format, max_display_boost, C.byref(dest.raw), y0, full_height)
        else:
            if y0 or full_height:
                raise ValueError("stripes are a device-buffer feature")
            A.check(self.lib.uhdr_hip_apply_gainmap(
                self.ctx.handle, C.byref(sdr_intent.raw), C.byref(gainmap_img.raw), C.byref(gainmap_metadata),
                output_ct, output_format, max_display_boost, C.byref(dest.raw)))

    def applyGainMapFromCoefficients(self, coefs, qtables, w: int, h: int, base_cg: int, gainmap_img: Image,
                                     gainmap_metadata: A.GainmapMetadata, output_ct: int, output_format: int,
                                     max_display_boost: float, dest: Image):
        """applyGainMap on a 4:2:0 base image still in coefficient form (what jpeg_read_coefficients() yields): coefs =
        three int16 [blocks_h, blocks_w, 64] CUDA tensors (Y, Cb, Cr), qtables = their three quantization tables; the
        dequantize + IDCT stage runs inside the kernel.  == idct_dequant x 3 + applyGainMap, bit for bit."""
        assert all(c.is_cuda for c in coefs) and _is_dev(gainmap_img, dest)
        jc = A.JpegCoefficients()
        for i in range(3):
            jc.coef[i] = coefs[i].data_ptr()
            jc.blocks_h[i], jc.blocks_w[i] = int(coefs[i].shape[0]), int(coefs[i].shape[1])
            for k in range(64):
                jc.qtable[i][k] = int(qtables[i][k])
        self._call(True, self.lib.uhdr_hip_apply_gainmap_coef_dev, self.ctx.handle, C.byref(jc), w, h, base_cg, C.byref(gainmap_img.raw),
                   C.byref(gainmap_metadata), output_ct, output_format, max_display_boost, C.byref(dest.raw))

    def applyGainMapBatch(self, sdr_intents, gainmap_imgs, gainmap_metadata: A.GainmapMetadata, output_ct: int,
                          output_format: int, max_display_boost: float, dests):
        """n device frames of identical geometry sharing one metadata block -> one kernel launch."""
        n = len(sdr_intents)
        assert n == len(gainmap_imgs) == len(dests) and n > 0
        arr = lambda imgs: (A.RawImage * n)(*[im.raw for im in imgs])
        s, g, d = arr(sdr_intents), arr(gainmap_imgs), arr(dests)
        self._call(True, self.lib.uhdr_hip_apply_gainmap_batch_dev, self.ctx.handle, n, s, g, C.byref(gainmap_metadata), output_ct,
                   output_format, max_display_boost, d)
        for im, r in zip(dests, d):
            im.raw.cg = r.cg

    # ---- convertYuv (ultrahdrcommon.h:545-546) -------------------------------------------------
    def convertYuv(self, image: Image, src_encoding: int, dst_encoding: int):
        dev = _is_dev(image)
        fn = self.lib.uhdr_hip_convert_yuv_dev if dev else self.lib.uhdr_hip_convert_yuv
        self._call(dev, fn, self.ctx.handle, C.byref(image.raw), src_encoding, dst_encoding)

    # ---- convert_raw_input_to_ycbcr (gainmapmath.h:604-605) -----------------------------------
    def convert_raw_input_to_ycbcr(self, src: Image, chroma_sampling_enabled=False) -> Image:
        ten = src.fmt == A.UHDR_IMG_FMT_32bppRGBA1010102
        if ten:
            fmt = A.UHDR_IMG_FMT_24bppYCbCrP010 if chroma_sampling_enabled else A.UHDR_IMG_FMT_30bppYCbCr444
        else:
            fmt = A.UHDR_IMG_FMT_12bppYCbCr420 if chroma_sampling_enabled else A.UHDR_IMG_FMT_24bppYCbCr444
        dst = Image(fmt, src.w, src.h, align=64, device=src.device)
        dev = _is_dev(src)
        fn = self.lib.uhdr_hip_convert_raw_input_to_ycbcr_dev if dev else self.lib.uhdr_hip_convert_raw_input_to_ycbcr
        self._call(dev, fn, self.ctx.handle, C.byref(src.raw), int(chroma_sampling_enabled), C.byref(dst.raw))
        return dst

    def copy_raw_image(self, src: Image, dst: Image) -> Image:
        """copy_raw_image(src, dst) between device images (equal formats, RGB888 -> RGBA8888, RGBA8888 -> Y400)."""
        self._call(True, self.lib.uhdr_hip_copy_raw_image_dev, self.ctx.handle, C.byref(src.raw), C.byref(dst.raw))
        return dst

    # ---- JPEG stage -----------------------------------------------------------------------------
    def quant_table(self, quality: int, is_chroma: bool) -> np.ndarray:
        qt = (C.c_uint16 * 64)()
        self.lib.uhdr_hip_jpeg_quant_table(quality, int(is_chroma), qt)
        return np.frombuffer(qt, dtype=np.uint16).copy()

    def fdct_quant(self, plane, stride: int, blocks_w: int, blocks_h: int, qtable: np.ndarray, coef=None):
        """plane: numpy uint8 array (host) or torch uint8 CUDA tensor (device), covering
        blocks_h*8 rows of ``stride`` bytes.  Returns int16 [blocks_h, blocks_w, 64]."""
        qt = (C.c_uint16 * 64)(*[int(v) for v in qtable])
        if isinstance(plane, np.ndarray):
            out = np.zeros((blocks_h, blocks_w, 64), dtype=np.int16) if coef is None else coef
            A.check(self.lib.uhdr_hip_fdct_quant(self.ctx.handle, C.c_void_p(plane.ctypes.data), stride,
                                                 blocks_w, blocks_h, qt, C.c_void_p(out.ctypes.data)))
            return out
        import torch

        out = torch.empty((blocks_h, blocks_w, 64), dtype=torch.int16, device=plane.device) if coef is None else coef
        self._call(True, self.lib.uhdr_hip_fdct_quant_dev, self.ctx.handle, C.c_void_p(plane.data_ptr()), stride, blocks_w, blocks_h, qt,
                   C.c_void_p(out.data_ptr()))
        return out

    def fdct_quant_rgb(self, rgb: Image, qt_luma: np.ndarray, qt_chroma: np.ndarray):
        """3-channel gain map (device RGB888 / RGBA8888 image, w and h multiples of 8): libjpeg's RGB -> YCbCr and the
        FDCT + quantize of all three components in one pass.  Returns three int16 [h/8, w/8, 64] CUDA tensors."""
        import torch

        assert _is_dev(rgb)
        ql = (C.c_uint16 * 64)(*[int(v) for v in qt_luma])
        qc = (C.c_uint16 * 64)(*[int(v) for v in qt_chroma])
        outs = [torch.empty((rgb.h // 8, rgb.w // 8, 64), dtype=torch.int16, device=rgb.buf.device) for _ in range(3)]
        self._call(True, self.lib.uhdr_hip_fdct_quant_rgb_dev, self.ctx.handle, C.byref(rgb.raw), ql, qc, *[C.c_void_p(o.data_ptr()) for o in outs])
        return outs

    # ---- entropy stage ------------------------------------------------------------------
    @staticmethod
    def _scan(coefs, w, h, sampling, restart_interval) -> "A.JpegScan":
        sc = A.JpegScan()
        sc.num_components = len(coefs)
        for i, cf in enumerate(coefs):
            sc.coef[i] = cf.data_ptr() if hasattr(cf, "data_ptr") else None
            sc.blocks_h[i], sc.blocks_w[i] = int(cf.shape[0]), int(cf.shape[1])
            sc.h_samp[i], sc.v_samp[i] = sampling[i]
        sc.w, sc.h, sc.restart_interval = w, h, restart_interval
        return sc

    def huffman_encode(self, coefs, w: int, h: int, sampling, restart_interval: int, out=None):
        """Baseline Huffman coding (Annex K tables) of quantized coefficients: coefs = int16 [blocks_h, blocks_w, 64] CUDA
        tensors per component, sampling = [(h, v)] per component, restart_interval in MCUs (one wavefront per interval).
        Returns a uint8 CUDA tensor holding the entropy-coded data (between the SOS header and EOI, RSTn markers included)."""
        import torch

        assert all(c.is_cuda for c in coefs)
        sc = self._scan(coefs, w, h, sampling, restart_interval)
        if out is None:
            # worst case: 1660 bits per block, every byte stuffed
            out = torch.empty(sum(int(c.numel()) for c in coefs) // 64 * 416 + 4096, dtype=torch.uint8, device=coefs[0].device)
        n = C.c_size_t(0)
        self._call(True, self.lib.uhdr_hip_huffman_encode_dev, self.ctx.handle, C.byref(sc), C.c_void_p(out.data_ptr()), out.numel(), C.byref(n))
        return out[: n.value]

    def huffman_decode(self, data, shapes, w: int, h: int, sampling, restart_interval: int, tables=None):
        """Entropy-coded data (uint8 CUDA tensor: the bytes between the SOS header and EOI) -> int16 [blocks_h, blocks_w, 64]
        CUDA tensors per component (shapes = [(blocks_h, blocks_w)]).  tables: (bits[4][17], vals[4][256]) from the file's
        DHT segments, None = Annex K.  One lane per restart interval."""
        import torch

        assert data.is_cuda and data.dtype == torch.uint8
        coefs = [torch.empty((bh, bw, 64), dtype=torch.int16, device=data.device) for (bh, bw) in shapes]
        sc = self._scan(coefs, w, h, sampling, restart_interval)
        ht = None
        if tables is not None:
            ht = A.HuffTables()
            for t in range(4):
                for i in range(17):
                    ht.bits[t][i] = int(tables[0][t][i])
                for i in range(256):
                    ht.vals[t][i] = int(tables[1][t][i])
        self._call(True, self.lib.uhdr_hip_huffman_decode_dev, self.ctx.handle, C.byref(sc), C.byref(ht) if ht is not None else None,
                   C.c_void_p(data.data_ptr()), data.numel())
        return coefs

    def huffman_encode2(self, coefs_a, w_a, h_a, sampling_a, coefs_b, w_b, h_b, sampling_b, restart_interval: int = 0, outs=None):
        """The two scans of one UltraHDR file (base image, gain map) coded concurrently (uhdr_hip_huffman_encode2_dev): two
        huffman_encode calls' arguments, returns the two uint8 CUDA tensors."""
        import torch

        sa, sb = self._scan(coefs_a, w_a, h_a, sampling_a, restart_interval), self._scan(coefs_b, w_b, h_b, sampling_b, restart_interval)
        if outs is None:
            outs = [torch.empty(sum(int(c.numel()) for c in cf) // 64 * 416 + 4096, dtype=torch.uint8, device=cf[0].device) for cf in (coefs_a, coefs_b)]
        na, nb = C.c_size_t(0), C.c_size_t(0)
        self._call(True, self.lib.uhdr_hip_huffman_encode2_dev, self.ctx.handle, C.byref(sa), C.c_void_p(outs[0].data_ptr()), outs[0].numel(), C.byref(na),
                   C.byref(sb), C.c_void_p(outs[1].data_ptr()), outs[1].numel(), C.byref(nb))
        return outs[0][: na.value], outs[1][: nb.value]

    def huffman_decode2(self, data_a, shapes_a, w_a, h_a, sampling_a, data_b, shapes_b, w_b, h_b, sampling_b, restart_interval: int = 0):
        """The two scans of one UltraHDR file decoded concurrently (uhdr_hip_huffman_decode2_dev), Annex K tables: returns the two
        lists of int16 [blocks_h, blocks_w, 64] CUDA tensors."""
        import torch

        ca = [torch.empty((bh, bw, 64), dtype=torch.int16, device=data_a.device) for (bh, bw) in shapes_a]
        cb = [torch.empty((bh, bw, 64), dtype=torch.int16, device=data_b.device) for (bh, bw) in shapes_b]
        sa, sb = self._scan(ca, w_a, h_a, sampling_a, restart_interval), self._scan(cb, w_b, h_b, sampling_b, restart_interval)
        self._call(True, self.lib.uhdr_hip_huffman_decode2_dev, self.ctx.handle, C.byref(sa), None, C.c_void_p(data_a.data_ptr()), data_a.numel(),
                   C.byref(sb), None, C.c_void_p(data_b.data_ptr()), data_b.numel())
        return ca, cb

    def jpeg_parse(self, jpeg: bytes) -> "A.JpegHeader":
        """Host helper: the headers of a baseline JPEG file in the form the device decode path takes (ValueError with the
        library's code for files outside that path, e.g. -8 for progressive)."""
        hdr = A.JpegHeader()
        buf = np.frombuffer(jpeg, dtype=np.uint8)
        rc = self.lib.uhdr_hip_jpeg_parse(C.c_void_p(buf.ctypes.data), buf.size, C.byref(hdr))
        if rc != 0:
            raise ValueError(f"uhdr_hip_jpeg_parse: {rc}")
        return hdr

    def jpeg_to_coefficients(self, jpeg: bytes, device="cuda:0"):
        """JPEG file bytes -> (header, [int16 [blocks_h, blocks_w, 64] CUDA tensors]): headers parsed on the host, the
        entropy-coded data decoded on the device (one lane per restart interval)."""
        import torch

        hdr = self.jpeg_parse(jpeg)
        sc = hdr.scan
        nc = sc.num_components
        data = torch.from_numpy(np.frombuffer(jpeg, dtype=np.uint8)[hdr.scan_offset: hdr.scan_offset + hdr.scan_bytes].copy()).to(device)
        bits = np.frombuffer(hdr.tables.bits, dtype=np.uint8).reshape(4, 17)
        vals = np.frombuffer(hdr.tables.vals, dtype=np.uint8).reshape(4, 256)
        coefs = self.huffman_decode(data, [(sc.blocks_h[c], sc.blocks_w[c]) for c in range(nc)], sc.w, sc.h,
                                    [(sc.h_samp[c], sc.v_samp[c]) for c in range(nc)], sc.restart_interval, tables=(bits, vals))
        return hdr, coefs

    def jpeg_decode(self, jpeg: bytes, rgb_channels: int = 0, libjpeg_variant: int = 0, outs=None):
        """JpegDecoderHelper::decompressImage for a baseline JPEG file, entirely on the device (entropy decode, dequantization,
        IDCT, and ycc -> rgb when rgb_channels is 3 / 4): host bytes in, numpy arrays out.  rgb_channels 0: the list of
        component planes [blocks_h*8, blocks_w*8] uint8 (block padding included, as libjpeg's raw-data mode);
        3 / 4: one [h, w, channels] array (4:4:4 files only).  outs: arrays of those shapes to decode into (a list in both
        modes) instead of fresh ones."""
        hdr = self.jpeg_parse(jpeg)
        sc = hdr.scan
        nc = sc.num_components
        buf = np.frombuffer(jpeg, dtype=np.uint8)
        if rgb_channels:
            outs = outs or [np.empty((sc.h, sc.w, rgb_channels), dtype=np.uint8)]
            hs = [sc.w, 0, 0]
            vs = [sc.h, 0, 0]
        else:
            outs = outs or [np.empty((sc.blocks_h[c] * 8, sc.blocks_w[c] * 8), dtype=np.uint8) for c in range(nc)]
            hs = [sc.blocks_w[c] * 8 if c < nc else 0 for c in range(3)]
            vs = [sc.blocks_h[c] * 8 if c < nc else 0 for c in range(3)]
        ptrs = (C.c_void_p * 3)(*[o.ctypes.data for o in outs] + [None] * (3 - len(outs)))
        self._call(False, self.lib.uhdr_hip_jpeg_decode_scan, self.ctx.handle, C.byref(hdr), C.c_void_p(buf.ctypes.data + hdr.scan_offset),
                   buf.size - hdr.scan_offset, rgb_channels, libjpeg_variant, ptrs, (C.c_uint * 3)(*hs), (C.c_uint * 3)(*vs))
        return outs[0] if rgb_channels else outs

    def jpeg_encode(self, planes, w: int, h: int, sampling, qt_luma, qt_chroma, rgb_channels: int = 0) -> bytes:
        """JpegEncoderHelper::compressImage on the device, Huffman pass included (restart intervals of 64 // blocks-per-MCU
        MCUs: one wavefront each): host samples in, a complete baseline JFIF file out.  planes: uint8 numpy arrays
        [blocks_h*8, blocks_w*8] per component (padded to whole blocks), or -- rgb_channels 3 / 4 -- one [h, w, channels]
        array for a 4:4:4 file (libjpeg's rgb_ycc_convert included)."""
        class _Grid:  # what _scan reads from a coefficient array
            def __init__(self, bh, bw):
                self.shape = (bh, bw, 64)

        if rgb_channels:
            img = np.ascontiguousarray(planes, dtype=np.uint8)
            grids = [_Grid(h // 8, w // 8)] * 3
            srcs, strides = [img], [w, 0, 0]
        else:
            srcs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
            grids = [_Grid(p.shape[0] // 8, p.shape[1] // 8) for p in srcs]
            strides = [p.shape[1] for p in srcs] + [0] * (3 - len(srcs))
        bpm = sum(hs * vs for hs, vs in sampling) if len(sampling) > 1 else 1
        ri = 64 // bpm
        sc = self._scan(grids, w, h, sampling, ri)
        qt = np.zeros((3, 64), dtype=np.uint16)
        qt[0], qt[1], qt[2] = qt_luma, qt_chroma, qt_chroma
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in srcs] + [None] * (3 - len(srcs)))
        out = np.zeros(sum(p.size for p in srcs) + (1 << 16), dtype=np.uint8)
        n = C.c_size_t(0)
        self._call(False, self.lib.uhdr_hip_jpeg_encode_scan, self.ctx.handle, C.byref(sc), C.c_void_p(qt.ctypes.data), ptrs,
                   (C.c_uint * 3)(*strides), rgb_channels, C.c_void_p(out.ctypes.data), out.size, C.byref(n))
        return self.jpeg_assemble(grids, w, h, sampling, ri, qt_luma, qt_chroma, out[: n.value].tobytes())

    def jpeg_encode_image(self, planes, w: int, h: int, sampling, qt_luma, qt_chroma, rgb_channels: int = 0, restart_interval: int = 0) -> bytes:
        """JpegEncoderHelper::compressImage's sample -> entropy-coded-data part for the IMAGE's own planes
        (uhdr_hip_jpeg_encode_image): partial edge blocks are padded on the device by the helper's rules
        (jpegencoderhelper.cpp:246-309) / libjpeg's edge replication (packed RGB).  planes: uint8 numpy arrays of at least
        ceil(h * vs / vmax) rows; the row length of each array is its stride (bytes beyond the plane width are the caller's
        stride bytes), or -- rgb_channels 3 / 4 -- one [h, stride_px, channels] array.  Returns the entropy-coded bytes."""
        class _Grid:
            def __init__(self, bh, bw):
                self.shape = (bh, bw, 64)

        hmax, vmax = max(s_[0] for s_ in sampling), max(s_[1] for s_ in sampling)
        if rgb_channels:
            img = np.ascontiguousarray(planes, dtype=np.uint8)
            grids = [_Grid(-(-h // 8), -(-w // 8))] * 3
            srcs, strides = [img], [img.shape[1], 0, 0]
        else:
            srcs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
            grids = []
            for (hs, vs) in (sampling if len(sampling) > 1 else [(1, 1)]):
                pw, ph = (-(-w * hs // hmax), -(-h * vs // vmax)) if len(sampling) > 1 else (w, h)
                grids.append(_Grid(-(-ph // 8), -(-pw // 8)))
            strides = [p.shape[1] for p in srcs] + [0] * (3 - len(srcs))
        sc = self._scan(grids, w, h, sampling, restart_interval)
        qt = np.zeros((3, 64), dtype=np.uint16)
        qt[0], qt[1], qt[2] = qt_luma, qt_chroma, qt_chroma
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in srcs] + [None] * (3 - len(srcs)))
        out = np.zeros(sum(p.size for p in srcs) * 2 + (1 << 16), dtype=np.uint8)
        n = C.c_size_t(0)
        self._call(False, self.lib.uhdr_hip_jpeg_encode_image, self.ctx.handle, C.byref(sc), C.c_void_p(qt.ctypes.data), ptrs,
                   (C.c_uint * 3)(*strides), rgb_channels, C.c_void_p(out.ctypes.data), out.size, C.byref(n))
        return out[: n.value].tobytes()

    def jpeg_assemble(self, coefs, w: int, h: int, sampling, restart_interval: int, qt_luma, qt_chroma, scan_data: bytes) -> bytes:
        """Host helper: a complete baseline JFIF file around entropy-coded data (coefs only supply the block grids)."""
        sc = self._scan(coefs, w, h, sampling, restart_interval)
        ql = (C.c_uint16 * 64)(*[int(v) for v in qt_luma])
        qc = (C.c_uint16 * 64)(*[int(v) for v in qt_chroma])
        src = np.frombuffer(scan_data, dtype=np.uint8)
        out = np.zeros(src.size + 2048, dtype=np.uint8)
        n = self.lib.uhdr_hip_jpeg_assemble(C.byref(sc), ql, qc, C.c_void_p(src.ctypes.data), src.size, C.c_void_p(out.ctypes.data), out.size)
        if n == 0:
            raise ValueError("uhdr_hip_jpeg_assemble rejected the scan description")
        return out[:n].tobytes()

    def idct_dequant(self, coef, qtable: np.ndarray, plane=None, stride: int = 0):
        """Inverse of fdct_quant.  coef: int16 [blocks_h, blocks_w, 64] numpy array (host) or CUDA
        tensor (device).  Returns the uint8 plane [blocks_h*8, stride] (stride defaults to blocks_w*8)."""
        qt = (C.c_uint16 * 64)(*[int(v) for v in qtable])
        blocks_h, blocks_w = int(coef.shape[0]), int(coef.shape[1])
        stride = stride or blocks_w * 8
        if isinstance(coef, np.ndarray):
            coef = np.ascontiguousarray(coef, dtype=np.int16)
            out = np.zeros((blocks_h * 8, stride), dtype=np.uint8) if plane is None else plane
            A.check(self.lib.uhdr_hip_idct_dequant(self.ctx.handle, C.c_void_p(coef.ctypes.data), blocks_w, blocks_h, qt,
                                                   C.c_void_p(out.ctypes.data), stride))
            return out
        import torch

        out = torch.empty((blocks_h * 8, stride), dtype=torch.uint8, device=coef.device) if plane is None else plane
        self._call(True, self.lib.uhdr_hip_idct_dequant_dev, self.ctx.handle, C.c_void_p(coef.data_ptr()), blocks_w, blocks_h, qt,
                   C.c_void_p(out.data_ptr()), stride)
        return out

    def idct_dequant_rgb(self, coefs, qt_luma: np.ndarray, qt_chroma: np.ndarray, w: int, h: int,
                         fmt=A.UHDR_IMG_FMT_32bppRGBA8888, libjpeg_variant: int = 0, dst: Image = None) -> Image:
        """Decoded 3-channel gain map straight from its coefficients: dequant + IDCT of Y, Cb, Cr (three int16
        [ceil(h/8), ceil(w/8), 64] CUDA tensors) and libjpeg's YCbCr -> RGB in one pass.  Returns the device image."""
        assert all(c.is_cuda for c in coefs)
        blocks_h, blocks_w = int(coefs[0].shape[0]), int(coefs[0].shape[1])
        ql = (C.c_uint16 * 64)(*[int(v) for v in qt_luma])
        qc = (C.c_uint16 * 64)(*[int(v) for v in qt_chroma])
        if dst is None:
            dst = Image(fmt, w, h, align=64, device=str(coefs[0].device))
        self._call(True, self.lib.uhdr_hip_idct_dequant_rgb_dev, self.ctx.handle, *[C.c_void_p(c.data_ptr()) for c in coefs], blocks_w, blocks_h,
                   ql, qc, libjpeg_variant, C.byref(dst.raw))
        return dst

    def jpeg_rgb_to_ycc(self, rgb: Image) -> Image:
        """libjpeg's JCS_RGB -> YCbCr (what happens to a 3-channel gain map inside jpeg_write_scanlines):
        RGB888 / RGBA8888 -> YCbCr 4:4:4 planes, ready for fdct_quant."""
        dst = Image(A.UHDR_IMG_FMT_24bppYCbCr444, rgb.w, rgb.h, align=64, device=rgb.device)
        dev = _is_dev(rgb)
        fn = self.lib.uhdr_hip_jpeg_rgb_to_ycc_dev if dev else self.lib.uhdr_hip_jpeg_rgb_to_ycc
        self._call(dev, fn, self.ctx.handle, C.byref(rgb.raw), C.byref(dst.raw))
        return dst

    def jpeg_ycc_to_rgb(self, ycc: Image, fmt=A.UHDR_IMG_FMT_24bppRGB888, libjpeg_variant: int = 0) -> Image:
        """libjpeg's YCbCr -> RGB of a decoded 3-channel gain map (variant 0: 6b / turbo, 1: IJG 9)."""
        dst = Image(fmt, ycc.w, ycc.h, align=64, device=ycc.device)
        dev = _is_dev(ycc)
        fn = self.lib.uhdr_hip_jpeg_ycc_to_rgb_dev if dev else self.lib.uhdr_hip_jpeg_ycc_to_rgb
        self._call(dev, fn, self.ctx.handle, C.byref(ycc.raw), libjpeg_variant, C.byref(dst.raw))
        return dst
