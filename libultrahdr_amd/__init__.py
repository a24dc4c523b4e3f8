"""libultrahdr_amd -- MI355X (gfx950) implementation of libultrahdr's gain-map hot path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/uhdr_hip.h)  -> lib/libuhdr_hip.so
  capi.py      ctypes binding of that ABI (what a cgo/JNI/C++ caller would bind)
  ultrahdr.py  host mirror of the reference's ultrahdr::UltraHdr operator interface
  images.py    uhdr_raw_image_ext-style containers (numpy host / torch device memory)
  synth.py     deterministic synthetic inputs
  stripes.py   row-stripe sharding across ranks + the one RCCL exchange (two-pass min/max)
"""
from . import capi  # noqa: F401
from .capi import *  # noqa: F401,F403  (enum constants)
from .images import Image, stripe_view  # noqa: F401

__all__ = ["capi", "Image", "stripe_view"]
