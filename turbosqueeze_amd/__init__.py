"""turbosqueeze_amd -- MI355X (gfx950) implementation of turbosqueeze's per-block
encode/decode hot path.  This Python package is a thin ctypes face over
libturbosqueeze_amd.so (hand-written HIP kernels behind the C ABI of
include/turbosqueeze_amd.h); it holds no codec logic and has no CPU fallback:
if the shared library or a gfx950 device is missing, calls raise.
"""
from .api import (  # noqa: F401
    BLOCK_SZ,
    OUTPUT_SZ,
    DeviceCodec,
    TsqError,
    build_info,
    build_native,
    container_bound,
    lib,
    lib_path,
    source_fingerprint,
    tsq_compress_mt,
    tsq_decode,
    tsq_decompress_mt,
    tsq_encode,
)
from . import synth  # noqa: F401
