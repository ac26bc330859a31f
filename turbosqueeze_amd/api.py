"""ctypes binding of libturbosqueeze_amd.so (include/turbosqueeze_amd.h).

Two faces:
  * DeviceCodec  -- tsqa_* device-resident entry points over torch CUDA(HIP) tensors.
    torch is used only for device memory and streams.
  * tsq_encode / tsq_decode / tsq_compress_mt / tsq_decompress_mt -- the reference's own
    API names (turbosqueeze.h:508,580,657,670) over host bytes, same argument meaning.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK_SZ = 1 << 22
OUTPUT_SZ = BLOCK_SZ + (BLOCK_SZ >> 2)

ERRORS = {1: "no usable gfx950 device", 2: "HIP runtime error", 3: "bad argument", 4: "malformed container",
          5: "malformed block stream", 6: "block expanded beyond TSQ_OUTPUT_SZ",
          7: "multi-workgroup decode stalled waiting for a sibling workgroup (the container may be fine: decode again with variant 4)"}


class TsqError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"turbosqueeze_amd error {code} ({ERRORS.get(code, '?')}) {detail}".strip())


def lib_path(ab=False) -> str:
    """The product library; ab=True: the A/B library that also carries the superseded kernel generations
    (`make -C turbosqueeze_amd/csrc ab`); ab="jitter": the hand-off stress build of the encoder (`make jitter`)."""
    if isinstance(ab, str):
        return os.path.join(HERE, f"libturbosqueeze_amd_{ab}.so")
    return os.path.join(HERE, "libturbosqueeze_amd_ab.so" if ab else "libturbosqueeze_amd.so")


def build_native(force: bool = False) -> None:
    """Compile every HIP source for gfx950 into turbosqueeze_amd/*.so (in-tree)."""
    csrc = os.path.join(HERE, "csrc")
    args = ["make", "-C", csrc, "all", "ab", "jitter"]
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)


def source_fingerprint() -> str:
    """Hash of the kernel and host sources the native libraries are built from (csrc/*.cuh, *.hip, *.h, ab/*): what a stored
    profile must carry to be attributed to the code that is running (bench.py: roofline.traffic)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(HERE, "csrc")
    for fn in sorted(glob.glob(os.path.join(csrc, "*.cuh")) + glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) +
                     glob.glob(os.path.join(csrc, "ab", "*.cuh"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def build_info(ab=False) -> str:
    """tsqa_build_info() of a library: which experiment / instrumentation switches it was compiled with (the product: none)."""
    return lib(ab).tsqa_build_info().decode()


_libs = {}


def lib(ab=False) -> C.CDLL:
    """Load the native library.  Fails loudly when it is missing: there is no other path."""
    if ab in _libs:
        return _libs[ab]
    path = lib_path(ab)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); turbosqueeze_amd has no CPU fallback")
    L = C.CDLL(path)
    vp, u8pp, szp = C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
    L.tsqa_create.restype = C.c_int
    L.tsqa_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.tsqa_destroy.restype = None
    L.tsqa_destroy.argtypes = [vp]
    L.tsqa_last_error.restype = C.c_char_p
    L.tsqa_last_error.argtypes = [vp]
    L.tsqa_device_id.restype = C.c_int
    L.tsqa_device_id.argtypes = [vp]
    L.tsqa_block_count.restype = C.c_size_t
    L.tsqa_block_count.argtypes = [C.c_size_t]
    L.tsqa_build_info.restype = C.c_char_p
    L.tsqa_build_info.argtypes = []
    L.tsqa_container_bound.restype = C.c_size_t
    L.tsqa_container_bound.argtypes = [C.c_size_t]
    L.tsqa_compress_device.restype = C.c_int
    L.tsqa_compress_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, szp, C.c_uint32, vp]
    L.tsqa_compress_device_async.restype = C.c_int
    L.tsqa_compress_device_async.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_uint32, vp]
    L.tsqa_decompress_device.restype = C.c_int
    L.tsqa_decompress_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, szp, vp]
    L.tsqa_decompress_device_async.restype = C.c_int
    L.tsqa_decompress_device_async.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, vp, vp, vp]
    L.tsqa_profile_enable.restype = C.c_int
    L.tsqa_profile_enable.argtypes = [vp, C.c_int]
    L.tsqa_profile_read.restype = C.c_int
    L.tsqa_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.tsqa_set_kernel_variant.restype = None
    L.tsqa_set_kernel_variant.argtypes = [vp, C.c_int, C.c_int]
    L.tsqa_set_decode_wait_limit.restype = None
    L.tsqa_set_decode_wait_limit.argtypes = [vp, C.c_uint32]
    # reference API
    L.tsqAllocateContext.restype = vp
    L.tsqDeallocateContext.argtypes = [vp]
    L.tsqInit.argtypes = [vp]
    L.tsqEncode.restype = None
    L.tsqEncode.argtypes = [vp, vp, vp, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
    L.tsqDecode.restype = None
    L.tsqDecode.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
    L.tsqAllocateContextCompression_MT.restype = vp
    L.tsqAllocateContextCompression_MT.argtypes = [C.c_bool]
    L.tsqDeallocateContextCompression_MT.argtypes = [vp]
    L.tsqAllocateContextDecompression_MT.restype = vp
    L.tsqAllocateContextDecompression_MT.argtypes = [C.c_bool]
    L.tsqDeallocateContextDecompression_MT.argtypes = [vp]
    L.tsqCompress_MT.restype = C.c_bool
    L.tsqCompress_MT.argtypes = [vp, vp, C.c_size_t, C.c_bool, u8pp, szp, C.c_bool, C.c_bool, C.c_uint32]
    L.tsqDecompress_MT.restype = C.c_bool
    L.tsqDecompress_MT.argtypes = [vp, vp, C.c_size_t, C.c_bool, u8pp, szp, C.c_bool]
    L.tsqa_compress_async_cb.restype = C.c_uint32
    L.tsqa_compress_async_cb.argtypes = [vp, vp, C.c_size_t, C.c_bool, u8pp, szp, C.c_bool, C.c_bool, C.c_uint32, vp, vp, vp]
    L.tsqa_decompress_async_cb.restype = C.c_uint32
    L.tsqa_decompress_async_cb.argtypes = [vp, vp, C.c_size_t, C.c_bool, u8pp, szp, C.c_bool, vp, vp, vp]
    L.tsqa_profile_read_calls.restype = C.c_int
    L.tsqa_profile_read_calls.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.tsqa_encode_blocks_async.restype = C.c_int
    L.tsqa_encode_blocks_async.argtypes = [vp, vp, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.tsqa_decode_blocks_async.restype = C.c_int
    L.tsqa_decode_blocks_async.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp]
    L.tsqa_frames_to_host_async.restype = C.c_int
    L.tsqa_frames_to_host_async.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.tsqa_frames_from_host_async.restype = C.c_int
    L.tsqa_frames_from_host_async.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, vp]
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.tsqa_frame_offsets.restype = C.c_int
    L.tsqa_frame_offsets.argtypes = [vp, C.c_uint32, vp, u64p]
    L.tsqa_walk_frames.restype = C.c_int
    L.tsqa_walk_frames.argtypes = [vp, C.c_size_t, C.c_uint32, vp, vp, vp, vp, u32p, u64p]
    L.tsqa_sharded_place_async.restype = C.c_int
    L.tsqa_sharded_place_async.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, u64p, vp]
    L.tsqa_sharded_fetch_decode_async.restype = C.c_int
    L.tsqa_sharded_fetch_decode_async.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, C.c_size_t, vp, u64p, vp]
    L.tsqa_sharded_decode_again_async.restype = C.c_int
    L.tsqa_sharded_decode_again_async.argtypes = [vp, vp, vp, vp, vp]
    L.tsqa_encode_lookahead_state.restype = C.c_int
    L.tsqa_encode_lookahead_state.argtypes = []
    L.tsqa_copy_probe_shape.restype = C.c_char_p
    L.tsqa_copy_probe_shape.argtypes = [vp]
    L.tsqa_measure_copy.restype = C.c_int
    L.tsqa_measure_copy.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tsqCompress.restype = None
    L.tsqCompress.argtypes = [vp, vp, C.c_bool, C.c_uint32]
    L.tsqDecompress.restype = None
    L.tsqDecompress.argtypes = [vp, vp]
    _libs[ab] = L
    return L


DONE_FN = C.CFUNCTYPE(None, C.c_uint32, C.c_bool, C.c_void_p)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint32, C.c_double, C.c_void_p)
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def container_bound(n: int) -> int:
    return int(lib().tsqa_container_bound(n))


class DeviceCodec:
    """Device-resident compress/decompress over torch uint8 CUDA tensors (tsqa_* C ABI)."""

    def __init__(self, device: int = -1, ab=False):
        import torch
        if not torch.cuda.is_available():
            raise TsqError(1, "torch sees no GPU")
        self.torch = torch
        self.L = lib(ab)
        self.h = C.c_void_p()
        if device < 0:
            device = torch.cuda.current_device()
        rc = self.L.tsqa_create(device, C.byref(self.h))
        if rc:
            raise TsqError(rc)
        self.device = torch.device("cuda", device)
        self._size = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)

    def close(self):
        if self.h:
            self.L.tsqa_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        return TsqError(rc, self.L.tsqa_last_error(self.h).decode())

    def last_error(self) -> str:
        return self.L.tsqa_last_error(self.h).decode()

    def set_variant(self, enc: int, dec: int) -> None:
        self.L.tsqa_set_kernel_variant(self.h, enc, dec)

    def set_decode_wait_limit(self, polls: int) -> None:
        """polls a several-workgroups-per-block decode waits for a sibling workgroup before it reports TSQA_ERR_STALL"""
        self.L.tsqa_set_decode_wait_limit(self.h, polls)

    def profile(self, on: bool) -> None:
        self.L.tsqa_profile_enable(self.h, 1 if on else 0)

    def profile_read(self):
        """-> (encode_ms_sum, encode_launches, decode_ms_sum, decode_launches) since the last read."""
        em, dm, en, dn = C.c_double(0), C.c_double(0), C.c_uint32(0), C.c_uint32(0)
        self.L.tsqa_profile_read(self.h, C.byref(em), C.byref(en), C.byref(dm), C.byref(dn))
        return em.value, en.value, dm.value, dn.value

    def profile_read_calls(self):
        """-> (compress_ms_sum, calls, decompress_ms_sum, calls): whole calls (encode + pack; frame walk + decode)."""
        em, dm, en, dn = C.c_double(0), C.c_double(0), C.c_uint32(0), C.c_uint32(0)
        self.L.tsqa_profile_read_calls(self.h, C.byref(em), C.byref(en), C.byref(dm), C.byref(dn))
        return em.value, en.value, dm.value, dn.value

    def measure_copy(self, nbytes: int = 1 << 30, reps: int = 7):
        """-> (best, median) GB/s of a plain device copy kernel, bytes read + written."""
        best, med = C.c_double(0), C.c_double(0)
        rc = self.L.tsqa_measure_copy(self.h, nbytes, reps, C.byref(best), C.byref(med))
        if rc:
            raise self._err(rc)
        return best.value, med.value

    def copy_probe_shape(self) -> str:
        """Which launch shape the last measure_copy chose."""
        return self.L.tsqa_copy_probe_shape(self.h).decode()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # ---- sharded operation: this device owns some of a job's blocks (tsqa_*_blocks_async) ----
    def encode_blocks_async(self, d_in, n_blocks: int, stride: int, last_len: int, ext: int, d_slots, d_sizes):
        rc = self.L.tsqa_encode_blocks_async(self.h, d_in.data_ptr(), n_blocks, stride, last_len, int(ext), d_slots.data_ptr(),
                                             d_sizes.data_ptr(), self._status.data_ptr(), self._stream())
        if rc:
            raise self._err(rc)

    def decode_blocks_async(self, d_streams, d_frames, n_blocks: int, d_out):
        rc = self.L.tsqa_decode_blocks_async(self.h, d_streams.data_ptr(), d_frames.data_ptr(), n_blocks, d_out.data_ptr(),
                                             self._status.data_ptr(), self._stream())
        if rc:
            raise self._err(rc)

    def frames_to_host_async(self, d_slots, sizes, frame_at, ext: int, host_ptr: int):
        """sizes (uint32) / frame_at (uint64): contiguous numpy arrays, one entry per owned block."""
        rc = self.L.tsqa_frames_to_host_async(self.h, d_slots.data_ptr(), sizes.ctypes.data, frame_at.ctypes.data, len(sizes), int(ext),
                                              host_ptr, self._stream())
        if rc:
            raise self._err(rc)

    def frames_from_host_async(self, host_ptr: int, frame_at, sizes, d_streams):
        rc = self.L.tsqa_frames_from_host_async(self.h, host_ptr, frame_at.ctypes.data, sizes.ctypes.data, len(sizes), d_streams.data_ptr(),
                                                self._stream())
        if rc:
            raise self._err(rc)

    def sharded_place_async(self, d_slots, all_sizes, n_total: int, rank: int, world: int, ext: int, host_ptr: int, host_cap: int) -> int:
        """all_sizes: contiguous numpy uint32, one entry per block of the job.  -> container size."""
        total = C.c_uint64(0)
        rc = self.L.tsqa_sharded_place_async(self.h, d_slots.data_ptr(), all_sizes.ctypes.data, len(all_sizes), n_total, rank, world, int(ext),
                                             host_ptr, host_cap, C.byref(total), self._stream())
        if rc:
            raise self._err(rc)
        return int(total.value)

    def sharded_fetch_decode_async(self, host_ptr: int, container_size: int, rank: int, world: int, d_streams, d_out) -> int:
        """-> the job's uncompressed size (from the container header)."""
        total = C.c_uint64(0)
        rc = self.L.tsqa_sharded_fetch_decode_async(self.h, host_ptr, container_size, rank, world, d_streams.data_ptr(), d_streams.numel(),
                                                    d_out.data_ptr(), d_out.numel(), self._status.data_ptr(), C.byref(total), self._stream())
        if rc:
            raise self._err(rc)
        return int(total.value)

    def sharded_decode_again_async(self, d_streams, d_out) -> None:
        """After status() == 7 (TSQA_ERR_STALL) behind sharded_fetch_decode_async: the owned frames, still on the device, once
        more on one workgroup per block."""
        rc = self.L.tsqa_sharded_decode_again_async(self.h, d_streams.data_ptr(), d_out.data_ptr(), self._status.data_ptr(), self._stream())
        if rc:
            raise self._err(rc)

    def status(self) -> int:
        return int(self._status.item())

    def compress(self, src, ext: int, out=None):
        """src: uint8 CUDA tensor.  Returns a uint8 CUDA tensor view holding the .tsq container."""
        n = src.numel()
        if out is None:
            out = self.torch.empty(container_bound(n), dtype=self.torch.uint8, device=self.device)
        sz = C.c_size_t(0)
        rc = self.L.tsqa_compress_device(self.h, src.data_ptr(), n, out.data_ptr(), out.numel(), C.byref(sz), int(ext), self._stream())
        if rc:
            raise self._err(rc)
        return out[: sz.value]

    def decompress(self, blob, out=None, out_cap=None):
        n = blob.numel()
        if out is None:
            total = int.from_bytes(bytes(blob[8:16].cpu().numpy()), "little") if out_cap is None else out_cap
            out = self.torch.empty(max(total, 1), dtype=self.torch.uint8, device=self.device)
        sz = C.c_size_t(0)
        rc = self.L.tsqa_decompress_device(self.h, blob.data_ptr(), n, out.data_ptr(), out.numel(), C.byref(sz), self._stream())
        if rc:
            raise self._err(rc)
        return out[: sz.value]

    # asynchronous forms: nothing is synchronised; results land in self._size / self._status
    def compress_async(self, src, ext: int, out):
        rc = self.L.tsqa_compress_device_async(self.h, src.data_ptr(), src.numel(), out.data_ptr(), out.numel(),
                                               self._size.data_ptr(), self._status.data_ptr(), int(ext), self._stream())
        if rc:
            raise self._err(rc)

    def decompress_async(self, blob, n_blocks: int, out):
        rc = self.L.tsqa_decompress_device_async(self.h, blob.data_ptr(), blob.numel(), n_blocks, out.data_ptr(), out.numel(),
                                                 self._size.data_ptr(), self._status.data_ptr(), self._stream())
        if rc:
            raise self._err(rc)

    def last_size_status(self):
        return int(self._size.item()), int(self._status.item())


# ---------------------------------------------------------------------------
# The reference's API over host bytes
# ---------------------------------------------------------------------------

def tsq_encode(data: bytes, ext: int, halo: bytes = b"") -> bytes:
    """tsqEncode (turbosqueeze.h:657): one block (<= 4 MiB) -> block stream.  Like the reference, the
    encoder looks a few bytes past the block (tsq_encode.cpp:74,126): `halo` is what follows the block
    in the caller's buffer (the next block's first bytes); zeros otherwise (canonical conditions)."""
    L = lib()
    n = len(data)
    src = C.create_string_buffer(bytes(data) + bytes(halo)[:128].ljust(128, b"\0"), n + 128)
    dst = C.create_string_buffer(OUTPUT_SZ)
    sz = C.c_uint32(0)
    ctx = L.tsqAllocateContext()
    try:
        L.tsqInit(ctx)
        L.tsqEncode(ctx, src, dst, C.byref(sz), n, int(ext))
    finally:
        L.tsqDeallocateContext(ctx)
    if sz.value == 0:
        raise TsqError(2, "tsqEncode produced no output (no device?)")
    return dst.raw[: sz.value]


def tsq_decode(stream: bytes, ext: int) -> bytes:
    """tsqDecode (turbosqueeze.h:670): block stream -> block; b'' when *outputSize == 0."""
    L = lib()
    src = C.create_string_buffer(bytes(stream), len(stream))
    dst = C.create_string_buffer(BLOCK_SZ + 256)
    sz = C.c_uint32(0)
    L.tsqDecode(src, dst, C.byref(sz), len(stream), int(ext))
    return dst.raw[: sz.value]


def _take_malloced(ptr: C.c_void_p, size: int) -> bytes:
    try:
        return C.string_at(ptr, size)
    finally:
        _libc.free(ptr)


def tsq_compress_mt(data: bytes, ext: bool, progress=None):
    """tsqAllocateContextCompression_MT + tsqCompress_MT (memory -> memory) + deallocate."""
    L = lib()
    ctx = L.tsqAllocateContextCompression_MT(False)
    if not ctx:
        raise TsqError(1, "tsqAllocateContextCompression_MT returned NULL")
    try:
        buf = C.create_string_buffer(bytes(data), len(data))
        out, sz = C.c_void_p(), C.c_size_t(0)
        ok = L.tsqCompress_MT(ctx, buf, len(data), False, C.byref(out), C.byref(sz), False, bool(ext), 0)
        if not ok:
            return None
        return _take_malloced(out, sz.value)
    finally:
        L.tsqDeallocateContextCompression_MT(ctx)


def tsq_decompress_mt(blob: bytes):
    L = lib()
    ctx = L.tsqAllocateContextDecompression_MT(False)
    if not ctx:
        raise TsqError(1, "tsqAllocateContextDecompression_MT returned NULL")
    try:
        buf = C.create_string_buffer(bytes(blob), len(blob))
        out, sz = C.c_void_p(), C.c_size_t(0)
        ok = L.tsqDecompress_MT(ctx, buf, len(blob), False, C.byref(out), C.byref(sz), False)
        if not ok:
            return None
        return _take_malloced(out, sz.value)
    finally:
        L.tsqDeallocateContextDecompression_MT(ctx)
