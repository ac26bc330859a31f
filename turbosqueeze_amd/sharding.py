"""Block sharding across ranks and ordered host-side gather.

Blocks are independent: block b goes to rank b % world (the reference's block i -> worker
i % num_cores, tsq_threads.cpp:71,463).  Each rank returns (block index, ext, stream bytes); the
gatherer lays frames out in block order, which is what compression_write_worker does
(tsq_threads.cpp:192-275).  No collective is needed on the data path; gather_streams() uses one
gather_object so that rank 0 can write the file.  The first half of this file is pure host logic (no device code).

The second half is the one-process-per-GPU form of the same thing: `ShardLayout` (which blocks a rank owns
and how they lie in its HBM), `HostContainer` (one container in host memory shared by the ranks of a node --
every rank DMAs its frames straight to their final place, so the "gather" is the prefix sum of the sizes and
nothing else) and `ShardedCodec` (encode owned blocks -> all-gather of the u32 sizes -> frames to the host
container; frame walk -> owned frames to HBM -> decode).  The only collective is that all-gather of sizes."""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

BLOCK_SZ = 1 << 22


def block_count(n: int) -> int:
    return (n + BLOCK_SZ - 1) // BLOCK_SZ


def block_owner(b: int, world: int) -> int:
    return b % world


def rank_blocks(n_blocks: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_blocks, world))


def block_extent(b: int, n: int) -> Tuple[int, int]:
    """(start, length) of block b in an n-byte input."""
    start = b * BLOCK_SZ
    return start, min(BLOCK_SZ, n - start)


def halo_of(data, b: int, n: int, halo: int = 128) -> bytes:
    """The bytes that follow block b (the encoder's look-ahead): the next block's first bytes,
    zeros after the last block (canonical conditions)."""
    start, length = block_extent(b, n)
    tail = bytes(data[start + length:start + length + halo])
    return tail.ljust(halo, b"\0")


def assemble_container(total: int, frames: Dict[int, Tuple[int, bytes]]) -> bytes:
    """frames: block index -> (ext, stream).  Returns the .tsq container (header + frames in order)."""
    nb = block_count(total)
    if sorted(frames) != list(range(nb)):
        raise ValueError("missing or extra blocks: %r" % sorted(frames)[:8])
    out = bytearray(b"TSQ1" + nb.to_bytes(4, "little") + total.to_bytes(8, "little"))
    for b in range(nb):
        ext, stream = frames[b]
        frame = len(stream) | (0x800000 if ext else 0)
        out += frame.to_bytes(3, "little")
        out += stream
    return bytes(out)


def split_container(blob: bytes) -> Tuple[int, List[Tuple[int, bytes]]]:
    """-> (total uncompressed, [(ext, stream)] in block order).  Raises ValueError on a bad container."""
    if len(blob) < 16 or blob[:4] != b"TSQ1":
        raise ValueError("bad magic")
    nb = int.from_bytes(blob[4:8], "little")
    total = int.from_bytes(blob[8:16], "little")
    at, frames = 16, []
    for _ in range(nb):
        if at + 3 > len(blob):
            raise ValueError("truncated")
        frame = int.from_bytes(blob[at:at + 3], "little")
        ln = frame & 0x7FFFFF
        if ln < 3 or at + 3 + ln > len(blob):
            raise ValueError("bad frame")
        frames.append((frame >> 23, blob[at + 3:at + 3 + ln]))
        at += 3 + ln
    return total, frames


def gather_streams(local: Dict[int, Tuple[int, bytes]], rank: int, world: int, dst: int = 0):
    """Gather every rank's {block: (ext, stream)} on `dst` (torch.distributed must be initialised
    when world > 1).  Returns the merged dict on dst, None elsewhere."""
    if world == 1:
        return dict(local)
    import torch.distributed as dist
    parts = [None] * world if rank == dst else None
    dist.gather_object(local, parts, dst=dst)
    if rank != dst:
        return None
    merged: Dict[int, Tuple[int, bytes]] = {}
    for p in parts:
        merged.update(p)
    return merged


# ---------------------------------------------------------------------------------------------
# One process per GPU
# ---------------------------------------------------------------------------------------------
HALO = 128
OUTPUT_SZ = BLOCK_SZ + (BLOCK_SZ >> 2)


class ShardLayout:
    """Block b of an n_total-byte job belongs to rank b % world.  In a rank's memory its blocks lie back to
    back, each IMMEDIATELY followed by its 128 look-ahead bytes (what tsqa_encode_blocks_async expects):
    stride = BLOCK_SZ + HALO; decoded blocks lie back to back at stride BLOCK_SZ."""

    def __init__(self, n_total: int, rank: int, world: int):
        self.n_total, self.rank, self.world = n_total, rank, world
        self.nb = block_count(n_total)
        self.blocks = rank_blocks(self.nb, rank, world)
        self.stride = BLOCK_SZ + HALO
        self.lengths = [block_extent(b, n_total)[1] for b in self.blocks]
        self.last_len = self.lengths[-1] if self.lengths else 0
        if any(l != BLOCK_SZ for l in self.lengths[:-1]):
            raise ValueError("only a rank's last block may be short")
        self.max_blocks = (self.nb + world - 1) // world          # the most blocks any rank owns

    @property
    def n_local(self) -> int:
        return len(self.blocks)

    @property
    def shard_bytes(self) -> int:
        return sum(self.lengths)

    def pack_input(self, data):
        """The rank's input in device layout, from the whole job's bytes (a numpy uint8 array)."""
        import numpy as np
        out = np.zeros(max(self.n_local, 1) * self.stride, dtype=np.uint8)
        for k, b in enumerate(self.blocks):
            start, length = block_extent(b, self.n_total)
            out[k * self.stride:k * self.stride + length] = data[start:start + length]
            out[k * self.stride + length:k * self.stride + length + HALO] = np.frombuffer(halo_of(data, b, self.n_total, HALO), dtype=np.uint8)
        return out

    def expected_output(self, data):
        """What the rank's decoded blocks must equal (blocks back to back)."""
        import numpy as np
        parts = [data[block_extent(b, self.n_total)[0]:block_extent(b, self.n_total)[0] + block_extent(b, self.n_total)[1]] for b in self.blocks]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)


def frame_offsets(all_sizes):
    """Container offset of every block's frame bytes: 16 + sum over earlier blocks of (3 + size)
    (tsq_threads.cpp:226-239).  -> (uint64 array of nb offsets, container size).  (C: tsqa_frame_offsets.)"""
    import ctypes as C
    import numpy as np
    from . import api
    sizes = np.ascontiguousarray(all_sizes, dtype=np.uint32)
    frame_at = np.zeros(len(sizes), dtype=np.uint64)
    total = C.c_uint64(0)
    rc = api.lib().tsqa_frame_offsets(sizes.ctypes.data, len(sizes), frame_at.ctypes.data, C.byref(total))
    if rc:
        raise ValueError("a block stream size is out of range")
    return frame_at, int(total.value)


def walk_frames(container, limit: int):
    """The serial frame walk of a container in host memory (tsq_threads.cpp:513-524) over a numpy uint8 view.
    -> (total, frame_at uint64[nb], sizes uint32[nb], ext uint32[nb], out_len uint32[nb]); raises ValueError.
    (C: tsqa_walk_frames -- no Python loop over the blocks.)"""
    import ctypes as C
    import numpy as np
    from . import api
    cap = max((limit - 16) // 6, 1) if limit >= 16 else 1
    frame_at = np.zeros(cap, dtype=np.uint64)
    sizes = np.zeros(cap, dtype=np.uint32)
    ext = np.zeros(cap, dtype=np.uint32)
    out_len = np.zeros(cap, dtype=np.uint32)
    nb, total = C.c_uint32(0), C.c_uint64(0)
    rc = api.lib().tsqa_walk_frames(container.ctypes.data, limit, cap, frame_at.ctypes.data, sizes.ctypes.data, ext.ctypes.data,
                                    out_len.ctypes.data, C.byref(nb), C.byref(total))
    if rc:
        raise ValueError("malformed container")
    k = nb.value
    return int(total.value), frame_at[:k], sizes[:k], ext[:k], out_len[:k]


class HostContainer:
    """One buffer in host memory shared by the ranks of a node (a file in /dev/shm mapped by every rank;
    hipHostRegister'ed where a GPU is present so that the DMAs run at full speed)."""

    def __init__(self, name: str, size: int, create: bool):
        import mmap
        import os
        self.path = os.path.join("/dev/shm", name)
        self.size = size
        self.created = create
        # the creator refuses an existing name (no truncation through a planted link) and reserves the space now: a /dev/shm that is
        # too small raises OSError here, not SIGBUS at the first DMA
        flags = os.O_RDWR | os.O_NOFOLLOW | (os.O_CREAT | os.O_EXCL if create else 0)
        if create:
            try:
                os.unlink(self.path)                      # a leftover of a killed run of ours (same uid: the directory is sticky)
            except FileNotFoundError:
                pass
        fd = os.open(self.path, flags, 0o600)
        try:
            if create:
                os.posix_fallocate(fd, 0, size)
            self.map = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        import numpy as np
        self.array = np.frombuffer(self.map, dtype=np.uint8)
        self.ptr = self.array.ctypes.data
        self.registered = False

    def register(self):
        import torch
        rc = torch.cuda.cudart().cudaHostRegister(self.ptr, self.size, 0)
        if int(rc) != 0:
            raise RuntimeError(f"hipHostRegister failed: {rc}")
        self.registered = True

    def close(self):
        import os
        if self.registered:
            import torch
            torch.cuda.cudart().cudaHostUnregister(self.ptr)
            self.registered = False
        self.array = None
        try:
            self.map.close()
        except BufferError as e:                          # a numpy view of the mapping is still alive somewhere: say so
            import warnings
            warnings.warn(f"HostContainer {self.path}: mapping still referenced at close ({e})")
        if self.created:
            try:
                os.unlink(self.path)
            except OSError:
                pass


class DeviceBlocks:
    """The per-block device operations ShardedCodec needs, over tsqa_*_blocks_async (one GPU)."""

    def __init__(self, codec, collective_device=None):
        import torch
        self.codec, self.torch, self.device = codec, torch, codec.device
        self.collective_device = collective_device or codec.device      # nccl: the GPU; gloo: the CPU

    def alloc(self, n_local: int):
        t = self.torch
        self.slots = t.empty(max(n_local, 1) * OUTPUT_SZ, dtype=t.uint8, device=self.device)
        self.sizes = t.zeros(max(n_local, 1), dtype=t.int32, device=self.device)

    def encode(self, d_in, n_local, stride, last_len, ext):
        self.codec.encode_blocks_async(d_in, n_local, stride, last_len, ext, self.slots, self.sizes)

    def sizes_tensor(self):
        # the sizes are needed on the host side of the gather now: wait for the encode kernel (it may run on the context's own
        # stream, which torch's copies do not order themselves behind)
        self.torch.cuda.synchronize(self.device)
        return self.sizes if self.collective_device == self.device else self.sizes.to(self.collective_device)

    def place(self, all_sizes, layout, ext, host) -> int:
        """Every owned frame to its place in the host container (tsqa_sharded_place_async).  -> container size."""
        return self.codec.sharded_place_async(self.slots, all_sizes, layout.n_total, layout.rank, layout.world, ext, host.ptr, host.size)

    def fetch_decode(self, host, container_size, layout, d_out) -> int:
        """Frame walk, owned frames to the device, decode (tsqa_sharded_fetch_decode_async).  -> uncompressed size of the job."""
        if container_size > host.size:
            raise ValueError(f"container size {container_size} exceeds the host mapping ({host.size} B)")
        self._last_out = d_out
        return self.codec.sharded_fetch_decode_async(host.ptr, container_size, layout.rank, layout.world, self.slots, d_out)

    STALL = 7                                             # TSQA_ERR_STALL

    def sync(self):
        self.torch.cuda.synchronize(self.device)
        st = self.codec.status()
        if st == self.STALL and getattr(self, "_last_out", None) is not None:
            # A sibling workgroup of the several-workgroups-per-block decoder did not get onto the GPU in time (a GPU of a sharded job
            # holds few blocks, so that decoder is always the first choice): the container is not at fault.  The frames are still
            # on the device: once more on one workgroup per block, which waits for nobody.
            self.stall_retries = getattr(self, "stall_retries", 0) + 1
            self.codec.sharded_decode_again_async(self.slots, self._last_out)
            self.torch.cuda.synchronize(self.device)
            st = self.codec.status()
        self._last_out = None
        if st:
            raise RuntimeError(f"device status {st}")


FRAME_DTYPE = [("stream_at", "<u8"), ("out_at", "<u8"), ("stream_len", "<u4"), ("ext", "<u4"), ("out_len", "<u4"), ("pad", "<u4")]


class ShardedCodec:
    """One job, its blocks dealt round-robin over the ranks (block b -> rank b % world), the .tsq container
    gathered in host memory.  `blocks` does the per-block work (DeviceBlocks on a GPU); everything else here --
    who owns what, the all-gather of sizes, where every frame goes, the frame walk -- is host logic and is
    what tests/test_multiprocess_cpu.py drives with world_size 2 on gloo."""

    def __init__(self, layout: ShardLayout, blocks, host: HostContainer, ext: int):
        self.layout, self.blocks, self.host, self.ext = layout, blocks, host, int(ext)
        blocks.alloc(layout.n_local)
        # wall seconds spent per phase on this rank (the calls below already end in a device sync where one is needed)
        self.seconds = {"encode": 0.0, "size_gather": 0.0, "place_d2h": 0.0, "fetch_h2d_decode": 0.0}

    def _all_sizes(self):
        """Every block's stream size, on every rank: ONE all-gather of max_blocks u32 per rank."""
        import time
        import numpy as np
        lay = self.layout
        t0 = time.perf_counter()
        mine = self.blocks.sizes_tensor()                # (waits for the encode kernel)
        t1 = time.perf_counter()
        self.seconds["encode"] += t1 - t0
        try:
            return self._gather_sizes(mine)
        finally:
            self.seconds["size_gather"] += time.perf_counter() - t1

    def _gather_sizes(self, mine):
        import numpy as np
        lay = self.layout
        if lay.world == 1 and not getattr(self, "always_gather", False):
            return np.ascontiguousarray(mine.cpu().numpy().astype(np.uint32)[:lay.n_local])
        import torch
        import torch.distributed as dist
        pad = torch.zeros(lay.max_blocks, dtype=mine.dtype, device=mine.device)
        pad[:lay.n_local] = mine[:lay.n_local]
        every = torch.empty(lay.world * lay.max_blocks, dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(every, pad)
        table = every.cpu().numpy().astype(np.uint32).reshape(lay.world, lay.max_blocks)
        # block b = k * world + r is entry (r, k): the transpose, flattened, is the job's block order
        return np.ascontiguousarray(table.T.reshape(-1)[:lay.nb])

    def compress(self, d_shard) -> int:
        """Encode the owned blocks and put their frames into the host container.  Returns the container size
        (the same on every rank).  The caller barriers before anyone reads the container."""
        lay = self.layout
        if lay.n_local:
            self.blocks.encode(d_shard, lay.n_local, lay.stride, lay.last_len, self.ext)
        import time
        sizes = self._all_sizes()
        t0 = time.perf_counter()
        total = self.blocks.place(sizes, lay, self.ext, self.host)
        self.blocks.sync()
        self.seconds["place_d2h"] += time.perf_counter() - t0
        return total

    def decompress(self, container_size: int, d_out) -> int:
        """Walk the container, bring the owned frames to the device, decode them back to back into d_out.
        Returns the job's uncompressed size."""
        import time
        lay = self.layout
        t0 = time.perf_counter()
        total = self.blocks.fetch_decode(self.host, container_size, lay, d_out)
        if total != lay.n_total:
            raise ValueError("container does not match the layout")
        self.blocks.sync()
        self.seconds["fetch_h2d_decode"] += time.perf_counter() - t0
        return total
