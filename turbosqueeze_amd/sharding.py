"""Block sharding across ranks and ordered host-side gather.

Blocks are independent: block b goes to rank b % world (the reference's block i -> worker
i % num_cores, tsq_threads.cpp:71,463).  Each rank returns (block index, ext, stream bytes); the
gatherer lays frames out in block order, which is what compression_write_worker does
(tsq_threads.cpp:192-275).  No collective is needed on the data path; gather_streams() uses one
gather_object so that rank 0 can write the file.  Pure host logic (no device code)."""
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
