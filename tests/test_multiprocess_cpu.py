"""world_size-2 gloo tests (CPU): the N>1 host logic -- block sharding b % world, ordered gather
into a container (turbosqueeze_amd/sharding.py), and bench.py's distributed timing harness
(barrier, max over ranks, whole-job aggregate).  The per-block codec used here as a stand-in
for the device is the oracle, which only tests may call."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    import torch, torch.distributed as dist
    import bench
    from turbosqueeze_amd import sharding, synth
    from oracle.pyoracle import Oracle

    world, rank, local_rank = bench.init_distributed("gloo")
    assert world == 2
    # ---- sharded compress of one buffer: block b -> rank b % world, gather on rank 0 in block order
    n = 3 * (1 << 22) + 54321
    host = synth.text(n, seed=5)
    orc = Oracle()
    nb = sharding.block_count(n)
    mine = {{}}
    for b in sharding.rank_blocks(nb, rank, world):
        start, length = sharding.block_extent(b, n)
        mine[b] = (1, orc.encode_block(host[start:start + length], 1, halo=sharding.halo_of(host, b, n)))
    merged = sharding.gather_streams(mine, rank, world)
    if rank == 0:
        blob = sharding.assemble_container(n, merged)
        assert blob == orc.compress(host, 1, threads=2), "sharded container differs from the single-process one"
        total, frames = sharding.split_container(blob)
        assert total == n and len(frames) == nb
    # ---- decode side: rank r decodes frames r, r+world, ...; no ordering problem, offsets are b * 4 MiB
    blob = orc.compress(host, 1, threads=2)
    total, frames = sharding.split_container(blob)
    out = {{}}
    for b in sharding.rank_blocks(nb, rank, world):
        data, st = orc.decode_block(frames[b][1], frames[b][0])
        assert st == 0
        out[b] = (0, data)
    merged = sharding.gather_streams(out, rank, world)
    if rank == 0:
        back = b"".join(merged[b][1] for b in range(nb))
        assert back == host.tobytes()

    # ---- the one-process-per-GPU product path (ShardLayout + HostContainer + ShardedCodec: the all-gather of sizes,
    #      frame offsets, header, frames DMA'd to their place, frame walk, decode of owned frames) with the per-block
    #      device work replaced by the oracle on numpy arrays -- the host logic is the code bench.py --gpus N runs
    class OracleBlocks:
        def alloc(self, n_local):
            self.slots = np.zeros(max(n_local, 1) * sharding.OUTPUT_SZ, dtype=np.uint8)
            self.sizes = torch.zeros(max(n_local, 1), dtype=torch.int32)
        def encode(self, d_in, n_local, stride, last_len, ext):
            for k in range(n_local):
                ln = last_len if k == n_local - 1 else sharding.BLOCK_SZ
                blk = d_in[k * stride:k * stride + ln]
                stream = orc.encode_block(blk, ext, halo=bytes(d_in[k * stride + ln:k * stride + ln + 128]))
                self.slots[k * sharding.OUTPUT_SZ:k * sharding.OUTPUT_SZ + len(stream)] = np.frombuffer(stream, dtype=np.uint8)
                self.sizes[k] = len(stream)
        def sizes_tensor(self):
            return self.sizes
        def place(self, all_sizes, lay, ext, hostc):
            frame_at, total = sharding.frame_offsets(all_sizes)              # (the C helper the device path uses)
            assert total <= hostc.size
            if lay.rank == 0:
                hostc.array[:16] = np.frombuffer(b"TSQ1" + lay.nb.to_bytes(4, "little") + lay.n_total.to_bytes(8, "little"), dtype=np.uint8)
            for k, b in enumerate(lay.blocks):
                frame = int(all_sizes[b]) | (0x800000 if ext else 0)
                at = int(frame_at[b])
                hostc.array[at:at + 3] = np.frombuffer(frame.to_bytes(3, "little"), dtype=np.uint8)
                hostc.array[at + 3:at + 3 + int(all_sizes[b])] = self.slots[k * sharding.OUTPUT_SZ:k * sharding.OUTPUT_SZ + int(all_sizes[b])]
            return total
        def fetch_decode(self, hostc, container_size, lay, d_out):
            total, frame_at, sizes, ext, out_len = sharding.walk_frames(hostc.array, container_size)   # (the C helper)
            assert len(sizes) == lay.nb
            for k, b in enumerate(lay.blocks):
                at = int(frame_at[b]) + 3
                data, st = orc.decode_block(bytes(hostc.array[at:at + int(sizes[b])]), int(ext[b]))
                assert st == 0 and len(data) == int(out_len[b])
                d_out[k * sharding.BLOCK_SZ:k * sharding.BLOCK_SZ + len(data)] = np.frombuffer(data, dtype=np.uint8)
            return total
        def sync(self):
            pass

    for ext in (0, 1):
        lay = sharding.ShardLayout(n, rank, world)
        assert sorted(lay.blocks) == sharding.rank_blocks(nb, rank, world) and lay.max_blocks == 2
        name = "tsq_test_gloo_%s_%d" % (os.environ.get("MASTER_PORT", "0"), ext)
        if rank == 0:
            hc = sharding.HostContainer(name, 16 + nb * (3 + sharding.OUTPUT_SZ), create=True)
        dist.barrier()
        if rank != 0:
            hc = sharding.HostContainer(name, 16 + nb * (3 + sharding.OUTPUT_SZ), create=False)
        sc = sharding.ShardedCodec(lay, OracleBlocks(), hc, ext)
        shard = lay.pack_input(host)
        size = sc.compress(shard)
        dist.barrier()                                   # every rank's frames are in place
        want = orc.compress(host, ext, threads=2)
        assert size == len(want)
        assert bytes(hc.array[:size]) == want, "host-gathered container differs from the single-process one"
        back = np.zeros(lay.shard_bytes, dtype=np.uint8)
        assert sc.decompress(size, back) == n
        assert np.array_equal(back, lay.expected_output(host))
        dist.barrier()
        hc.close()
    # ---- bench.py timing harness: K steps between barriers, max over ranks, whole-job aggregate
    calls = []
    def step():
        calls.append(1); time.sleep(0.02 * (rank + 1))
    dt = bench.timed_steps(step, steps=3, warmup=1, world=world, device_sync=lambda: None, reduce_device=None)
    assert len(calls) == 4
    assert 0.11 < dt < 0.5, dt                      # the slower rank (2 x 0.02 x 3) sets the time on BOTH ranks
    value = bench.aggregate_value(10**9, dt, 3)
    if rank == 0:
        print(json.dumps({{"ok": True, "dt": dt, "value": value}}))
    dist.destroy_process_group()
''')


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert '"ok": true' in r.stdout


def test_sharding_helpers():
    from turbosqueeze_amd import sharding
    assert sharding.block_count(10**9) == 239
    assert sharding.rank_blocks(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((sharding.rank_blocks(239, r, 8) for r in range(8)), [])) == list(range(239))
    assert sharding.block_extent(238, 10**9) == (238 << 22, 10**9 - (238 << 22))
    data = np.arange(300, dtype=np.uint8)
    assert sharding.halo_of(data, 0, 300) == bytes(128)            # after the last block: zeros
    try:
        sharding.assemble_container(5 << 22, {0: (0, b"x")})
    except ValueError:
        pass
    else:
        raise AssertionError("missing blocks must be rejected")
    for bad in (b"", b"TSQ2" + bytes(12), b"TSQ1" + (1).to_bytes(4, "little") + bytes(8) + b"\\xff\\xff\\x7f"):
        try:
            sharding.split_container(bad)
        except ValueError:
            continue
        raise AssertionError("bad container accepted")
