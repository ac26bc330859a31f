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
    # ---- bench.py timing harness: K steps between barriers, max over ranks, whole-job aggregate
    calls = []
    def step():
        calls.append(1); time.sleep(0.02 * (rank + 1))
    dt = bench.timed_steps(step, steps=3, warmup=1, world=world, device_sync=lambda: None, reduce_device=None)
    assert len(calls) == 4
    assert 0.11 < dt < 0.5, dt                      # the slower rank (2 x 0.02 x 3) sets the time on BOTH ranks
    value = bench.aggregate_value(world, 10**9, dt, 3)
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
