#!/usr/bin/env python3
"""bench.py -- encode+decode GB/s of the turbosqueeze hot path on MI355X.

One "step" = one pass of the hot path over one batch: compress an enwik9-shaped 10^9-byte buffer
that is already resident in HBM into a .tsq container (encode kernel + container pack), then
decompress that container back (frame walk + decode kernel).  value = uncompressed bytes of all
ranks / wall time per step (GB/s = 1e9 B/s), weak scaling: every rank owns its own 10^9-byte shard
(blocks are independent, SURVEY.md 8e -- no collective on the data path; RCCL is used only for the
timing barrier and the max over ranks).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events recorded inside the
library on the launch stream) and, at N=1, `cpu_baseline` (the reference's own tsqEncode/tsqDecode
compiled into oracle/_ref, or the oracle port, on the host cores over a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic(kernel_substr: str):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC summary of this same
    command (tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes), corrected as
    MI355X_MICROARCH.md prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
    bench.py cannot read PMCs itself; returns (bytes, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        f = next(v["per_dispatch"] for k, v in d["fetch"].items() if kernel_substr in k)
        w = next(v["per_dispatch"] for k, v in d["write"].items() if kernel_substr in k)
        return int((2 * f + w) * 1024), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def cpu_baseline(sample_bytes: int, ext: int):
    """Block-parallel CPU encode+decode of a bounded sample of the same workload on every host
    thread (oracle/tsq_oracle.c: tsqo_cpubench, a pthread pool with block i -> thread i % T like
    tsq_threads.cpp:71).  Runs the reference's own tsqEncode/tsqDecode (oracle/_ref) when that
    library is present ("reference"), else the oracle port ("port")."""
    import ctypes as C

    import turbosqueeze_amd as tsq
    from oracle import pyoracle

    cores = os.cpu_count() or 1
    host = tsq.synth.text(sample_bytes, seed=1, pad=256)
    orc = pyoracle.Oracle()
    fn = orc.L.tsqo_cpubench
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int,
                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    enc = dec = None
    kind = "port"
    if pyoracle.Reference.available():
        ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtsq_ref.so"))
        enc = C.cast(ref.tsqEncode, C.c_void_p)
        dec = C.cast(ref.tsqDecode, C.c_void_p)
        kind = "reference"
    best = None
    for threads in sorted({cores, max(cores // 2, 1)}):           # SMT siblings do not always help: keep the better
        te, td, cb = C.c_double(0), C.c_double(0), C.c_uint64(0)
        bad = fn(enc, dec, host.ctypes.data, sample_bytes, ext, threads, 2, C.byref(te), C.byref(td), C.byref(cb))
        r = {"threads": threads, "te": te.value, "td": td.value, "ok": bad == 0, "ratio": cb.value / sample_bytes}
        if best is None or r["te"] + r["td"] < best["te"] + best["td"]:
            best = r
    te, td = best["te"], best["td"]
    return {
        "value": round(sample_bytes / (te + td) / 1e9, 4), "unit": "GB/s", "cores": best["threads"], "kind": kind,
        "sample": f"{sample_bytes} B of the same enwik9-shaped text, block-parallel pthreads (block i -> thread i % T), "
                  f"best of 2 warm passes each for encode and decode; host has {cores} hardware threads; "
                  f"roundtrip_ok={best['ok']} ratio={best['ratio']:.4f}",
        "encode_GBps": round(sample_bytes / te / 1e9, 4), "decode_GBps": round(sample_bytes / td / 1e9, 4),
    }


def init_distributed(backend: str):
    """(world, rank, local_rank); one process per GPU, rendezvous on 127.0.0.1."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, local_rank


def timed_steps(step, steps: int, warmup: int, world: int, device_sync, reduce_device=None):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds (the contract of the driver)."""
    import torch
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        device_sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def aggregate_value(world: int, bytes_per_rank: int, dt: float, steps: int):
    """Whole-job GB/s over all ranks (weak scaling: every rank owns bytes_per_rank)."""
    return world * bytes_per_rank / (dt / steps) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1_000_000_000, help="uncompressed bytes per GPU (enwik9 = 1e9)")
    ap.add_argument("--ext", type=int, default=0, help="0 = --no-ext fast level (the published enwik9 row), 1 = with extensions")
    ap.add_argument("--cpu-sample", type=int, default=256 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = production, 1 = serial baseline)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import turbosqueeze_amd as tsq

    world, rank, local_rank = init_distributed("nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.size
    nb = (n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ
    host = tsq.synth.text(n, seed=1 + rank)
    src = torch.from_numpy(host).to(dev)
    del host
    codec = tsq.DeviceCodec(local_rank)
    codec.set_variant(args.variant, args.variant)
    container = torch.empty(tsq.container_bound(n), dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)

    def step():
        codec.compress_async(src, args.ext, container)
        codec.decompress_async(container, nb, back)

    for _ in range(args.warmup):                      # W untimed, unprofiled steps
        step()
    torch.cuda.synchronize()
    if args.warmup:
        _, status = codec.last_size_status()
        assert status == 0, f"device status {status}"
    codec.profile(True)                               # HIP events around the kernels of the timed steps only
    dt = timed_steps(step, args.steps, 0, world, torch.cuda.synchronize, dev)
    enc_ms, enc_n, dec_ms, dec_n = codec.profile_read()
    codec.profile(False)

    # parity of what was timed: exact round trip on the GPU, container size from the frame table
    total_out, status = codec.last_size_status()
    assert status == 0 and total_out == n, (status, total_out)
    assert torch.equal(back, src), "round trip mismatch"
    codec.compress_async(src, args.ext, container)
    torch.cuda.synchronize()
    comp_bytes, status = codec.last_size_status()
    assert status == 0

    ms_per_step = dt / args.steps * 1e3
    value = aggregate_value(world, n, dt, args.steps)

    if rank == 0:
        enc_avg = enc_ms / max(enc_n, 1) * 1e-3
        dec_avg = dec_ms / max(dec_n, 1) * 1e-3
        alg = n + comp_bytes                      # encode: N read + C written; decode: C read + N written
        enc_gbs = alg / enc_avg / 1e9 if enc_avg > 0 else 0.0
        dec_gbs = alg / dec_avg / 1e9 if dec_avg > 0 else 0.0
        dom = ("encode", enc_gbs, enc_avg) if enc_avg >= dec_avg else ("decode", dec_gbs, dec_avg)
        traffic, traffic_src = (None, None)
        if args.variant == 0 and args.ext == 0 and n == 1_000_000_000:
            traffic, traffic_src = pmc_traffic("enc_" if dom[0] == "encode" else "dec_")
        line = {
            "metric": "encode+decode GB/s on enwik9-shaped input (round trip of uncompressed bytes)",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"enwik9-shaped synthetic text, {n} B per GPU ({nb} blocks of 4 MiB), "
                                   f"{'with-extensions' if args.ext else '--no-ext'} level, device-resident, bit-exact round trip",
                       "bytes_per_gpu": n, "blocks_per_gpu": nb, "ext": args.ext, "ratio": round(comp_bytes / n, 5),
                       "sharding": f"{world} independent shard(s), no data-path collective", "kernel_variant": args.variant},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(dom[1], 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom[1] / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(dom[2] * 1e3, 4)},
            "roofline_encode": {"achieved": round(enc_gbs, 3), "frac": round(enc_gbs / HBM_PEAK_GBS, 6), "avg_launch_ms": round(enc_avg * 1e3, 4), "launches": enc_n},
            "roofline_decode": {"achieved": round(dec_gbs, 3), "frac": round(dec_gbs / HBM_PEAK_GBS, 6), "avg_launch_ms": round(dec_avg * 1e3, 4), "launches": dec_n,
                                "read_only_frac": round(comp_bytes / dec_avg / 1e9 / HBM_PEAK_GBS, 6) if dec_avg > 0 else 0.0},
            "encode_GBps": round(n / enc_avg / 1e9, 4) if enc_avg > 0 else 0.0,
            "decode_GBps": round(n / dec_avg / 1e9, 4) if dec_avg > 0 else 0.0,
        }
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample, but never fewer blocks than 2 per host thread (block-parallel CPU code)
            cores = os.cpu_count() or 1
            line["cpu_baseline"] = cpu_baseline(min(n, max(args.cpu_sample, 2 * cores * tsq.BLOCK_SZ)), args.ext)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
