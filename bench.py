#!/usr/bin/env python3
"""bench.py -- encode+decode GB/s of the turbosqueeze hot path on MI355X.

One "step" = one pass of the hot path over ONE enwik9-shaped 10^9-byte job (239 blocks of 4 MiB):

  N = 1   the job is resident in HBM; compress it into a .tsq container (encode kernel + container pack),
          then decompress that container back (frame walk + decode kernel), all on the device.
  N > 1   one process per GPU (`python bench.py --gpus N` typed plainly launches the N ranks itself; under
          torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE and refuses a world that is not N).
          `value` ("scaling": "weak"): EVERY rank runs the N = 1 step on a 10^9-byte job of its own, resident in
          its HBM -- blocks are independent units, there is no data-path collective, nothing crosses PCIe inside
          the timed region -- so `value` = N x 10^9 B / wall time per step (max over ranks) and divides by N = 1's
          like for like.  The same line carries `config4_host_gather` ("strong"), BASELINE.json config 4: ONE
          10^9-byte job, block b -> rank b % N (tsq_threads.cpp:71), every rank encodes the blocks it owns -> one
          RCCL all-gather of the u32 stream sizes -> every rank DMAs its frames to their final place in ONE
          container in host memory (a prefix sum, no gathering rank) -> barrier -> every rank walks the frames,
          brings its own back to HBM and decodes them.  Both PCIe legs and the gather are inside THAT timed
          region, which is why it is never `value`.  With one workgroup per block the encode time of a
          239-block job does not shrink with N -- that section shows exactly that.

value = uncompressed bytes all ranks worked through / wall time per step (GB/s = 1e9 B/s), max over ranks.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events recorded inside the library on
the launch stream; `peak` = 8 TB/s specification, `peak_measured` = a plain device copy kernel on this GPU)
and, at N=1, `cpu_baseline` (the reference's own tsqEncode/tsqDecode compiled into oracle/_ref, or the oracle
port, on the host cores: pinned threads, first touch by the owner, 1 warm + 5 timed passes, median and best,
idealised and reference-shaped pipelines, 1-thread rate and parallel efficiency).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GUIDE_COPY_GBS = 6290.0        # same guide: what a float4 copy kernel measures on an MI355X (79 % of the specification)


def pmc_traffic(kernel_substr: str, blocks: int, fingerprint: str):
    """HBM-side bytes per launch of a kernel from the rocprofv3 PMC summary of this same command
    (tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes), corrected as
    MI355X_MICROARCH.md prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
    The summary is keyed by kernel AND launch shape ("<kernel>@<blocks>", tools/pmc_summary.py): the entry taken is the one of
    the timed job's own grid size -- the profiled command also launches the codec kernels at 1 024 blocks (`throughput`).
    bench.py cannot read PMCs itself (rocprofv3 wraps the process): it takes the newest stored summary ONLY IF that summary was
    made from the kernel sources that are running now (same `kernel_fingerprint`); otherwise the field is null.
    Returns (bytes, source) or (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write.json")))
    if not files:
        return None, "no PMC summary under profiles/"
    try:
        d = json.load(open(files[-1]))
        if d.get("kernel_fingerprint") != fingerprint:
            return None, f"{os.path.relpath(files[-1], ROOT)} was made from other kernel sources ({d.get('kernel_fingerprint')} != {fingerprint})"
        want = f"@{blocks}"

        def pick(table):
            shaped = [v["per_dispatch"] for k, v in table.items() if kernel_substr in k and k.endswith(want)]
            if shaped:
                return shaped[0]
            raise KeyError(f"no entry for a '{kernel_substr}' kernel at {blocks} blocks")
        return int((2 * pick(d["fetch"]) + pick(d["write"])) * 1024), os.path.relpath(files[-1], ROOT) + f" ({kernel_substr}*{want})"
    except Exception as e:
        return None, f"unreadable PMC summary: {e}"


def cpu_baseline(sample_bytes: int, ext: int, reps: int = 5):
    """The CPU path on the host cores over a bounded sample of the same workload (oracle/tsq_oracle.c:
    tsqo_cpubench2).  Block i -> thread i % T as tsq_threads.cpp:71; threads pinned one per allowed CPU; every
    worker first-touches the pages of its own blocks; 1 warm + `reps` timed passes, median and best.
    Runs the reference's own tsqEncode/tsqDecode (oracle/_ref) when that library is present ("reference"),
    else the oracle port ("port").  `value` is the idealised block-parallel round trip (median) at the
    better of T = all hardware threads / half of them; the reference-shaped pipeline (one ordered writer
    thread, tsq_threads.cpp:192-275,604-676) and the 1-thread rate are reported beside it."""
    import ctypes as C

    import turbosqueeze_amd as tsq
    from oracle import pyoracle

    hw = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores, quota_note = hw, "no cgroup CPU quota"
    try:                                               # a container may be allowed fewer CPUs than it can see
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            allowed = max(1, int(int(q) / int(period) + 0.5))
            quota_note = f"cgroup cpu.max allows {allowed} CPUs"
            cores = min(hw, allowed)
    except Exception:
        pass
    host = tsq.synth.text(sample_bytes, seed=1, pad=256)
    orc = pyoracle.Oracle()
    fn = orc.L.tsqo_cpubench2
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    enc = dec = None
    kind = "port"
    if pyoracle.Reference.available():
        ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtsq_ref.so"))
        enc = C.cast(ref.tsqEncode, C.c_void_p)
        dec = C.cast(ref.tsqDecode, C.c_void_p)
        kind = "reference"

    def run(nbytes, threads, shape, n_reps):
        te, td, cb = (C.c_double * n_reps)(), (C.c_double * n_reps)(), C.c_uint64(0)
        bad = fn(enc, dec, host.ctypes.data, nbytes, ext, threads, n_reps, shape, 1, te, td, C.byref(cb))
        te, td = list(te), list(td)
        rt = [a + b for a, b in zip(te, td)]
        g = lambda t: round(nbytes / t / 1e9, 4)
        return {"threads": threads, "ok": bad == 0, "ratio": cb.value / nbytes,
                "encode_GBps": g(statistics.median(te)), "decode_GBps": g(statistics.median(td)), "roundtrip_GBps": g(statistics.median(rt)),
                "encode_best": g(min(te)), "decode_best": g(min(td)), "roundtrip_best": g(min(rt)), "rt_median_s": statistics.median(rt)}

    ideal = None
    # under / at / over the allowance and every hardware thread (a CFS quota throttles long phases, not short bursts): keep the best
    for threads in sorted({cores, max(cores // 2, 1), min(2 * cores, hw), hw}):
        r = run(sample_bytes, threads, 0, reps)
        if ideal is None or r["rt_median_s"] < ideal["rt_median_s"]:
            ideal = r
    shaped = run(sample_bytes, ideal["threads"], 1, max(reps // 2, 2))
    one_bytes = min(sample_bytes, 16 * tsq.BLOCK_SZ)
    one = run(one_bytes, 1, 0, 3)
    eff = lambda k: round(ideal[k] / (one[k] * ideal["threads"]), 4) if one[k] > 0 else None
    strip = lambda r: {k: v for k, v in r.items() if k not in ("rt_median_s",)}
    return {
        # `cores` = the CPUs this process may use (affinity mask and cgroup quota); `threads` = the pthreads of the best run
        "value": ideal["roundtrip_GBps"], "unit": "GB/s", "cores": cores, "threads": ideal["threads"], "kind": kind,
        "sample": f"{sample_bytes} B of the same enwik9-shaped text; idealised block-parallel pthreads (block i -> thread i % T), "
                  f"threads pinned, pages first-touched by their owner, 1 warm + {reps} timed passes, value = median round trip; "
                  f"host shows {hw} hardware threads, {quota_note}; roundtrip_ok={ideal['ok'] and shaped['ok'] and one['ok']} ratio={ideal['ratio']:.4f}",
        "median": ideal["roundtrip_GBps"], "best": ideal["roundtrip_best"],
        "encode_GBps": ideal["encode_GBps"], "decode_GBps": ideal["decode_GBps"],
        "encode_best": ideal["encode_best"], "decode_best": ideal["decode_best"],
        "per_thread_GBps": {"encode": round(ideal["encode_GBps"] / ideal["threads"], 4), "decode": round(ideal["decode_GBps"] / ideal["threads"], 4)},
        "one_thread": {"sample_bytes": one_bytes, "encode_GBps": one["encode_GBps"], "decode_GBps": one["decode_GBps"]},
        "efficiency": {"encode": eff("encode_GBps"), "decode": eff("decode_GBps")},
        "reference_shaped": strip(shaped),
    }


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` typed plainly (no torch.distributed.run around it, WORLD_SIZE unset) with N > 1: this process is
    only the launcher.  It refuses when fewer than N GPUs are visible (unless TSQ_BENCH_SHARE_GPU puts every rank on GPU 0: the
    dry run of a 1-GPU box), then runs the same command line under `python -m torch.distributed.run`, one rank per GPU, rendezvous
    on 127.0.0.1, and returns that run's exit code.  The ranks print the JSON line (rank 0); nothing is printed here."""
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    if not os.environ.get("TSQ_BENCH_SHARE_GPU") and visible < n_ranks:
        print(f"bench.py: --gpus {n_ranks} needs {n_ranks} visible GPUs, this node shows {visible} "
              f"(TSQ_BENCH_SHARE_GPU=1 TSQ_BENCH_BACKEND=gloo runs every rank on GPU 0 as a dry run)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    port = env.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def init_distributed(backend: str):
    """(world, rank, local_rank); one process per GPU, rendezvous on 127.0.0.1."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("TSQ_BENCH_FORCE_SHARDED"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, local_rank


def timed_steps(step, steps: int, warmup: int, world: int, device_sync, reduce_device=None):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds (the contract of the driver)."""
    import torch
    import torch.distributed as dist

    def barrier():
        if dist.is_available() and dist.is_initialized():
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
    if dist.is_available() and dist.is_initialized():
        tt = torch.tensor([dt], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def aggregate_value(job_bytes: int, dt: float, steps: int):
    """Whole-job GB/s: bytes of the job(s) one step works through / wall time per step."""
    return job_bytes / (dt / steps) / 1e9


def roofline_entries(n, comp_bytes, enc_ms, enc_n, dec_ms, dec_n, peak_measured):
    enc_avg = enc_ms / max(enc_n, 1) * 1e-3
    dec_avg = dec_ms / max(dec_n, 1) * 1e-3
    alg = n + comp_bytes                          # encode: N read + C written; decode: C read + N written
    enc_gbs = alg / enc_avg / 1e9 if enc_avg > 0 else 0.0
    dec_gbs = alg / dec_avg / 1e9 if dec_avg > 0 else 0.0
    pm = peak_measured or 0.0

    def entry(gbs, avg, launches, extra=None):
        e = {"achieved": round(gbs, 3), "frac": round(gbs / HBM_PEAK_GBS, 6),
             "frac_of_measured": round(gbs / pm, 6) if pm else None, "avg_launch_ms": round(avg * 1e3, 4), "launches": launches}
        if extra:
            e.update(extra)
        return e
    enc = entry(enc_gbs, enc_avg, enc_n)
    dec = entry(dec_gbs, dec_avg, dec_n, {"read_only_frac": round(comp_bytes / dec_avg / 1e9 / HBM_PEAK_GBS, 6) if dec_avg > 0 else 0.0})
    dom = ("encode", enc_gbs, enc_avg) if enc_avg >= dec_avg else ("decode", dec_gbs, dec_avg)
    return alg, enc, dec, dom, enc_avg, dec_avg


def host_gather_step(tsq, sharding, codec, dev, n, ext, steps, check_oracle):
    """The sharded step of bench.py --gpus N with a world of ONE rank and no process group: encode the owned blocks -> every frame by
    DMA to its place in ONE container in host memory (/dev/shm, hipHostRegister'ed) -> walk the frames, bring them back, decode."""
    import torch
    lay = sharding.ShardLayout(n, 0, 1)
    host = tsq.synth.text(n, seed=1)
    d_shard = torch.from_numpy(lay.pack_input(host)).to(dev)
    expect = torch.from_numpy(lay.expected_output(host)).to(dev)
    hc = sharding.HostContainer("tsq_bench_gather_%d" % os.getpid(), tsq.container_bound(n), create=True)
    hc.register()
    try:
        sc = sharding.ShardedCodec(lay, sharding.DeviceBlocks(codec), hc, ext)
        d_back = torch.empty(max(lay.shard_bytes, 1), dtype=torch.uint8, device=dev)
        size = [0]

        def step():
            size[0] = sc.compress(d_shard)
            sc.decompress(size[0], d_back)
        step()
        for k in sc.seconds:
            sc.seconds[k] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.equal(d_back[:lay.shard_bytes], expect), "host-gather round trip mismatch"
        equal = None
        if check_oracle:
            from oracle import pyoracle
            want = pyoracle.Oracle().compress(host, ext, threads=min(32, os.cpu_count() or 1))
            equal = bool(len(want) == size[0] and bytes(hc.array[:size[0]]) == want)
            assert equal, "the host-gathered container differs from the oracle's"
        out = {"what": "the N > 1 step at a world of one rank: encode -> frames DMA'd to one container in host memory -> frame walk, frames back, decode "
                       "(both PCIe legs inside the step; no process group, so no all-gather and no barriers)",
               "value": round(aggregate_value(n, dt, steps), 4), "unit": "GB/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
               "step_breakdown_ms": {k: round(v / steps * 1e3, 3) for k, v in sc.seconds.items()},
               "container_bytes": size[0], "container_equals_oracle": equal}
    finally:
        hc.close()
    del d_shard, expect
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1_000_000_000, help="uncompressed bytes of the job (enwik9 = 1e9)")
    ap.add_argument("--ext", type=int, default=0, help="0 = --no-ext fast level (the published enwik9 row), 1 = with extensions")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="N = 1: skip the comparison of the timed job's container with the CPU oracle (after the timed region)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the oracle comparison of rank 0's timed job (the job itself is always timed: it is `value`)")
    ap.add_argument("--no-throughput", action="store_true", help="N = 1: skip the extra chip-filling measurement (4 GiB of the same text = 1 024 blocks)")
    ap.add_argument("--throughput-size", type=int, default=4 << 30)
    ap.add_argument("--kind", default="text", choices=["text", "zeros", "random", "mix"],
                    help="the job's synthetic input: enwik9-shaped text (the headline), or one of BASELINE config 5's inputs")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the extra sections (the other level, config 5's inputs, the host-gather step)")
    ap.add_argument("--config5-size", type=int, default=1 << 30)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = production, 1 = serial baselines, 5 = the previous round's production encoder, frozen in the A/B library)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))               # `--gpus N` means N ranks, whoever starts the script

    import numpy as np
    import torch
    import torch.distributed as dist

    import turbosqueeze_amd as tsq
    from turbosqueeze_amd import sharding

    # (TSQ_BENCH_BACKEND=gloo TSQ_BENCH_SHARE_GPU=1: dry run of the N > 1 path with every rank on GPU 0 -- for a 1-GPU box;
    #  TSQ_BENCH_FORCE_SHARDED=1: the N > 1 path with a world of ONE rank -- process group, RCCL all-gather, host container,
    #  hipHostRegister, both sharded C calls -- which is how tests/test_gpu_parity.py executes the RCCL branch on a 1-GPU box)
    force_sharded = bool(os.environ.get("TSQ_BENCH_FORCE_SHARDED"))
    backend = os.environ.get("TSQ_BENCH_BACKEND", "nccl")
    world, rank, local_rank = init_distributed(backend)
    if world != args.gpus:
        # a launcher that started another number of ranks than --gpus says: refuse rather than print a line whose n_gpus is not
        # what was asked for
        print(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE')})", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("TSQ_BENCH_SHARE_GPU"):
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU of its own (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")       # where the small collectives' tensors live

    def device_identity():
        p = torch.cuda.get_device_properties(local_rank)
        return {"rank": rank, "device": local_rank, "uuid": str(getattr(p, "uuid", "")), "name": p.name,
                "arch": getattr(p, "gcnArchName", ""), "cus": p.multi_processor_count}

    n = args.size
    nb = (n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ

    def make_input(m, seed):
        if args.kind == "zeros":
            return np.zeros(m, dtype=np.uint8)
        if args.kind == "random":
            return tsq.synth.random_bytes(m, seed)
        if args.kind == "mix":
            return tsq.synth.mix(m, seed)
        return tsq.synth.text(m, seed=seed)
    kind_name = {"text": "enwik9-shaped synthetic text", "zeros": "zeros", "random": "random bytes", "mix": "50 % mix (64 KiB chunks random / text)"}[args.kind]
    codec = tsq.DeviceCodec(local_rank, ab=args.variant == 5)
    codec.set_variant(args.variant, args.variant if args.variant in (0, 1) else 0)

    def single_gpu_job(seed, steps, warmup, check_oracle=False, n=n, nb=nb, ext=None, make=None):
        """The N=1 step on this rank's own job.  -> (dt, comp_bytes, enc/dec kernel ms+launches, call ms, container == oracle's)"""
        ext = args.ext if ext is None else ext
        host = make(n, seed) if make else make_input(n, seed)
        src = torch.from_numpy(host).to(dev)
        container = torch.empty(tsq.container_bound(n), dtype=torch.uint8, device=dev)
        back = torch.empty(n, dtype=torch.uint8, device=dev)

        def step():
            codec.compress_async(src, ext, container)
            codec.decompress_async(container, nb, back)

        for _ in range(warmup):                           # W untimed, unprofiled steps
            step()
        torch.cuda.synchronize()
        if warmup:
            _, status = codec.last_size_status()
            assert status == 0, f"device status {status}"
        codec.profile(True)                               # HIP events around the kernels of the timed steps only
        dt = timed_steps(step, steps, 0, world, torch.cuda.synchronize, red_dev)
        kern = codec.profile_read()
        calls = codec.profile_read_calls()
        codec.profile(False)
        # parity of what was timed: exact round trip on the GPU, container size from the frame table
        total_out, status = codec.last_size_status()
        assert status == 0 and total_out == n, (status, total_out)
        assert torch.equal(back, src), "round trip mismatch"
        codec.compress_async(src, ext, container)
        torch.cuda.synchronize()
        comp_bytes, status = codec.last_size_status()
        assert status == 0
        oracle_equal = None
        if check_oracle:
            # ... and, outside the timed region, the timed job's container byte for byte against the CPU oracle (the checker)
            from oracle import pyoracle
            want = pyoracle.Oracle().compress(host, ext, threads=min(32, os.cpu_count() or 1))
            got = container[:comp_bytes].cpu().numpy()
            oracle_equal = bool(len(want) == comp_bytes and got.tobytes() == want)
            assert oracle_equal, "the timed job's container differs from the oracle's"
        del src, container, back, host
        torch.cuda.empty_cache()
        return dt, comp_bytes, kern, calls, oracle_equal

    line = None
    if world == 1 and not force_sharded:
        dt, comp_bytes, (enc_ms, enc_n, dec_ms, dec_n), (cmp_ms, cmp_n, dcm_ms, dcm_n), oracle_equal = single_gpu_job(1, args.steps, args.warmup, check_oracle=not args.no_oracle_check)
        peak_best, peak_med = codec.measure_copy(4 << 30, 7)
        probe_shape = codec.copy_probe_shape()
        alg, enc_e, dec_e, dom, enc_avg, dec_avg = roofline_entries(n, comp_bytes, enc_ms, enc_n, dec_ms, dec_n, peak_best)
        traffic, traffic_src = (None, "PMC summaries are collected for the default job only")
        fingerprint = tsq.source_fingerprint()
        if args.variant == 0 and args.ext == 0 and n == 1_000_000_000:
            traffic, traffic_src = pmc_traffic("enc_" if dom[0] == "encode" else "dec_", nb, fingerprint)
        cmp_avg = cmp_ms / max(cmp_n, 1) * 1e-3
        dcm_avg = dcm_ms / max(dcm_n, 1) * 1e-3
        line = {
            "metric": "encode+decode GB/s on enwik9-shaped input (round trip of uncompressed bytes)",
            "value": round(aggregate_value(n, dt, args.steps), 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "devices": [device_identity()],
            "config": {"workload": f"{kind_name}, one {n} B job ({nb} blocks of 4 MiB), "
                                   f"{'with-extensions' if args.ext else '--no-ext'} level, device-resident, bit-exact round trip",
                       "job_bytes": n, "blocks": nb, "ext": args.ext, "ratio": round(comp_bytes / n, 5),
                       "sharding": "1 GPU owns every block", "kernel_variant": args.variant,
                       "container_equals_oracle": oracle_equal, "kernel_fingerprint": fingerprint},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(dom[1], 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom[1] / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(dom[2] * 1e3, 4),
                         # the north star's decode criterion: compressed bytes READ per second by the decode kernel / HBM peak
                         "decode_read_only_frac": dec_e["read_only_frac"],
                         "peak_measured": round(peak_best, 1), "peak_measured_median": round(peak_med, 1),
                         "frac_of_measured": round(dom[1] / peak_best, 6) if peak_best else None,
                         "peak_measured_by": "copy_probe_kernel over 4 GiB, bytes read + written, best of 7; shape chosen by the probe: " + probe_shape,
                         # the guide's own figure for the same kind of kernel (MI355X_MICROARCH.md: float4 copy, 6.29 TB/s = 79 % of the
                         # specification): this box's probe reaches a few per cent less, so the fraction against it is the stricter one
                         "peak_guide_copy": GUIDE_COPY_GBS, "frac_of_guide_copy": round(dom[1] / GUIDE_COPY_GBS, 6)},
            "roofline_encode": enc_e, "roofline_decode": dec_e,
            # uncompressed bytes / time of the whole call: encode kernel + container pack; frame walk + decode kernel
            "encode_GBps": round(n / cmp_avg / 1e9, 4) if cmp_avg > 0 else 0.0,
            "decode_GBps": round(n / dcm_avg / 1e9, 4) if dcm_avg > 0 else 0.0,
            "encode_kernel_GBps": round(n / enc_avg / 1e9, 4) if enc_avg > 0 else 0.0,
            "decode_kernel_GBps": round(n / dec_avg / 1e9, 4) if dec_avg > 0 else 0.0,
        }
        if not args.no_throughput:
            # The headline job has fewer blocks (239) than the GPU has CUs (256): with one workgroup per block its kernel times are
            # per-block latencies.  The chip-filling figure beside it (device-resident kernel throughput): 4 GiB of the
            # same text = 1 024 blocks, same level, same HIP events, the timed container compared with the oracle's afterwards.
            tn = args.throughput_size
            tnb = (tn + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ
            tsteps = 3
            tdt, tcomp, (te_ms, te_n, td_ms, td_n), _, t_equal = single_gpu_job(1, tsteps, 1, check_oracle=not args.no_oracle_check, n=tn, nb=tnb)
            t_alg, te_e, td_e, _, te_avg, td_avg = roofline_entries(tn, tcomp, te_ms, te_n, td_ms, td_n, peak_best)
            line["throughput"] = {
                "workload": f"{tn} B of the same enwik9-shaped text ({tnb} blocks of 4 MiB: more blocks than CUs), "
                            f"{'with-extensions' if args.ext else '--no-ext'} level, device-resident, bit-exact round trip",
                "job_bytes": tn, "blocks": tnb, "steps": tsteps, "ratio": round(tcomp / tn, 5), "container_equals_oracle": t_equal,
                "value": round(aggregate_value(tn, tdt, tsteps), 4), "unit": "GB/s",
                "encode_kernel_GBps": round(tn / te_avg / 1e9, 3) if te_avg > 0 else 0.0,
                "decode_kernel_GBps": round(tn / td_avg / 1e9, 3) if td_avg > 0 else 0.0,
                "encode_kernel_ms": round(te_avg * 1e3, 3), "decode_kernel_ms": round(td_avg * 1e3, 3),
                "algorithmic_bytes_per_launch": t_alg,
                "roofline_encode": te_e, "roofline_decode": td_e,     # (N + C) / t against 8 TB/s and against the measured copy
            }
        if not args.no_extras:
            # ---- what a user of the reference sees beyond the published --no-ext row, same HIP events, every
            #      timed container compared with the oracle's outside the timed region.  Short runs (1 warm + 2 timed steps).
            def kernel_line(job_n, dtj, comp, kern, equal, steps_):
                e_ms, e_n, d_ms, d_n = kern
                e_avg, d_avg = e_ms / max(e_n, 1) * 1e-3, d_ms / max(d_n, 1) * 1e-3
                return {"job_bytes": job_n, "blocks": (job_n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ, "ratio": round(comp / job_n, 5),
                        "value": round(aggregate_value(job_n, dtj, steps_), 4), "unit": "GB/s",
                        "encode_kernel_ms": round(e_avg * 1e3, 3), "decode_kernel_ms": round(d_avg * 1e3, 3),
                        "encode_kernel_GBps": round(job_n / e_avg / 1e9, 3) if e_avg > 0 else 0.0,
                        "decode_kernel_GBps": round(job_n / d_avg / 1e9, 3) if d_avg > 0 else 0.0,
                        "roofline_frac_encode": round((job_n + comp) / e_avg / 1e9 / HBM_PEAK_GBS, 6) if e_avg > 0 else 0.0,
                        "roofline_frac_decode": round((job_n + comp) / d_avg / 1e9 / HBM_PEAK_GBS, 6) if d_avg > 0 else 0.0,
                        "container_equals_oracle": equal}
            xsteps = 2
            # (a) the reference CLI's default level: extensions ON (sample/main.cpp:130,144), the same 10^9 B of text
            xdt, xcomp, xkern, _, xeq = single_gpu_job(1, xsteps, 1, check_oracle=not args.no_oracle_check, ext=1 - args.ext)
            key = "ext1" if args.ext == 0 else "ext0"
            line[key] = dict(kernel_line(n, xdt, xcomp, xkern, xeq, xsteps),
                             workload=f"the same {n} B of enwik9-shaped text, {'with-extensions' if args.ext == 0 else '--no-ext'} level "
                                      f"({'the reference CLI default, sample/main.cpp:130' if args.ext == 0 else 'the published row'})")
            # (b) BASELINE config 5's inputs at one GPU's scale, with extensions: zeros / random / 50 % mix
            c5n = args.config5_size
            makers = {"zeros": lambda m, sd: np.zeros(m, dtype=np.uint8), "random": lambda m, sd: tsq.synth.random_bytes(m, 3),
                      "mix": lambda m, sd: tsq.synth.mix(m, 3)}
            c5 = {"workload": f"{c5n} B per input ({(c5n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ} blocks), with-extensions level, device-resident, bit-exact round trip"}
            for kind, mk in makers.items():
                kdt, kcomp, kkern, _, keq = single_gpu_job(3, xsteps, 1, check_oracle=not args.no_oracle_check, n=c5n,
                                                           nb=(c5n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ, ext=1, make=mk)
                c5[kind] = kernel_line(c5n, kdt, kcomp, kkern, keq, xsteps)
            line["config5"] = c5
            # (c) the N > 1 step's shape at a world of one rank: frames placed in ONE container in host memory, barrier, owned frames
            #     back and decoded -- both PCIe legs and the host gather inside the step, so that the first point of the 1 -> 8 curve
            #     can be read against points that carry them (bench.py --gpus N runs exactly this per rank)
            line["host_gather"] = host_gather_step(tsq, sharding, codec, dev, n, args.ext, args.steps, not args.no_oracle_check)
        if not args.no_cpu_baseline:
            # bounded sample, but never fewer blocks than 2 per host thread (block-parallel CPU code)
            line["cpu_baseline"] = cb = cpu_baseline(min(n, args.cpu_sample), args.ext)
            # the north star's criterion as a number: GPU decode over the reference's multithreaded CPU decode on this box's host cores
            dk = line["decode_kernel_GBps"]
            tk = line.get("throughput", {}).get("decode_kernel_GBps")
            shaped_d, ideal_d = cb["reference_shaped"]["decode_GBps"], cb["decode_GBps"]
            # Efficiency per execution unit: one CU works on one 4 MiB block at a time (two in the encoder's lean
            # layout), one host thread on one block -- GB/s of a CU over GB/s of a host core, for both kernels.  Below 1 means a CU is
            # slower than a core: the GPU's lead over the CPU is its number of CUs, not the speed of each.
            cus = line["devices"][0]["cus"]
            one = cb["one_thread"]
            per_cu = {"encode": line["encode_kernel_GBps"] / min(nb, cus), "decode": line["decode_kernel_GBps"] / min(nb, cus)}
            line["per_cu_vs_host_core"] = {
                "what": "kernel GB/s per busy CU (headline job: one block per CU) over the CPU code's GB/s on ONE host thread (cpu_baseline.one_thread)",
                "busy_cus": min(nb, cus), "gpu_per_cu_GBps": {k: round(v, 4) for k, v in per_cu.items()},
                "host_one_thread_GBps": {"encode": one["encode_GBps"], "decode": one["decode_GBps"]},
                "encode": round(per_cu["encode"] / one["encode_GBps"], 4) if one["encode_GBps"] else None,
                "decode": round(per_cu["decode"] / one["decode_GBps"], 4) if one["decode_GBps"] else None,
            }
            tp = line.get("throughput")
            if tp:
                line["per_cu_vs_host_core"]["chip_filled_1024_blocks"] = {
                    "encode": round(tp["encode_kernel_GBps"] / cus / one["encode_GBps"], 4) if one["encode_GBps"] else None,
                    "decode": round(tp["decode_kernel_GBps"] / cus / one["decode_GBps"], 4) if one["decode_GBps"] else None}
            line["decode_vs_cpu_mt"] = {
                "what": "device-resident decode GB/s of ONE MI355X over the CPU decode GB/s on this box's host cores (reference-shaped pipeline: "
                        "tsq_threads.cpp's reader / workers / ordered writer; idealised: block-parallel threads, no writer); the north star asks >= 10x at 8 GPUs",
                "gpu_decode_call_GBps": line["decode_GBps"], "gpu_decode_kernel_GBps": dk, "gpu_decode_kernel_GBps_1024_blocks": tk,
                "cpu_reference_shaped_GBps": shaped_d, "cpu_idealised_GBps": ideal_d, "cpu_cores": cb["cores"], "cpu_threads": cb["threads"],
                "x_reference_shaped": round(line["decode_GBps"] / shaped_d, 2) if shaped_d else None,
                "x_idealised": round(line["decode_GBps"] / ideal_d, 2) if ideal_d else None,
                "x_reference_shaped_kernel_1024_blocks": round(tk / shaped_d, 2) if tk and shaped_d else None,
                "x_idealised_kernel_1024_blocks": round(tk / ideal_d, 2) if tk and ideal_d else None,
            }
    else:
        # ---- N > 1.  Two measurements in one line:
        # (1) `value` ("scaling": "weak"): every rank runs the N = 1 step -- one job of `n` bytes resident in ITS HBM, compressed to a
        #     container in HBM and decompressed again -- so the work per GPU is fixed as N grows, nothing crosses PCIe inside the timed
        #     region (a PCIe-inclusive rate is never `value`), the blocks of the N jobs are independent units with no data-path
        #     collective, and the figure divides by N = 1's like for like.  K steps, W warm-ups, barrier + device sync on both sides,
        #     MAX over ranks: the contract's timing exactly.
        # (2) `config4_host_gather` ("strong"): BASELINE config 4 -- ONE job of `n` bytes, block b owned by rank b % N
        #     (tsq_threads.cpp:71), one all-gather of the u32 sizes (RCCL), every frame by DMA to its place in ONE container in host
        #     memory, barrier, owned frames back and decoded.  Both PCIe legs and the gather are inside ITS timed region.
        group_world = dist.get_world_size()
        idents = [None] * group_world
        dist.all_gather_object(idents, device_identity())
        distinct = len({d["uuid"] or (d["rank"], d["device"]) for d in idents})

        # (2) first: it leaves nothing behind on the device
        gsteps, gwarm = min(args.steps, 3), min(args.warmup, 1)
        lay = sharding.ShardLayout(n, rank, world)
        host = make_input(n, 1)                           # every rank generates the same job and keeps its blocks
        d_shard = torch.from_numpy(lay.pack_input(host)).to(dev)
        expect = torch.from_numpy(lay.expected_output(host)).to(dev)
        del host
        name = "tsq_bench_%s" % os.environ.get("MASTER_PORT", "0")
        cap = tsq.container_bound(n)
        hc = sharding.HostContainer(name, cap, create=True) if rank == 0 else None
        dist.barrier()
        if rank != 0:
            hc = sharding.HostContainer(name, cap, create=False)
        hc.register()
        sc = sharding.ShardedCodec(lay, sharding.DeviceBlocks(codec, collective_device=red_dev), hc, args.ext)
        sc.always_gather = force_sharded                  # (a world of one still runs the all-gather)
        d_back = torch.empty(max(lay.shard_bytes, 1), dtype=torch.uint8, device=dev)
        size_seen = [0]

        barrier_s = [0.0, 0.0]

        def step():
            size_seen[0] = sc.compress(d_shard)
            t0 = time.perf_counter()
            dist.barrier()                                # every rank's frames are in the host container
            t1 = time.perf_counter()
            sc.decompress(size_seen[0], d_back)
            t2 = time.perf_counter()
            dist.barrier()                                # nobody overwrites the container while another still reads it
            barrier_s[0] += t1 - t0
            barrier_s[1] += time.perf_counter() - t2

        for _ in range(gwarm):
            step()
        codec.profile(True)
        for k in sc.seconds:
            sc.seconds[k] = 0.0
        barrier_s[0] = barrier_s[1] = 0.0
        gdt = timed_steps(step, gsteps, 0, world, torch.cuda.synchronize, red_dev)
        g_enc_ms, g_enc_n, g_dec_ms, g_dec_n = codec.profile_read()
        codec.profile(False)
        # where this rank's step went (wall ms per step; rank 0's view): encode = launch + wait for the kernel, size_gather = the
        # all-gather of u32 sizes, place_d2h = frames to the host container, fetch_h2d_decode = frame walk + frames back + decode
        breakdown = {k: round(v / gsteps * 1e3, 3) for k, v in sc.seconds.items()}
        breakdown["barrier_after_place"] = round(barrier_s[0] / gsteps * 1e3, 3)
        breakdown["barrier_after_decode"] = round(barrier_s[1] / gsteps * 1e3, 3)
        assert torch.equal(d_back[:lay.shard_bytes], expect), "round trip mismatch on rank %d" % rank
        g_comp = size_seen[0]
        # the host-gathered container is a well-formed .tsq file of the whole job
        g_equal = None
        if rank == 0:
            total, frame_at, sizes, ext_bits, out_len = sharding.walk_frames(hc.array, g_comp)
            assert total == n and len(sizes) == nb and int(frame_at[-1]) + 3 + int(sizes[-1]) == g_comp
            if not args.no_oracle_check:
                # ... and, outside the timed region, byte for byte the CPU oracle's container of the same job (the checker)
                from oracle import pyoracle
                want = pyoracle.Oracle().compress(make_input(n, 1), args.ext, threads=min(32, os.cpu_count() or 1))
                g_equal = bool(len(want) == g_comp and bytes(hc.array[:g_comp]) == want)
                assert g_equal, "the host-gathered container differs from the oracle's"
        # the slowest rank's kernel times
        gkt = torch.tensor([g_enc_ms / max(g_enc_n, 1), g_dec_ms / max(g_dec_n, 1)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(gkt, op=dist.ReduceOp.MAX)
        dist.barrier()
        hc.close()
        del d_shard, d_back, expect, sc
        torch.cuda.empty_cache()

        # (1) the timed job of the line
        dt, comp_bytes, (enc_ms, enc_n, dec_ms, dec_n), _, oracle_equal = single_gpu_job(
            1 + rank, args.steps, args.warmup, check_oracle=(rank == 0 and not args.no_oracle_check and not args.no_weak))
        kt = torch.tensor([enc_ms / max(enc_n, 1), dec_ms / max(dec_n, 1), float(comp_bytes)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        if rank == 0:
            enc_avg, dec_avg = float(kt[0]) * 1e-3, float(kt[1]) * 1e-3
            alg = n + comp_bytes
            dom_name, dom_avg = ("encode", enc_avg) if enc_avg >= dec_avg else ("decode", dec_avg)
            dom_gbs = alg / dom_avg / 1e9 if dom_avg > 0 else 0.0
            line = {
                "metric": "encode+decode GB/s on enwik9-shaped input (round trip of uncompressed bytes)",
                "value": round(aggregate_value(world * n, dt, args.steps), 4), "unit": "GB/s", "n_gpus": group_world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8", "data": "synthetic",
                "process_group_world_size": group_world, "distinct_devices": distinct, "devices": idents,
                "config": {"workload": f"{kind_name}, {world} jobs of {n} B ({nb} blocks of 4 MiB each), one per GPU and resident in its HBM, "
                                       f"{'with-extensions' if args.ext else '--no-ext'} level, the N = 1 step on every rank, bit-exact round trip",
                           "job_bytes": n, "jobs": world, "blocks": nb, "ext": args.ext, "ratio": round(comp_bytes / n, 5),
                           "sharding": "independent blocks: every GPU owns the blocks of its job; no data-path collective",
                           "kernel_variant": args.variant, "container_equals_oracle": oracle_equal, "collective_backend": backend,
                           "kernel_fingerprint": tsq.source_fingerprint()},
                # per GPU (the slowest rank's kernel averages): the same quantity as the N = 1 line's roofline
                "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(dom_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(dom_gbs / HBM_PEAK_GBS, 6), "traffic": None, "traffic_source": "PMC summaries are collected at N = 1",
                             "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(dom_avg * 1e3, 4), "per": "GPU (slowest rank)"},
                "slowest_rank_kernel_ms": {"encode": round(enc_avg * 1e3, 4), "decode": round(dec_avg * 1e3, 4)},
                "config4_host_gather": {
                    "what": f"BASELINE config 4: ONE {n} B job ({nb} blocks) block-sharded over {world} GPUs (block b -> rank b % {world}), one all-gather of "
                            f"the u32 sizes per step, frames DMA'd to one container in host memory, barrier, owned frames back and decoded; both PCIe "
                            f"legs inside the step, so this is never `value`",
                    "scaling": "strong", "value": round(aggregate_value(n, gdt, gsteps), 4), "unit": "GB/s", "steps": gsteps, "warmup": gwarm,
                    "ms_per_step": round(gdt / gsteps * 1e3, 3), "ratio": round(g_comp / n, 5), "container_equals_oracle": g_equal,
                    "collective_backend": backend,
                    "slowest_rank_kernel_ms": {"encode": round(float(gkt[0]), 4), "decode": round(float(gkt[1]), 4)},
                    "rank0_step_breakdown_ms": breakdown,
                    "note": "one workgroup per 4 MiB block: the encode time of a job of at most 256 blocks per GPU is the per-block latency at any N, "
                            "so this curve is flat in encode by construction and carries two PCIe legs N = 1's device-resident step does not "
                            "(DESIGN.md section 6); what shrinks with N is decode and the legs",
                },
            }
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
