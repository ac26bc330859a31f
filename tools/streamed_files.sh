#!/bin/bash
# File modes of the _MT API on a synthetic file in /dev/shm, contexts warm (tsq_cli bf): (1) streamed -- files larger than
# TSQ_AMD_FILE_INMEM_MAX go through pinned staging in 64-block batches: the device's feeder thread reads batch k+1 (parallel pread)
# while batch k's kernels run and batch k-1 lands in the MAPPED output file; (2) the default for files that fit in memory (read whole,
# memory path, one write at the end); and, for scale, what the box's tmpfs takes on one file.  JSON lines.
# Usage (GPU box, repo root): bash tools/streamed_files.sh [bytes]   (default 4 GiB)
set -u
N=${1:-4294967296}
D=/dev/shm/tsq_streamed_$$
mkdir -p $D
python - <<PY
import sys; sys.path.insert(0, ".")
import turbosqueeze_amd as tsq
tsq.synth.text($N, seed=7).tofile("$D/in.bin")
PY
echo -n '{"what": "streamed (TSQ_AMD_FILE_INMEM_MAX=1)", "result": '; TSQ_AMD_FILE_INMEM_MAX=1 tools/tsq_cli bf $D/in.bin $D --no-ext --reps 3 | tr -d '\n'; echo '}'
echo -n '{"what": "default file mode (whole file in memory)", "result": '; tools/tsq_cli bf $D/in.bin $D --no-ext --reps 3 | tr -d '\n'; echo '}'
python - <<PY
import json, os, time
n = $N
buf = bytearray(64 << 20)
t0 = time.time()
with open("$D/w1", "wb") as f:
    for _ in range(n // len(buf)): f.write(buf)
t1 = time.time(); os.unlink("$D/w1")
fd = os.open("$D/w2", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
t2 = time.time(); os.posix_fallocate(fd, 0, n); t3 = time.time(); os.close(fd); os.unlink("$D/w2")
t4 = time.time()
with open("$D/in.bin", "rb") as f:
    while f.readinto(buf): pass
t5 = time.time()
print(json.dumps({"what": "tmpfs on this box, one file of %d B" % n, "write_one_thread_GBps": round(n / (t1 - t0) / 1e9, 2),
                  "posix_fallocate_GBps": round(n / (t3 - t2) / 1e9, 2), "read_one_thread_GBps": round(n / (t5 - t4) / 1e9, 2)}))
PY
rm -rf $D
