// tsq_compat.hip -- the reference's public API (turbosqueeze.h:441-674) on top of the device
// path, and the HIP stream scheduler that replaces the reference's reader/worker/writer thread
// pool (tsq_threads.cpp).  Host code only; the data path is H2D copy -> kernels -> D2H copy.
//
// Scheduler shape.  A job is cut into batches of consecutive 4 MiB blocks.  Batch k runs on
// pipeline lane k % L; a lane owns one tsqa_ctx (HIP stream + HBM scratch), pinned host staging
// buffers and device staging buffers.  Lanes are spread round-robin over the devices listed in
// TSQ_AMD_DEVICES (default: the current device), so with 8 devices consecutive batches go to
// consecutive GPUs and the host gathers the compressed pieces in block order -- the analogue
// of block i -> worker i % num_cores (tsq_threads.cpp:71,463) and of the ordered writer
// (tsq_threads.cpp:192-275, 604-676).  Three kinds of threads per context, as in the reference (reader / workers / writer,
// tsq_threads.cpp:52-275): one FEEDER per listed device issues that device's batches -- file read into pinned staging (parallel
// pread), copy to the device, kernel launches -- so that the devices' feeds do not queue behind each other and a file read overlaps
// the previous batch's kernels and copy back; the scheduler thread cuts the job and, for compression, drains the oldest lane
// (D2H + append + progress callbacks); decompression has its own drainer thread.
#include "tsq_internal.h"
#include "tsq_common.cuh"

#include "../../include/turbosqueeze.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using tsq::FrameInfo;
using tsq::kBlockSize;
using tsq::kSlotSize;

namespace {

constexpr size_t kHalo = 128;   // look-ahead the encoder may read past a block

std::vector<int> device_list()
{
    std::vector<int> devs;
    if (const char* env = getenv("TSQ_AMD_DEVICES")) {
        const char* p = env;
        while (*p) {
            char* end = nullptr;
            long v = strtol(p, &end, 10);
            if (end == p) break;
            devs.push_back((int)v);
            p = (*end == ',') ? end + 1 : end;
        }
    }
    if (devs.empty()) {
        int d = 0;
        if (const char* env = getenv("TSQ_AMD_DEVICE")) d = atoi(env);
        else if (hipGetDevice(&d) != hipSuccess) d = 0;
        devs.push_back(d);
    }
    return devs;
}

size_t env_size(const char* name, size_t dflt)
{
    const char* e = getenv(name);
    if (!e) return dflt;
    long v = atol(e);
    return v > 0 ? (size_t)v : dflt;
}

// TSQ_AMD_DEBUG=1: wall-clock marks of the scheduler's steps on stderr
struct Marks {
    const bool on = getenv("TSQ_AMD_DEBUG") != nullptr;
    double t0 = now();
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void at(const char* what, int feeder = -1) {
        if (!on) return;
        const double t = now();
        if (feeder < 0) fprintf(stderr, "turbosqueeze_amd: %8.2f ms  %s\n", (t - t0) * 1e3, what);
        else fprintf(stderr, "turbosqueeze_amd: %8.2f ms  [feeder %d] %s\n", (t - t0) * 1e3, feeder, what);
    }
};

// Positional file I/O in a few threads: one thread moves 2-3 GB/s in and out of the page cache, a batch is hundreds of megabytes.
size_t io_threads_for(size_t n)
{
    if (n < (size_t(8) << 20)) return 1;
    unsigned hw = std::thread::hardware_concurrency();
    size_t k = hw >= 8 ? 2 : 1;            // (measured on the 16-CPU-quota box: 2 threads 0.72 / 0.85 s per 4 GiB, 4: 0.83 / 0.97, 8: 0.95 / 1.36 -- the quota throttles)
    if (const char* e = getenv("TSQ_AMD_IO_THREADS")) { long v = atol(e); if (v >= 1 && v <= 64) k = (size_t)v; }
    return k;
}
template <class F>   // F(fd, ptr, len, offset) -> ssize_t (pread / pwrite)
bool io_par(F op, int fd, uint8_t* p, size_t n, size_t at)
{
    auto all = [&](uint8_t* q, size_t len, size_t off) {
        size_t done = 0;
        while (done < len) { ssize_t r = op(fd, q + done, len - done, (off_t)(off + done)); if (r <= 0) return false; done += (size_t)r; }
        return true;
    };
    const size_t k = io_threads_for(n);
    if (k <= 1) return all(p, n, at);
    const size_t per = ((n / k) + 4095) & ~size_t(4095);
    std::atomic<bool> good{true};
    std::vector<std::thread> th;
    for (size_t i = 0; i < k; ++i) {
        const size_t a = per * i;
        if (a >= n) break;
        const size_t len = n - a < per ? n - a : per;
        th.emplace_back([&, a, len] { if (!all(p + a, len, at + a)) good = false; });
    }
    for (auto& t : th) t.join();
    return good;
}

void complain_once(const char* what)
{
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
        fprintf(stderr, "turbosqueeze_amd: %s -- there is no CPU fallback, the call fails\n", what);
}

// A fresh malloc() of a gigabyte is a quarter of a million untouched pages; letting the device-to-host copy fault them
// in one by one costs more than the copy.  A few threads touch them while the kernels run.
struct Prefault {
    std::vector<std::thread> th;
    std::atomic<bool> stop{false};
    // A mapped output file: its pages are provided by fallocate AHEAD of the batches that land in it (one call allocates at several
    // times the rate page faults or write() do on one file: tools/streamed_files.sh).  The sink owns the allocation (Sink::ensure_allocated:
    // Linux fallocate, never glibc's read-modify-write emulation, and nothing is ever copied into a range that was not allocated);
    // this thread only keeps it a bounded distance in front of the write cursor, so that a lying container header cannot make it
    // allocate more than the job ever writes.
    void start_mapped(std::function<bool(size_t)> ensure, std::function<size_t()> cursor, size_t n) {
        if (n < (size_t(64) << 20)) return;
        th.emplace_back([this, ensure, cursor, n] {
            const size_t ahead = size_t(1) << 30;
            size_t done = 0;
            while (!stop && done < n) {
                const size_t want = cursor() + ahead < n ? cursor() + ahead : n;
                if (want > done) { if (!ensure(want)) return; done = want; }
                else std::this_thread::sleep_for(std::chrono::microseconds(500));
            }
        });
    }
    // A malloc()ed result: its pages are touched by a few threads, in stripes dealt round-robin (stripe s by thread s % k) so that the
    // FRONT of the buffer -- where the first batch lands -- is resident first; wait_front(upto) returns as soon as every stripe below
    // `upto` has been touched (the whole buffer took 5.4 ms with contiguous shares, and the first copy back waited for all of it).
    static constexpr size_t kStripe = size_t(2) << 20;
    uint8_t* tp = nullptr; size_t tn = 0; unsigned tk = 0;
    std::unique_ptr<std::atomic<size_t>[]> stripes_done;       // per thread: stripes it has finished
    void start(uint8_t* p, size_t n) {
        if (!p || n < (size_t(64) << 20)) return;
        {   // transparent huge pages where the system allows them on request: 2 MiB per fault instead of 4 KiB
            uintptr_t a0 = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
            (void)madvise(reinterpret_cast<void*>(a0), (n - 4096) & ~size_t(4095), MADV_HUGEPAGE);
        }
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned k = hw >= 32 ? 16 : hw >= 8 ? 4 : 1;
        tp = p; tn = n; tk = k;
        stripes_done.reset(new std::atomic<size_t>[k]);
        for (unsigned i = 0; i < k; ++i) stripes_done[i] = 0;
        const size_t n_stripes = (n + kStripe - 1) / kStripe;
        for (unsigned i = 0; i < k; ++i)
            th.emplace_back([this, p, n, k, i, n_stripes] {
                for (size_t s = i; s < n_stripes; s += k) {
                    const size_t a = s * kStripe, len = n - a < kStripe ? n - a : kStripe;
                    for (size_t o = 0; o < len; o += 4096) { volatile uint8_t* q = p + a + o; *q = *q; }   // (keeps what is already there: the header)
                    stripes_done[i].store(s / k + 1, std::memory_order_release);
                }
            });
    }
    // every page below p + upto has been touched (returns at once when nothing was started)
    void wait_front(size_t upto) {
        if (!tk) return;
        if (upto > tn) upto = tn;
        const size_t need = (upto + kStripe - 1) / kStripe;          // stripes 0 .. need-1
        for (unsigned i = 0; i < tk; ++i) {
            const size_t mine = need > i ? (need - i + tk - 1) / tk : 0;     // how many of them are thread i's
            while (stripes_done[i].load(std::memory_order_acquire) < mine) std::this_thread::yield();
        }
    }
    // (the touching threads of a malloc'ed buffer run to completion; the allocator of a mapped file is told to stop)
    void join() { for (auto& t : th) t.join(); th.clear(); }
    void cancel() { stop = true; join(); }
    ~Prefault() { cancel(); }
};

// ---- sequential byte source / sink over memory or FILE* ----
struct Source {
    const uint8_t* mem = nullptr; size_t size = 0; FILE* f = nullptr; bool own = false; uint8_t* loaded = nullptr;
    bool open(const uint8_t* in, size_t szin, bool infile) {
        if (infile) {
            f = fopen(reinterpret_cast<const char*>(in), "rb");          // tsq_threads.cpp:294-310
            if (!f) return false;
            own = true;
            fseek(f, 0, SEEK_END); long s = ftell(f); fseek(f, 0, SEEK_SET);
            if (s < 0) return false;
            size = (size_t)s;
            // Files that fit in memory are read whole, by several threads, and then go down the memory path (device-filling
            // batches, direct copies); larger ones stream through pinned staging in smaller batches.
            if (size > 0 && size <= env_size("TSQ_AMD_FILE_INMEM_MAX", size_t(64) << 20)) {
                loaded = static_cast<uint8_t*>(malloc(size + kHalo));
                if (loaded) {
                    if (io_par(pread, fileno(f), loaded, size, 0)) mem = loaded; else { free(loaded); loaded = nullptr; }
                }
            }
        } else { mem = in; size = szin; }
        return true;
    }
    // read [at, at+len) into dst; returns bytes obtained.  Positional reads (pread, several threads for a large piece): safe from
    // any thread, the scheduler's frame walk and a feeder's batch read do not disturb each other.
    size_t read_at(size_t at, size_t len, uint8_t* dst) {
        if (at >= size) return 0;
        if (len > size - at) len = size - at;
        if (mem) { memcpy(dst, mem + at, len); return len; }
        return io_par(pread, fileno(f), dst, len, at) ? len : 0;
    }
    ~Source() { if (own && f) fclose(f); free(loaded); }
};

struct Sink {
    uint8_t* mem = nullptr; size_t cap = 0, at = 0; FILE* f = nullptr; bool own = false; bool failed = false;
    std::atomic<size_t> cursor{0};                        // `at` as the allocator thread of a mapped file may read it
    bool open_file(const char* path) { f = fopen(path, "wb"); own = true; return f != nullptr; }
    // Mapped output files.  The file is ftruncate'd to its bound (sparse) and its pages are provided by fallocate(2), 64 MiB at a
    // time from the start, before anything is copied into them: a memcpy into a page the file system cannot back (disk or quota
    // full) would raise SIGBUS and kill the caller's process, where the positional-write path reports a failed job.  Linux fallocate
    // is called directly: on a file system without it (NFS before 4.2, FUSE, vfat) glibc's posix_fallocate falls back to a
    // read-one-byte-write-it-back emulation that races with the writers of the same range.  EOPNOTSUPP turns the sink into a
    // positional writer (pwrite through the same descriptor); any other failure (ENOSPC, EDQUOT, EFBIG) fails the job.
    std::mutex alloc_m;
    size_t alloc_upto = 0;
    bool no_falloc = false;
    // `ahead` = the call comes from the look-ahead thread (Prefault::start_mapped), which runs up to 1 GiB in front of the write
    // cursor towards the job's BOUND: space it cannot get is space the job may never need, so its failure stops the look-ahead and
    // nothing else.  Only the writer's own range failing (the exact bytes about to be copied) fails the job.
    bool ensure_allocated(size_t upto, bool ahead = false) {
        if (!mapped) return true;
        std::lock_guard<std::mutex> g(alloc_m);
        if (upto > cap) upto = cap;
        if (upto <= alloc_upto) return true;                 // already backed, whatever happened further out
        if (no_falloc || alloc_failed) return false;
        const size_t step = size_t(64) << 20;
        while (alloc_upto < upto) {
            size_t len = cap - alloc_upto < step ? cap - alloc_upto : step;
            if (ahead_refused && !ahead) len = upto - alloc_upto;   // the disk is nearly full: ask for what is written, not for a step
            int r;
            do r = fallocate(map_fd, 0, (off_t)alloc_upto, (off_t)len); while (r != 0 && errno == EINTR);
            if (r != 0 && errno != EOPNOTSUPP && errno != ENOSYS && !ahead && len > upto - alloc_upto) {
                len = upto - alloc_upto;                     // a whole step does not fit; the range being written may
                do r = fallocate(map_fd, 0, (off_t)alloc_upto, (off_t)len); while (r != 0 && errno == EINTR);
            }
            if (r != 0) {
                if (errno == EOPNOTSUPP || errno == ENOSYS) no_falloc = true;
                else if (ahead) ahead_refused = true;
                else alloc_failed = true;
                return false;
            }
            alloc_upto += len;
        }
        return true;
    }
    bool ahead_refused = false;                              // (under alloc_m)
    std::atomic<bool> alloc_failed{false};
    // small file outputs: collected like a memory sink, written once at the end.  Larger ones are MAPPED at their bound and
    // filled like memory too (pinned staging + a CPU copy into the page cache, whose pages a background fallocate provides ahead of
    // the batches; the file is cut to its length at the end): write() calls on one file take the inode lock one after the other and
    // top out near 3-6 GB/s however many threads make them (tools/streamed_files.sh).
    bool open_file_buffered(const char* path, size_t capacity) {
        if (!open_file(path)) return false;
        if (capacity <= env_size("TSQ_AMD_FILE_MAP_MIN", size_t(64) << 20)) {
            mem = static_cast<uint8_t*>(malloc(capacity ? capacity : 1));
            if (mem) { cap = capacity; pending_file = f; f = nullptr; }
        } else if (!getenv("TSQ_AMD_FILE_NO_MMAP")) {
            fclose(f); f = nullptr;
            map_fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (map_fd < 0) return false;
            if (ftruncate(map_fd, (off_t)capacity) == 0) {
                void* p = mmap(nullptr, capacity, PROT_READ | PROT_WRITE, MAP_SHARED, map_fd, 0);
                if (p != MAP_FAILED) { mem = static_cast<uint8_t*>(p); cap = capacity; mapped = true; return true; }
            }
            ::close(map_fd); map_fd = -1;
            return open_file(path);                                      // (a file system without mmap: positional writes)
        }
        return true;
    }
    bool finish() {
        if (mapped) {
            (void)munmap(mem, cap); mem = nullptr; mapped = false;
            if (ftruncate(map_fd, (off_t)at) != 0) failed = true;
            ::close(map_fd); map_fd = -1;
            return !failed;
        }
        if (pending_file) {
            if (!failed && fwrite(mem, 1, at, pending_file) != at) failed = true;
            f = pending_file; pending_file = nullptr;
            free(mem); mem = nullptr;
        }
        return !failed;
    }
    FILE* pending_file = nullptr;
    int map_fd = -1; bool mapped = false;
    bool open_mem(size_t capacity) { mem = static_cast<uint8_t*>(malloc(capacity ? capacity : 1)); cap = capacity; return mem != nullptr; }
    // where the next n bytes will land (memory sinks): the device copies there directly
    uint8_t* claim(size_t n) { if (f || at + n > cap) { failed = true; return nullptr; } uint8_t* p = mem + at; at += n; return p; }
    void write(const uint8_t* p, size_t n) {
        if (f) {      // streamed file output: positional writes at the running offset, several threads for a large piece
            if (at == 0) (void)fflush(f);
            if (!io_par([](int fd, uint8_t* q, size_t len, off_t off) { return pwrite(fd, q, len, off); }, fileno(f), const_cast<uint8_t*>(p), n, at)) failed = true;
        }
        else if (mapped && at + n <= cap && !ensure_allocated(at + n)) {
            // no pages behind this range: the file system has no fallocate -> positional writes through the descriptor (a full disk is
            // then an error return, not a signal); it has, and refused -> the job fails
            if (alloc_failed) failed = true;
            else if (!io_par([](int fd, uint8_t* q, size_t len, off_t off) { return pwrite(fd, q, len, off); }, map_fd, const_cast<uint8_t*>(p), n, at)) failed = true;
        }
        else if (at + n <= cap) {
            const size_t k = mapped ? io_threads_for(n) : 1;      // (a mapped file: the copy takes the pages' minor faults, in a few threads)
            if (k <= 1) memcpy(mem + at, p, n);
            else {
                const size_t per = ((n / k) + 4095) & ~size_t(4095);
                std::vector<std::thread> th;
                for (size_t i = 0; i < k && per * i < n; ++i)
                    th.emplace_back([this, p, n, per, i] { const size_t a = per * i; memcpy(mem + at + a, p + a, n - a < per ? n - a : per); });
                for (auto& t : th) t.join();
            }
        }
        else failed = true;
        at += n;
        cursor = at;
    }
    ~Sink() {
        if (mapped) { (void)munmap(mem, cap); ::close(map_fd); }
        else if (pending_file) { fclose(pending_file); free(mem); }
        else if (own && f) fclose(f);
    }
};

// ---- one pipeline lane ----
struct Lane {
    tsqa_ctx* dev = nullptr;
    uint8_t *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    FrameInfo* h_frames = nullptr;
    uint64_t* h_frame_at = nullptr;                            // pinned: frame offsets of a compressed batch (n_frames + 1)
    size_t in_cap = 0, out_cap = 0, frames_cap = 0, h_in_cap = 0, h_out_cap = 0;
    uint64_t* h_size = nullptr; int32_t* h_status = nullptr;   // pinned result words
    hipEvent_t ev = nullptr;

    bool init(int device) {
        if (tsqa_create(device, &dev) != TSQA_OK) return false;
        if (hipHostMalloc(&h_size, sizeof(uint64_t), hipHostMallocPortable) != hipSuccess) return false;
        if (hipHostMalloc(&h_status, sizeof(int32_t), hipHostMallocPortable) != hipSuccess) return false;
        return hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    }
    // Device buffers always; pinned host staging only where a FILE* is on that side (memory-to-memory jobs copy
    // straight between the caller's buffers and HBM: staging would cost a full extra pass over host memory).
    bool reserve(size_t in_bytes, size_t out_bytes, size_t n_frames, bool stage_in, bool stage_out) {
        (void)hipSetDevice(dev->device);
        if (in_bytes > in_cap) {
            (void)hipFree(d_in); d_in = nullptr; in_cap = 0;
            (void)hipHostFree(h_in); h_in = nullptr; h_in_cap = 0;
            if (hipMalloc(&d_in, in_bytes) != hipSuccess) return false;
            in_cap = in_bytes;
        }
        if (stage_in && h_in_cap < in_cap) {
            (void)hipHostFree(h_in); h_in = nullptr; h_in_cap = 0;
            if (hipHostMalloc(&h_in, in_cap, hipHostMallocPortable) != hipSuccess) return false;
            h_in_cap = in_cap;
        }
        if (out_bytes > out_cap) {
            (void)hipFree(d_out); d_out = nullptr; out_cap = 0;
            (void)hipHostFree(h_out); h_out = nullptr; h_out_cap = 0;
            if (hipMalloc(&d_out, out_bytes) != hipSuccess) return false;
            out_cap = out_bytes;
        }
        if (stage_out && h_out_cap < out_cap) {
            (void)hipHostFree(h_out); h_out = nullptr; h_out_cap = 0;
            if (hipHostMalloc(&h_out, out_cap, hipHostMallocPortable) != hipSuccess) return false;
            h_out_cap = out_cap;
        }
        if (n_frames > frames_cap) {
            (void)hipHostFree(h_frames); h_frames = nullptr; frames_cap = 0;
            (void)hipHostFree(h_frame_at); h_frame_at = nullptr;
            if (hipHostMalloc(&h_frames, n_frames * sizeof(FrameInfo), hipHostMallocPortable) != hipSuccess) return false;
            if (hipHostMalloc(&h_frame_at, (n_frames + 1) * sizeof(uint64_t), hipHostMallocPortable) != hipSuccess) return false;
            frames_cap = n_frames;
        }
        return true;
    }
    void destroy() {
        if (dev) (void)hipSetDevice(dev->device);
        if (ev) (void)hipEventDestroy(ev);
        (void)hipHostFree(h_in); (void)hipHostFree(h_out); (void)hipHostFree(h_frames); (void)hipHostFree(h_frame_at);
        (void)hipHostFree(h_size); (void)hipHostFree(h_status);
        (void)hipFree(d_in); (void)hipFree(d_out);
        tsqa_destroy(dev);
        dev = nullptr;
    }
};

struct Job {
    uint8_t* in = nullptr; size_t szin = 0; bool infile = false;
    uint8_t** out = nullptr; size_t* szout = nullptr; bool outfile = false;
    std::string out_path;
    bool ext = false;
    uint32_t id = 0;
    std::function<void(uint32_t, bool)> done;
    std::function<void(uint32_t, double)> progress;
};

struct InFlight { size_t lane; uint32_t first_block, n_blocks; size_t out_bytes; };

// One thread per listed device: it issues that device's batches (file read, copy to the device, launches).  A copy from pageable
// memory holds the issuing thread for its whole duration; with one issuing thread for all devices the feeds would queue behind each
// other (the reference gives every worker its own thread, tsq_threads.cpp:137-189).
struct Feeder {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool quit = false;
    void start() {
        th = std::thread([this] {
            for (;;) {
                std::function<void()> fn;
                {
                    std::unique_lock<std::mutex> g(m);
                    cv.wait(g, [this] { return quit || !q.empty(); });
                    if (q.empty()) return;
                    fn = std::move(q.front());
                    q.pop_front();
                }
                fn();
            }
        });
    }
    void push(std::function<void()> fn) { { std::lock_guard<std::mutex> g(m); q.push_back(std::move(fn)); } cv.notify_one(); }
    void stop() { { std::lock_guard<std::mutex> g(m); quit = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};
// "the lane's batch has been issued" (1) or could not be (2), set by the feeder, awaited by whoever drains the lane
struct Issued {
    std::mutex m;
    std::condition_variable cv;
    int state = 0;
    void reset() { std::lock_guard<std::mutex> g(m); state = 0; }
    void set(bool ok) { { std::lock_guard<std::mutex> g(m); state = ok ? 1 : 2; } cv.notify_all(); }
    bool wait() { std::unique_lock<std::mutex> g(m); cv.wait(g, [this] { return state != 0; }); return state == 1; }
};

class Scheduler {
public:
    explicit Scheduler(bool compress, bool verbose) : compress_(compress), verbose_(verbose) {}

    bool start() {
        std::vector<int> devs = device_list();
        n_devices_ = (uint32_t)devs.size();
        size_t per_dev = env_size("TSQ_AMD_LANES", 4);
        for (size_t k = 0; k < per_dev * devs.size(); ++k) {
            Lane l;
            if (!l.init(devs[k % devs.size()])) { l.destroy(); for (auto& x : lanes_) x.destroy(); lanes_.clear(); return false; }
            lanes_.push_back(l);
            issued_.emplace_back(new Issued());
        }
        for (size_t d = 0; d < devs.size(); ++d) { feeders_.emplace_back(new Feeder()); feeders_.back()->start(); }
        // one batch should fill the device: a block is one workgroup for its whole (serial) parse, and two of them share a CU
        // (the encoder's lean layout) at 1.5x the throughput of one
        batch_blocks_ = (uint32_t)env_size("TSQ_AMD_BATCH_BLOCKS", 512);
        file_batch_blocks_ = (uint32_t)env_size("TSQ_AMD_FILE_BATCH_BLOCKS", batch_blocks_ < 64 ? batch_blocks_ : 64);
        thread_ = std::thread([this] { loop(); });
        return true;
    }
    uint32_t n_cus() const { return lanes_.empty() ? 0 : (uint32_t)lanes_[0].dev->n_cus; }

    uint32_t submit(Job&& j) {
        std::lock_guard<std::mutex> g(m_);
        j.id = next_id_++;
        uint32_t id = j.id;
        inflight_++;
        q_.push_back(std::move(j));
        cv_.notify_all();
        return id;
    }
    // tsq_context.cpp:150-155: deallocation waits for every queued job, then joins.
    void stop() {
        {
            std::unique_lock<std::mutex> g(m_);
            idle_cv_.wait(g, [this] { return inflight_ == 0; });
            exit_ = true;
            cv_.notify_all();
        }
        if (thread_.joinable()) thread_.join();
        for (auto& f : feeders_) f->stop();
        feeders_.clear();
        for (auto& l : lanes_) l.destroy();
        lanes_.clear();
    }

private:
    // Blocks per batch for a job of nb blocks.  A batch is what one device works on at a time, so a job that would fit
    // one batch is still cut into one slice of consecutive blocks per listed device (the analogue of block i ->
    // worker i % num_cores, tsq_threads.cpp:71,463, at slice granularity: each slice carries its 128-byte look-ahead,
    // and the host gathers the slices in block order).
    uint32_t job_batch(uint32_t nb, bool through_files) const {
        uint32_t batch = through_files ? file_batch_blocks_ : batch_blocks_;
        // Memory-mode jobs are cut into one batch per pipeline lane (at least 32 blocks each): the lanes' streams overlap one
        // batch's copy to the device, another's kernels and a third's copy back (PCIe is full duplex; the kernels of different
        // lanes run side by side), which a single device-filling batch cannot -- its three phases would run one after the other.
        if (!through_files && !lanes_.empty()) {
            const uint32_t per_lane = (nb + (uint32_t)lanes_.size() - 1) / (uint32_t)lanes_.size();
            const uint32_t cut = per_lane < 32u ? 32u : per_lane;
            if (cut < batch) batch = cut;
        }
        if (n_devices_ > 1) {
            const uint32_t per_dev = (nb + n_devices_ - 1) / n_devices_;
            if (per_dev < batch) batch = per_dev ? per_dev : 1;
        }
        return batch;
    }
    // The sizes of a job's batches, in order.  File jobs and jobs over several devices: equal batches (job_batch).  A memory-to-memory
    // job on one device is paced by the link, not by the batch count, so its batches are RAMPED (tools/tsq_cli b with TSQ_AMD_DEBUG=1):
    //   decompress  small first: nothing can come back before the first batch has gone up and been decoded, so the first batch is a
    //               sixteenth of the job and the following ones grow by half -- the copy back starts after 3.5 ms instead of 6.5 and
    //               then runs without a gap while the larger batches go up beside it (the link is full duplex);
    //   compress    small last: every block takes the encoder's block latency from the moment its bytes are on the device, the job ends
    //               with the last batch's frames coming back, so the last batch is the small one.
    std::vector<uint32_t> job_batches(uint32_t nb, bool through_files, bool compress) const {
        std::vector<uint32_t> out;
        const uint32_t batch = job_batch(nb, through_files);
        if (through_files || n_devices_ > 1 || nb < 64 || getenv("TSQ_AMD_NO_RAMP")) {
            for (uint32_t b = 0; b < nb; b += batch) out.push_back(nb - b < batch ? nb - b : batch);
            return out;
        }
        if (compress) {
            // (a batch holds its lane for the encoder's whole block latency: no more batches than lanes, each half again as large as the
            //  one after it)
            const uint32_t k = (uint32_t)lanes_.size() < 2u ? 1u : (uint32_t)lanes_.size();
            double w = 1.0, sum = 0.0;
            std::vector<double> ws(k);
            for (uint32_t i = 0; i < k; ++i) { ws[k - 1u - i] = w; sum += w; w *= 1.5; }
            uint32_t left = nb;
            for (uint32_t i = 0; i < k && left; ++i) {
                uint32_t take = i + 1u == k ? left : (uint32_t)(nb * ws[i] / sum + 0.5);
                if (take > left) take = left;
                if (take > batch_blocks_) take = batch_blocks_;
                if (take == 0u) take = 1u;
                out.push_back(take);
                left -= take;
            }
            while (left) { const uint32_t take = left < batch_blocks_ ? left : batch_blocks_; out.push_back(take); left -= take; }   // (more than lanes x 512 blocks)
            return out;
        }
        uint32_t left = nb, size = nb / 16u < 8u ? 8u : nb / 16u;
        while (left) {
            uint32_t take = size < left ? size : left;
            if (left - take < 8u) take = left;                    // (no crumbs)
            if (take > batch_blocks_) take = batch_blocks_;
            out.push_back(take);
            left -= take;
            size += size / 2u;
        }
        return out;
    }
    // Blocks per device-to-host piece: progress is reported per block as its bytes land; a piece of a few blocks keeps
    // the copies large enough for the DMA engines (one 4 MiB block per copy costs about a third of the bandwidth).
    // (through staging -- a file on the output side -- the pieces are larger: each is copied to pinned memory while the one before
    //  it is written, and a piece's write wants tens of megabytes to be worth its threads)
    static uint32_t progress_piece(uint32_t n_blocks, bool staged = false) {
        if (staged && n_blocks >= 32) return n_blocks / 4;
        // a staged (file) batch below 32 blocks still goes in at least two pieces, so that the next piece comes back while this one is
        // written and the per-block progress calls do not arrive as one burst per batch (the reference reports a block as it lands,
        // tsq_threads.cpp:248-254)
        if (staged && n_blocks >= 2) return (n_blocks + 1) / 2 < 8u ? (n_blocks + 1) / 2 : 8u;
        // (a copy to pageable memory has a fixed cost of its own -- the runtime locks and unlocks the pages around it --: 8 MB pieces came
        //  back at 17-26 GB/s where 32-64 MB pieces reach the link's 46-50)
        return n_blocks >= 32 ? 16u : n_blocks >= 16 ? 8u : n_blocks;
    }

    void loop() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return exit_ || !q_.empty(); });
                if (q_.empty()) return;
                j = std::move(q_.front());
                q_.pop_front();
            }
            if (verbose_) {
                // tsq_threads.cpp:385-391,832-838: a verbose context prints every block's progress in front of the caller's callback
                std::function<void(uint32_t, double)> user = std::move(j.progress);
                j.progress = [user](uint32_t id, double p) { printf("Job %u progress: %.2f%%\r", id, p * 100.0); if (user) user(id, p); };
            }
            bool ok = compress_ ? run_compress(j) : run_decompress(j);
            if (verbose_) {                                     // tsq_threads.cpp:367-373,814-821
                if (ok) printf("Job %u completed successfully.\n", j.id);
                else printf("Job %u failed.                \n", j.id);
            }
            if (j.done) j.done(j.id, ok);                       // tsq_threads.cpp:256-262,657-663
            {
                std::lock_guard<std::mutex> g(m_);
                inflight_--;
                idle_cv_.notify_all();
            }
        }
    }

    // ------------------------------------------------------------------ compression
    bool run_compress(Job& j) {
        Source src;
        if (!src.open(j.in, j.szin, j.infile) || src.size == 0) {
            if (verbose_ && j.infile) printf("Error: could not open input file.\n");          // tsq_threads.cpp:298-301
            return false;
        }
        const size_t total = src.size;
        const uint32_t nb = (uint32_t)tsqa_block_count(total);       // tsq_threads.cpp:313
        Sink sink;
        if (j.outfile) {
            if (!sink.open_file_buffered(j.out_path.c_str(), tsqa_container_bound(total))) {
                if (verbose_) printf("Error: could not open output file.\n");                   // tsq_threads.cpp:323-326
                return false;
            }
        } else if (!sink.open_mem(tsqa_container_bound(total))) {                               // tsq_threads.cpp:339
            if (verbose_) printf("Error: could not allocate output buffer.\n");                 // tsq_threads.cpp:343-346
            return false;
        }

        uint8_t header[16];                                          // tsq_threads.cpp:333-335,355-359
        memcpy(header, "TSQ1", 4); memcpy(header + 4, &nb, 4);
        uint64_t t64 = total; memcpy(header + 8, &t64, 8);
        sink.write(header, 16);

        bool ok = true;
        std::deque<InFlight> fly;
        uint32_t done_blocks = 0;
        // (a mapped output file takes its bytes through pinned staging and a CPU copy: a device copy straight into file-backed pages
        //  runs at a fraction of the link's rate)
        const bool stage_in = src.mem == nullptr, stage_out = sink.mem == nullptr || sink.mapped;
        const std::vector<uint32_t> sizes = job_batches(nb, stage_in || stage_out, true);
        Marks mk; mk.at("compress: buffers opened");
        Prefault touch;
        if (sink.mapped) touch.start_mapped([&sink](size_t upto) { return sink.ensure_allocated(upto, true); }, [&sink] { return sink.cursor.load(); }, sink.cap);
        else if (!stage_out) touch.start(sink.mem, sink.cap < total ? sink.cap : total);
        auto drain_one = [&]() {
            InFlight f = fly.front(); fly.pop_front();
            Lane& l = lanes_[f.lane];
            if (!issued_[f.lane]->wait()) { ok = false; return; }     // the lane's feeder has issued the batch (or failed to)
            (void)hipSetDevice(l.dev->device);
            if (hipEventSynchronize(l.ev) != hipSuccess || *l.h_status != 0) { ok = false; return; }
            size_t sz = (size_t)*l.h_size;                            // batch container: 16-byte header + frames
            mk.at("compress: kernels done");
            if (sz < 16 || sz > l.out_cap) { ok = false; return; }
            // (a mapped file is populated from its start while the batches land: nothing to wait for; a malloc'ed buffer is touched in
            //  one share per thread and must be complete)
            if (!stage_out) { touch.wait_front(sink.at + sz); mk.at("compress: output pages touched"); }
            // The frames come back in pieces of a few blocks, in block order, and every block reports progress as soon as
            // its bytes have landed (tsq_threads.cpp:226-254: the writer emits a frame, then calls progress_cb).
            const uint64_t* fat = l.h_frame_at;                       // frame offsets inside the batch container
            const uint32_t piece = progress_piece(f.n_blocks, stage_out);
            // (through staging the next piece is copied to pinned memory while this one is written)
            auto fetch = [&](uint32_t p0) -> bool {
                const uint32_t p1 = p0 + piece < f.n_blocks ? p0 + piece : f.n_blocks;
                const size_t from = (size_t)fat[p0], len = (size_t)fat[p1] - from;
                if (from < 16 || from + len > sz) return false;
                return hipMemcpyAsync(l.h_out + from, l.d_out + from, len, hipMemcpyDeviceToHost, l.dev->stream) == hipSuccess;
            };
            if (stage_out && !fetch(0)) { ok = false; return; }
            for (uint32_t p0 = 0; p0 < f.n_blocks && ok; p0 += piece) {
                const uint32_t p1 = p0 + piece < f.n_blocks ? p0 + piece : f.n_blocks;
                const size_t from = (size_t)fat[p0], len = (size_t)fat[p1] - from;
                if (from < 16 || from + len > sz) { ok = false; return; }
                if (stage_out) {
                    if (hipStreamSynchronize(l.dev->stream) != hipSuccess) { ok = false; return; }
                    if (p1 < f.n_blocks && !fetch(p1)) { ok = false; return; }
                    sink.write(l.h_out + from, len);
                } else {
                    uint8_t* dst = sink.claim(len);                   // frames land in the caller's buffer directly
                    if (!dst || hipMemcpyAsync(dst, l.d_out + from, len, hipMemcpyDeviceToHost, l.dev->stream) != hipSuccess ||
                        hipStreamSynchronize(l.dev->stream) != hipSuccess) { ok = false; return; }
                }
                for (uint32_t b = p0; b < p1; ++b) {                  // tsq_threads.cpp:248-254
                    done_blocks++;
                    if (j.progress) j.progress(j.id, (double)done_blocks / (double)nb);
                }
            }
            mk.at("compress: D2H done");
        };

        for (uint32_t b0 = 0, k = 0; b0 < nb && ok; b0 += sizes[k], ++k) {
            const uint32_t bn = sizes[k];
            const size_t lane_i = k % lanes_.size();
            while (ok && fly.size() >= lanes_.size()) drain_one();
            if (!ok) break;
            Lane& l = lanes_[lane_i];
            const size_t at = (size_t)b0 * kBlockSize;
            const size_t want = (size_t)bn * kBlockSize;
            const size_t n = total - at < want ? total - at : want;
            if (!l.reserve(want + kHalo, tsqa_container_bound(want), bn, stage_in, stage_out)) { ok = false; break; }
            // the batch is issued by the lane's device feeder: read (file sources), copy to the device, kernels, result words
            Issued* flag = issued_[lane_i].get();
            flag->reset();
            const int feeder = (int)(lane_i % feeders_.size());
            const uint32_t ext = j.ext ? 1u : 0u;
            feeders_[feeder]->push([&src, &mk, &l, flag, at, n, total, bn, stage_in, ext, feeder] {
                bool good = true;
                (void)hipSetDevice(l.dev->device);
                hipStream_t s = l.dev->stream;
                // the batch plus the first bytes of the next one: block k's look-ahead reads block k+1
                size_t got = total - at < n + kHalo ? total - at : n + kHalo;
                if (stage_in) {
                    got = src.read_at(at, n + kHalo, l.h_in);
                    mk.at("compress: batch read", feeder);
                    good = got >= n && hipMemcpyAsync(l.d_in, l.h_in, got, hipMemcpyHostToDevice, s) == hipSuccess;
                } else good = hipMemcpyAsync(l.d_in, src.mem + at, got, hipMemcpyHostToDevice, s) == hipSuccess;
                mk.at("compress: H2D issued", feeder);
                good = good && hipMemsetAsync(l.dev->d_status, 0, sizeof(int32_t), s) == hipSuccess;
                good = good && l.dev->launch_encode(l.d_in, n, got, ext, l.dev->d_status, s) == TSQA_OK;
                good = good && l.dev->launch_pack(n, ext, l.d_out, l.out_cap, l.dev->d_size, l.dev->d_status, s) == TSQA_OK;
                if (good) {
                    (void)hipMemcpyAsync(l.h_size, l.dev->d_size, sizeof(uint64_t), hipMemcpyDeviceToHost, s);
                    (void)hipMemcpyAsync(l.h_frame_at, l.dev->frame_at, (bn + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
                    (void)hipMemcpyAsync(l.h_status, l.dev->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s);
                    good = hipEventRecord(l.ev, s) == hipSuccess;
                }
                flag->set(good);
            });
            fly.push_back({lane_i, b0, bn, 0});
        }
        while (ok && !fly.empty()) drain_one();
        for (const InFlight& f : fly) (void)issued_[f.lane]->wait();    // (after a failure: no feeder may still be working on this job's buffers)
        for (auto& l : lanes_) { (void)hipSetDevice(l.dev->device); (void)hipStreamSynchronize(l.dev->stream); }
        touch.cancel();                                               // nobody may still be touching (or allocating behind) the buffer when it is freed
        if (!sink.finish()) ok = false;
        if (!j.outfile) {
            if (ok) { *j.out = sink.mem; *j.szout = sink.at; }       // tsq_threads.cpp:375-379; caller free()s
            else { free(sink.mem); }
        }
        return ok;
    }

    // ------------------------------------------------------------------ decompression
    bool run_decompress(Job& j) {
        Source src;
        if (!src.open(j.in, j.szin, j.infile)) {
            if (verbose_ && j.infile) printf("Error opening input file: %s\n", reinterpret_cast<const char*>(j.in));   // tsq_threads.cpp:714-717
            return false;
        }
        uint8_t header[16];
        if (src.read_at(0, 16, header) != 16 || memcmp(header, "TSQ1", 4) != 0) {              // tsq_threads.cpp:732-752
            if (verbose_) printf("Error: signature mismatch (expected TSQ1).\n");
            return false;
        }
        uint32_t nb; uint64_t total;
        memcpy(&nb, header + 4, 4); memcpy(&total, header + 8, 8);
        if (nb == 0) {                                                                         // tsq_threads.cpp:759-768
            if (verbose_) printf("Error: no blocks to decode in input file.\n");
            return false;
        }
        // The header is not trusted with an allocation before it has been held against the container: a frame is at
        // least 6 bytes, a block at most 4 MiB, and no stream expands more than 64x (eight 64-byte copies per 13-byte group).
        if ((uint64_t)nb > (src.size - 16) / 6) return false;
        if ((uint64_t)nb * kBlockSize < total || total / 64 > src.size) return false;
        Sink sink;
        if (j.outfile) { if (!sink.open_file_buffered(j.out_path.c_str(), (size_t)total + 128)) return false; }
        else if (!sink.open_mem((size_t)total + 128)) return false;                            // tsq_threads.cpp:795

        Marks mk; mk.at("decompress: buffers opened");
        std::atomic<bool> ok{true};
        std::deque<InFlight> fly;
        std::mutex fly_m;
        std::condition_variable fly_cv;
        bool fly_closed = false;
        uint32_t done_blocks = 0;
        const bool stage_in = src.mem == nullptr, stage_out = sink.mem == nullptr || sink.mapped;
        const uint32_t batch = job_batch(nb, stage_in || stage_out);
        const std::vector<uint32_t> sizes = job_batches(nb, stage_in || stage_out, false);
        Prefault touch;
        bool touching = false;
        auto drain = [&](const InFlight& f) {
            Lane& l = lanes_[f.lane];
            if (!issued_[f.lane]->wait()) { ok = false; return; }     // the lane's feeder has issued the batch (or failed to)
            (void)hipSetDevice(l.dev->device);
            if (hipEventSynchronize(l.ev) != hipSuccess) { ok = false; return; }
            if (*l.h_status == tsq::kErrStall) {
                // a sibling workgroup of a several-workgroups-per-block decode did not get onto the GPU in time: the batch is still on
                // the device -- once more with one workgroup per block
                hipStream_t s = l.dev->stream;
                bool good = hipMemsetAsync(l.dev->d_status, 0, sizeof(int32_t), s) == hipSuccess;
                good = good && l.dev->launch_decode(l.d_in, f.n_blocks, l.d_out, l.dev->d_status, s, 4) == TSQA_OK;
                good = good && hipMemcpyAsync(l.h_status, l.dev->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s) == hipSuccess;
                good = good && hipStreamSynchronize(s) == hipSuccess;
                mk.at("decompress: batch decoded again (stall)");
                if (!good) { ok = false; return; }
            }
            if (*l.h_status != 0) { ok = false; return; }
            mk.at("decompress: kernels done");
            // the blocks come back in pieces of a few blocks, in order; each block reports progress once it has landed
            // (tsq_threads.cpp:648-655: the writer copies a block out, then calls progress_cb)
            uint8_t* dst = stage_out ? nullptr : sink.claim(f.out_bytes);
            if (!stage_out && dst) { touch.wait_front((size_t)(dst - sink.mem) + f.out_bytes); mk.at("decompress: output pages touched"); }
            if (!stage_out && !dst) { ok = false; return; }
            const FrameInfo* fr = l.h_frames;
            const uint32_t piece = progress_piece(f.n_blocks, stage_out);
            auto piece_span = [&](uint32_t p0, size_t& from, size_t& len) {
                const uint32_t p1 = p0 + piece < f.n_blocks ? p0 + piece : f.n_blocks;
                from = (size_t)fr[p0].out_at;
                len = (size_t)(fr[p1 - 1].out_at + fr[p1 - 1].out_len) - from;
            };
            auto fetch = [&](uint32_t p0) {      // piece p0.. to its place in the caller's buffer, or in the pinned staging buffer
                size_t from, len; piece_span(p0, from, len);
                return hipMemcpyAsync(stage_out ? l.h_out + from : dst + from, l.d_out + from, len, hipMemcpyDeviceToHost, l.dev->stream) == hipSuccess;
            };
            if (!fetch(0)) { ok = false; return; }
            for (uint32_t p0 = 0; p0 < f.n_blocks && ok; p0 += piece) {
                const uint32_t p1 = p0 + piece < f.n_blocks ? p0 + piece : f.n_blocks;
                size_t from, len; piece_span(p0, from, len);
                if (hipStreamSynchronize(l.dev->stream) != hipSuccess) { ok = false; return; }
                if (stage_out && p1 < f.n_blocks && !fetch(p1)) { ok = false; return; }      // the next piece comes in while this one is written
                if (stage_out) sink.write(l.h_out + from, len);
                else if (p1 < f.n_blocks && !fetch(p1)) { ok = false; return; }
                for (uint32_t b = p0; b < p1; ++b) {
                    done_blocks++;
                    if (j.progress) j.progress(j.id, (double)done_blocks / (double)nb);
                }
            }
            mk.at("decompress: D2H done");
        };

        // The copies back run on their own thread (the reference's writer thread, tsq_threads.cpp:604-676): a copy from or to the
        // caller's pageable memory holds the calling thread for its whole duration, so with one thread the copies to the device of
        // the later batches and the copies back of the earlier ones would take turns on a link that can do both at once.
        std::thread drainer([&]() {
            for (;;) {
                InFlight f;
                {
                    std::unique_lock<std::mutex> g(fly_m);
                    fly_cv.wait(g, [&] { return fly_closed || !fly.empty(); });
                    if (fly.empty()) return;
                    f = fly.front();
                }
                if (ok) drain(f); else (void)issued_[f.lane]->wait();  // (after a failure: the feeder must be done with the lane all the same)
                {
                    std::lock_guard<std::mutex> g(fly_m);
                    fly.pop_front();                                  // only now may the lane be used again
                }
                fly_cv.notify_all();
            }
        });
        auto close_drainer = [&]() {
            { std::lock_guard<std::mutex> g(fly_m); fly_closed = true; }
            fly_cv.notify_all();
            if (drainer.joinable()) drainer.join();
        };
        size_t at = 16;                       // container cursor: the frame walk is serial (tsq_threads.cpp:513-524)
        uint64_t produced = 0;
        for (uint32_t b0 = 0, k = 0; b0 < nb && ok; ++k) {
            const size_t lane_i = k % lanes_.size();
            {
                std::unique_lock<std::mutex> g(fly_m);
                fly_cv.wait(g, [&] { return fly.size() < lanes_.size(); });
            }
            if (!ok) break;
            Lane& l = lanes_[lane_i];
            // sized by what is left of the job, not by the configured batch (a one-block container must not reserve gigabytes)
            const uint32_t sched = k < sizes.size() ? sizes[k] : batch;
            const uint32_t want_blocks = nb - b0 < sched ? nb - b0 : sched;
            size_t in_budget = (size_t)want_blocks * (3 + kSlotSize);
            if (in_budget > src.size - at) in_budget = src.size - at;
            if (!l.reserve(in_budget + 16, (size_t)want_blocks * kBlockSize + 256, want_blocks, stage_in, stage_out)) { ok = false; break; }
            // walk whole frames until the batch is full; the frames of a batch are one contiguous slice of the container
            size_t cur = 0; uint32_t bn = 0; size_t out_bytes = 0;
            while (b0 + bn < nb && bn < want_blocks) {
                uint8_t fh[6];
                if (src.read_at(at + cur, 6, fh) != 6) break;
                uint32_t frame = (uint32_t)fh[0] | ((uint32_t)fh[1] << 8) | ((uint32_t)fh[2] << 16);
                uint32_t len = frame & 0x7FFFFFu;                                             // tsq_threads.cpp:513-517
                if (len < 3 || len > kSlotSize) { ok = false; break; }                         // tsq_threads.cpp:526-531
                if (at + cur + 3 + len > src.size) { ok = false; break; }
                if (cur + 3 + len > l.in_cap) break;
                uint32_t usize = (uint32_t)fh[3] | ((uint32_t)fh[4] << 8) | ((uint32_t)fh[5] << 16);
                if (usize > kBlockSize || produced + out_bytes + usize > total) { ok = false; break; }
                FrameInfo& fi = l.h_frames[bn];
                fi.stream_at = cur + 3; fi.out_at = out_bytes; fi.stream_len = len; fi.ext = frame >> 23; fi.out_len = usize; fi.pad = 0;
                out_bytes += usize; cur += 3 + len; bn++;
            }
            if (!ok) break;
            if (bn == 0) { ok = false; break; }                        // truncated container
            // the result buffer is made resident only now that a first batch of frames has been found well formed
            if (!touching && sink.mapped) { touch.start_mapped([&sink](size_t upto) { return sink.ensure_allocated(upto, true); }, [&sink] { return sink.cursor.load(); }, (size_t)total); touching = true; }
            if (!touching && !stage_out) { touch.start(sink.mem, (size_t)total); touching = true; }
            // the batch's frames are one contiguous slice of the container: the lane's device feeder reads it (file sources), copies
            // it to the device and launches the kernels
            Issued* flag = issued_[lane_i].get();
            flag->reset();
            const int feeder = (int)(lane_i % feeders_.size());
            feeders_[feeder]->push([&src, &mk, &l, flag, at, cur, bn, stage_in, feeder] {
                (void)hipSetDevice(l.dev->device);
                hipStream_t s = l.dev->stream;
                bool good = l.dev->reserve(bn, false, false) == TSQA_OK;
                if (good && stage_in) { good = src.read_at(at, cur, l.h_in) == cur; mk.at("decompress: batch read", feeder); }
                good = good && hipMemcpyAsync(l.d_in, stage_in ? l.h_in : src.mem + at, cur, hipMemcpyHostToDevice, s) == hipSuccess;
                good = good && hipMemcpyAsync(l.dev->frames, l.h_frames, bn * sizeof(FrameInfo), hipMemcpyHostToDevice, s) == hipSuccess;
                good = good && hipMemsetAsync(l.dev->d_status, 0, sizeof(int32_t), s) == hipSuccess;
                good = good && l.dev->launch_decode(l.d_in, bn, l.d_out, l.dev->d_status, s) == TSQA_OK;
                if (good) {
                    (void)hipMemcpyAsync(l.h_status, l.dev->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s);
                    good = hipEventRecord(l.ev, s) == hipSuccess;
                }
                mk.at("decompress: H2D + kernels issued", feeder);
                flag->set(good);
            });
            {
                std::lock_guard<std::mutex> g(fly_m);
                fly.push_back({lane_i, b0, bn, out_bytes});
            }
            fly_cv.notify_all();
            produced += out_bytes; at += cur; b0 += bn;
        }
        close_drainer();                                              // (drains what is in flight first)
        for (auto& l : lanes_) { (void)hipSetDevice(l.dev->device); (void)hipStreamSynchronize(l.dev->stream); }
        touch.cancel();                                               // nobody may still be touching (or allocating behind) the buffer when it is freed
        if (ok && produced != total) ok = false;
        if (!sink.finish()) ok = false;
        if (!j.outfile) {
            if (ok) { *j.out = sink.mem; *j.szout = (size_t)total; }
            else { free(sink.mem); }
        }
        return ok;
    }

    const bool compress_, verbose_;
    std::vector<Lane> lanes_;                          // lane k works on device k % n_devices_ and is fed by feeder k % n_devices_
    std::vector<std::unique_ptr<Issued>> issued_;      // one per lane
    std::vector<std::unique_ptr<Feeder>> feeders_;     // one per listed device
    uint32_t batch_blocks_ = 512, file_batch_blocks_ = 64, n_devices_ = 1;
    std::thread thread_;
    std::mutex m_;
    std::condition_variable cv_, idle_cv_;
    std::deque<Job> q_;
    uint32_t next_id_ = 1;          // turbosqueeze.h: maxjobid starts at 1
    int inflight_ = 0;
    bool exit_ = false;
};

Scheduler* make_scheduler(bool compress, bool verbose)
{
    Scheduler* s = new Scheduler(compress, verbose);
    if (!s->start()) {
        complain_once("no usable gfx950 device (or HIP initialisation failed)");
        delete s;
        return nullptr;
    }
    return s;
}

uint32_t submit_job(Scheduler* s, uint8_t* in, size_t szin, bool infile, uint8_t** out, size_t* szout, bool outfile,
                    bool ext, std::function<void(uint32_t, bool)> done, std::function<void(uint32_t, double)> progress)
{
    // tsq_threads.cpp:296-306,415-418: argument failures report through the callback with job id 0
    if (!s || !in || !out || (!outfile && !szout) || (!infile && szin == 0) || (outfile && !*out)) {
        if (done) done(0, false);
        return 0;
    }
    Job j;
    j.in = in; j.szin = szin; j.infile = infile; j.out = out; j.szout = szout; j.outfile = outfile; j.ext = ext;
    if (outfile) j.out_path = reinterpret_cast<const char*>(*out);
    j.done = std::move(done); j.progress = std::move(progress);
    return s->submit(std::move(j));
}

bool run_sync(Scheduler* s, uint8_t* in, size_t szin, bool infile, uint8_t** out, size_t* szout, bool outfile, bool ext)
{
    if (!s || !in || !out || (!infile && szin == 0)) return false;     // tsq_threads.cpp:415-418
    std::mutex m; std::condition_variable cv; bool finished = false, result = false;
    uint32_t id = submit_job(s, in, szin, infile, out, szout, outfile, ext,
                             [&](uint32_t, bool ok) { std::lock_guard<std::mutex> g(m); result = ok; finished = true; cv.notify_all(); },
                             nullptr);
    if (id == 0) return false;
    std::unique_lock<std::mutex> g(m);                                 // tsq_threads.cpp:435-438
    cv.wait(g, [&] { return finished; });
    return result;
}

// ---- single-block codec used by tsqEncode / tsqDecode ----
struct BlockCodec {
    std::mutex m;
    Lane lane;
    bool tried = false, ready = false;
    bool ensure() {
        if (tried) return ready;
        tried = true;
        ready = lane.init(device_list()[0]) && lane.reserve(kBlockSize + kHalo + kSlotSize, kSlotSize + 256, 1, true, true);
        if (!ready) { complain_once("no usable gfx950 device (or HIP initialisation failed)"); }
        return ready;
    }
};
BlockCodec& block_codec() { static BlockCodec* c = new BlockCodec(); return *c; }

}  // namespace

// =============================================================================================
// extern "C" -- the reference's names
// =============================================================================================

extern "C" struct TSQCompressionContext* tsqAllocateContext(void)
{
    // tsq_context.cpp:56-74: a 128-byte aligned 256 KiB table the caller may touch
    auto* c = static_cast<TSQCompressionContext*>(malloc(sizeof(TSQCompressionContext)));
    if (!c) return nullptr;
    c->refhash = static_cast<uint16_t*>(aligned_alloc(128, TSQ_HASH_SZ));
    if (!c->refhash) { free(c); return nullptr; }
    return c;
}

extern "C" void tsqDeallocateContext(struct TSQCompressionContext* ctx)
{
    if (!ctx) return;
    free(ctx->refhash);
    free(ctx);
}

extern "C" void tsqInit(struct TSQCompressionContext* ctx)
{
    if (ctx && ctx->refhash) memset(ctx->refhash, 0, TSQ_HASH_SZ);    // tsq_context.cpp:77-80
}

static std::atomic<int> g_lookahead_state{0};
extern "C" int tsqa_encode_lookahead_state(void) { return g_lookahead_state.load(); }

extern "C" void tsqEncode(struct TSQCompressionContext* ctx, uint8_t* inputBlock, uint8_t* outputBlock,
                          uint32_t* outputSize, uint32_t inputSize, uint32_t withExtensions)
{
    (void)ctx;   // the device kernel zeroes and owns its table; the host table is API compatibility only
    if (outputSize) *outputSize = 0;
    if (!inputBlock || !outputBlock || !outputSize || inputSize == 0 || inputSize > kBlockSize) return;
    BlockCodec& bc = block_codec();
    std::lock_guard<std::mutex> g(bc.m);
    if (!bc.ensure()) return;
    Lane& l = bc.lane;
    (void)hipSetDevice(l.dev->device);
    hipStream_t s = l.dev->stream;
    memcpy(l.h_in, inputBlock, inputSize);
    // The reference's encoder reads up to ~67 bytes past inputBlock[inputSize-1] (tsq_encode.cpp:74,108,126,162) and its
    // own scheduler relies on that: a worker gets a pointer into the caller's contiguous buffer, so block k's look-ahead
    // sees block k+1 (tsq_threads.cpp:109).  The same bytes are taken here, as far as they are readable: the copy goes
    // through process_vm_readv on this process, which stops at an unmapped page instead of faulting; what cannot be
    // read is seen as zeros (the canonical conditions after the last block).
    size_t halo = 0;
    if (getenv("TSQ_AMD_ENCODE_NO_LOOKAHEAD")) g_lookahead_state = 3;
    else {
        struct iovec to = { l.h_in + inputSize, kHalo }, from = { inputBlock + inputSize, kHalo };
        ssize_t got = process_vm_readv(getpid(), &to, 1, &from, 1, 0);
        if (got > 0) { halo = (size_t)got; g_lookahead_state = 1; }
        else if (got < 0 && errno == EFAULT) g_lookahead_state = 1;       // (the very next byte is unmapped: nothing to read, nothing refused)
        else if (got < 0 && (errno == EPERM || errno == ENOSYS)) {
            g_lookahead_state = 2;
            // a seccomp profile or a sandbox refuses the call: the look-ahead is then ALWAYS seen as zeros, and a loop over the blocks
            // of one buffer no longer gives the container's streams at block edges.  Said once, not silently.
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "turbosqueeze_amd: process_vm_readv is not permitted here (%s): tsqEncode sees zeros behind every block "
                                "(streams stay valid; they differ from the reference's at block edges of one buffer)\n", strerror(errno));
        }
    }
    if (hipMemcpyAsync(l.d_in, l.h_in, inputSize + halo, hipMemcpyHostToDevice, s) != hipSuccess) return;
    (void)hipMemsetAsync(l.dev->d_status, 0, sizeof(int32_t), s);
    if (l.dev->launch_encode(l.d_in, inputSize, inputSize + halo, withExtensions, l.dev->d_status, s) != TSQA_OK) return;
    *l.h_size = 0;
    (void)hipMemcpyAsync(l.h_size, l.dev->sizes, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(l.h_status, l.dev->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess || *l.h_status != 0) return;
    uint32_t sz = (uint32_t)(*l.h_size & 0xFFFFFFFFu);
    if (sz > kSlotSize) return;
    if (hipMemcpyAsync(l.h_out, l.dev->slots, sz, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return;
    memcpy(outputBlock, l.h_out, sz);
    *outputSize = sz;
}

extern "C" void tsqDecode(uint8_t* inputBlock, uint8_t* outputBlock, uint32_t* outputSize, uint32_t inputSize,
                          uint32_t withExtensions)
{
    if (outputSize) *outputSize = 0;
    if (!inputBlock || !outputBlock || !outputSize || inputSize < 3 || inputSize > kSlotSize) return;
    uint32_t usize = (uint32_t)inputBlock[0] | ((uint32_t)inputBlock[1] << 8) | ((uint32_t)inputBlock[2] << 16);
    if (usize > kBlockSize) return;                                    // tsq_decode.cpp:53,146
    BlockCodec& bc = block_codec();
    std::lock_guard<std::mutex> g(bc.m);
    if (!bc.ensure()) return;
    Lane& l = bc.lane;
    (void)hipSetDevice(l.dev->device);
    hipStream_t s = l.dev->stream;
    if (l.dev->reserve(1, false, false) != TSQA_OK) return;
    memcpy(l.h_in, inputBlock, inputSize);
    FrameInfo& fi = l.h_frames[0];
    fi.stream_at = 0; fi.out_at = 0; fi.stream_len = inputSize; fi.ext = withExtensions ? 1u : 0u; fi.out_len = usize; fi.pad = 0;
    if (hipMemcpyAsync(l.d_in, l.h_in, inputSize, hipMemcpyHostToDevice, s) != hipSuccess) return;
    (void)hipMemcpyAsync(l.dev->frames, l.h_frames, sizeof(FrameInfo), hipMemcpyHostToDevice, s);
    (void)hipMemsetAsync(l.dev->d_status, 0, sizeof(int32_t), s);
    if (l.dev->launch_decode(l.d_in, 1, l.d_out, l.dev->d_status, s) != TSQA_OK) return;
    (void)hipMemcpyAsync(l.h_status, l.dev->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return;
    if (*l.h_status == tsq::kErrStall) {          // (a sibling workgroup did not get onto the GPU in time: once more on one workgroup)
        (void)hipMemsetAsync(l.dev->d_status, 0, sizeof(int32_t), s);
        if (l.dev->launch_decode(l.d_in, 1, l.d_out, l.dev->d_status, s, 4) != TSQA_OK) return;
        (void)hipMemcpyAsync(l.h_status, l.dev->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) return;
    }
    if (*l.h_status != 0) return;
    (void)hipMemcpyAsync(l.h_out, l.d_out, usize, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return;
    memcpy(outputBlock, l.h_out, usize);
    *outputSize = usize;
}

// ---- _MT contexts ----

extern "C" struct TSQCompressionContext_MT* tsqAllocateContextCompression_MT(bool verbose)
{
    Scheduler* s = make_scheduler(true, verbose);
    if (!s) return nullptr;
    auto* c = new TSQCompressionContext_MT{s->n_cus(), s};
    return c;
}

extern "C" void tsqDeallocateContextCompression_MT(struct TSQCompressionContext_MT* ctx)
{
    if (!ctx) return;
    auto* s = static_cast<Scheduler*>(ctx->impl);
    s->stop();
    delete s;
    delete ctx;
}

extern "C" struct TSQDecompressionContext_MT* tsqAllocateContextDecompression_MT(bool verbose)
{
    Scheduler* s = make_scheduler(false, verbose);
    if (!s) return nullptr;
    auto* c = new TSQDecompressionContext_MT{s->n_cus(), s};
    return c;
}

extern "C" void tsqDeallocateContextDecompression_MT(struct TSQDecompressionContext_MT* ctx)
{
    if (!ctx) return;
    auto* s = static_cast<Scheduler*>(ctx->impl);
    s->stop();
    delete s;
    delete ctx;
}

extern "C" bool tsqCompress_MT(struct TSQCompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile, uint8_t** out,
                               size_t* szout, bool outfile, bool useextensions, uint32_t level)
{
    (void)level;                                                       // ignored, as tsq_threads.cpp:98
    if (!ctx) return false;
    return run_sync(static_cast<Scheduler*>(ctx->impl), in, szin, infile, out, szout, outfile, useextensions);
}

extern "C" bool tsqDecompress_MT(struct TSQDecompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile, uint8_t** out,
                                 size_t* szout, bool outfile)
{
    if (!ctx) return false;
    return run_sync(static_cast<Scheduler*>(ctx->impl), in, szin, infile, out, szout, outfile, false);
}

extern "C" uint32_t tsqCompressAsync_MT(TSQCompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile, uint8_t** out,
                                        size_t* szout, bool outfile, bool useextensions, uint32_t level,
                                        std::function<void(uint32_t, bool)> done, std::function<void(uint32_t, double)> progress)
{
    (void)level;
    return submit_job(ctx ? static_cast<Scheduler*>(ctx->impl) : nullptr, in, szin, infile, out, szout, outfile, useextensions,
                      std::move(done), std::move(progress));
}

extern "C" uint32_t tsqDecompressAsync_MT(TSQDecompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile, uint8_t** out,
                                          size_t* szout, bool outfile, std::function<void(uint32_t, bool)> done,
                                          std::function<void(uint32_t, double)> progress)
{
    return submit_job(ctx ? static_cast<Scheduler*>(ctx->impl) : nullptr, in, szin, infile, out, szout, outfile, false,
                      std::move(done), std::move(progress));
}

extern "C" uint32_t tsqa_compress_async_cb(struct TSQCompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile, uint8_t** out,
                                           size_t* szout, bool outfile, bool useextensions, uint32_t level, tsqa_done_fn done,
                                           tsqa_progress_fn progress, void* user)
{
    std::function<void(uint32_t, bool)> d;
    std::function<void(uint32_t, double)> p;
    if (done) d = [done, user](uint32_t id, bool ok) { done(id, ok, user); };
    if (progress) p = [progress, user](uint32_t id, double f) { progress(id, f, user); };
    return tsqCompressAsync_MT(ctx, in, szin, infile, out, szout, outfile, useextensions, level, std::move(d), std::move(p));
}

extern "C" uint32_t tsqa_decompress_async_cb(struct TSQDecompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile,
                                             uint8_t** out, size_t* szout, bool outfile, tsqa_done_fn done,
                                             tsqa_progress_fn progress, void* user)
{
    std::function<void(uint32_t, bool)> d;
    std::function<void(uint32_t, double)> p;
    if (done) d = [done, user](uint32_t id, bool ok) { done(id, ok, user); };
    if (progress) p = [progress, user](uint32_t id, double f) { progress(id, f, user); };
    return tsqDecompressAsync_MT(ctx, in, szin, infile, out, szout, outfile, std::move(d), std::move(p));
}

// ---- FILE* to FILE* (turbosqueeze.cpp:48-147): same scheduler, streams supplied by the caller ----

namespace {
bool slurp(FILE* f, std::vector<uint8_t>& buf)
{
    if (!f) return false;
    fseek(f, 0, SEEK_END); long s = ftell(f); fseek(f, 0, SEEK_SET);
    if (s < 0) return false;
    buf.resize((size_t)s);
    return s == 0 || fread(buf.data(), 1, (size_t)s, f) == (size_t)s;
}
}  // namespace

extern "C" void tsqCompress(FILE* in, FILE* out, bool useextensions, uint32_t level)
{
    (void)level;                                                       // turbosqueeze.cpp:48
    std::vector<uint8_t> buf;
    if (!out || !slurp(in, buf)) return;
    if (buf.empty()) {                                                 // header only, as turbosqueeze.cpp:64-67 with n_blocks = 0
        uint8_t header[16] = {'T', 'S', 'Q', '1'};
        fwrite(header, 1, 16, out);
        return;
    }
    TSQCompressionContext_MT* ctx = tsqAllocateContextCompression_MT(false);
    if (!ctx) return;
    uint8_t* blob = nullptr; size_t sz = 0;
    if (tsqCompress_MT(ctx, buf.data(), buf.size(), false, &blob, &sz, false, useextensions, 0)) {
        fwrite(blob, 1, sz, out);
        free(blob);
    }
    tsqDeallocateContextCompression_MT(ctx);
}

extern "C" void tsqDecompress(FILE* in, FILE* out)
{
    std::vector<uint8_t> buf;
    if (!out || !slurp(in, buf) || buf.size() < 16) return;            // turbosqueeze.cpp:106-117: silent return
    TSQDecompressionContext_MT* ctx = tsqAllocateContextDecompression_MT(false);
    if (!ctx) return;
    uint8_t* blob = nullptr; size_t sz = 0;
    if (tsqDecompress_MT(ctx, buf.data(), buf.size(), false, &blob, &sz, false)) {
        fwrite(blob, 1, sz, out);
        free(blob);
    }
    tsqDeallocateContextDecompression_MT(ctx);
}
