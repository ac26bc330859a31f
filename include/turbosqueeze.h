/*
 * turbosqueeze.h -- source-compatible C++ face of libturbosqueeze_amd.so.
 *
 * A program written against the reference's turbosqueeze.h (the 13 functions of
 * /root/reference/turbosqueeze.h:441-674) compiles against this header and links against
 * libturbosqueeze_amd.so unchanged: same names, same argument meaning, same ownership rules.
 * The codec itself runs as HIP kernels on an MI355X; see include/turbosqueeze_amd.h for the plain
 * C ABI underneath and for the device-resident entry points.
 *
 * What callers touch is kept: TSQCompressionContext::refhash (test/test.cpp:42) and the leading
 * num_cores field of both _MT contexts (turbosqueeze.h:343,404).  The reference also shows its
 * thread-pool records (TSQBuffer, TSQWorker, TSQJob; turbosqueeze.h:81-316) in the header.  No entry
 * point takes or returns them and the HIP stream scheduler has no use for them, but a program that
 * names them still compiles: they are declared below with the reference's member names and types.
 */
#pragma once

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <vector>

#include "turbosqueeze_amd.h"

/* ---- records of the reference's thread pool (turbosqueeze.h:81-106,129-188,215-316).  Never created or
 *      consumed by this library; present for source compatibility only. ---- */
class TSQJob;
struct TSQBuffer {                       /* one slot of a worker's input or output ring */
    uint8_t* buffer = nullptr;
    uint8_t* filebuffer = nullptr;
    TSQJob* job = nullptr;
    uint32_t size = 0, ext = 0, compression_level = 0;
};
struct TSQWorker {                       /* a worker's two rings and their hand-off counters */
    std::vector<TSQBuffer> inputs;
    uint32_t n_inputs = 0;
    volatile uint64_t currentReadInput = 0, currentWorkInput = 0;
    std::mutex input_mtx;
    std::condition_variable input_cv;
    std::vector<TSQBuffer> outputs;
    uint32_t n_outputs = 0;
    volatile uint64_t currentWorkOutput = 0, currentWriteOutput = 0;
    std::mutex output_mtx;
    std::condition_variable output_cv;
    uint32_t blocksPerWorker = 0;
};
class TSQJob {                           /* one queued compress / decompress request */
public:
    ~TSQJob() {
        if (input_file && input_stream) fclose(input_stream);
        if (output_file && output_stream) fclose(output_stream);
    }
    uint8_t* input = nullptr;
    uint64_t size = 0;
    bool input_file = false;
    uint32_t jobid = 0;
    bool use_extensions = false;
    uint32_t compression_level = 0;
    FILE* input_stream = nullptr;
    uint64_t input_size = 0, start_block = 0, n_blocks = 0;
    uint8_t* output = nullptr;
    uint64_t outsize = 0;
    bool output_file = false;
    FILE* output_stream = nullptr;
    bool error_occurred = false;
    std::function<void(uint32_t jobid, bool)> completion_cb;
    std::function<void(uint32_t jobid, double)> progress_cb;
};

/* Both _MT contexts: `num_cores` is what the reference calls its worker count; here it is the
 * number of compute units of the device the context drives.  Everything else is private. */
struct TSQCompressionContext_MT {
    uint32_t num_cores;
    void* impl;
};
struct TSQDecompressionContext_MT {
    uint32_t num_cores;
    void* impl;
};

extern "C" {

/* turbosqueeze.h:543-544.  Jobs run FIFO; `done` fires once, `progress` once per block in block
 * order, both on the library's scheduler thread; *out / *szout are valid from `done` onward.
 * Returns the job id (>= 1), or 0 after calling done(0, false). */
uint32_t tsqCompressAsync_MT(TSQCompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile,
                             uint8_t** out, size_t* szout, bool outfile, bool useextensions, uint32_t level,
                             std::function<void(uint32_t jobid, bool)> user_completion_cb,
                             std::function<void(uint32_t jobid, double)> user_progress_cb);

/* turbosqueeze.h:615-616 */
uint32_t tsqDecompressAsync_MT(TSQDecompressionContext_MT* ctx, uint8_t* in, size_t szin, bool infile,
                               uint8_t** out, size_t* szout, bool outfile,
                               std::function<void(uint32_t jobid, bool)> user_completion_cb,
                               std::function<void(uint32_t jobid, double)> user_progress_cb);
}
