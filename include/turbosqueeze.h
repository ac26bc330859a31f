/*
 * turbosqueeze.h -- source-compatible C++ face of libturbosqueeze_amd.so.
 *
 * A program written against the reference's turbosqueeze.h (the 13 functions of
 * /root/reference/turbosqueeze.h:441-674) compiles against this header and links against
 * libturbosqueeze_amd.so unchanged: same names, same argument meaning, same ownership rules.
 * The codec itself runs as HIP kernels on an MI355X; see include/turbosqueeze_amd.h for the plain
 * C ABI underneath and for the device-resident entry points.
 *
 * The reference exposes its thread-pool internals (TSQBuffer, TSQWorker, TSQJob) in the header;
 * callers only ever receive context pointers from the library, so those are not reproduced.
 * What callers do touch is kept: TSQCompressionContext::refhash (test/test.cpp:42) and the
 * leading num_cores field of both _MT contexts (turbosqueeze.h:343,404).
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <functional>

#include "turbosqueeze_amd.h"

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
