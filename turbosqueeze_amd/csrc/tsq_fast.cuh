// tsq_fast.cuh -- the production encode/decode kernels (variant 0).
#pragma once

#include <atomic>

#include "tsq_common.cuh"
#include "tsq_emit.cuh"
#include "tsq_internal.h"
#include "tsq_serial.cuh"
#include "tsq_dec_fast.cuh"
#include "tsq_dec_ring.cuh"
#include "tsq_enc_fast.cuh"
#include "tsq_enc_orbit.cuh"
#include "tsq_enc_pipe.cuh"
#include "tsq_enc_tile.cuh"
#include "tsq_enc_stage.cuh"

namespace tsq {

constexpr uint32_t kEncMaxLds = StageCfg::total > TileCfg::total ? StageCfg::total : TileCfg::total;

inline int launch_encode_fast(tsqa_ctx* c, const uint8_t* in, size_t n, size_t readable, uint32_t ext, int32_t* status, hipStream_t s)
{
    const uint32_t nb = (uint32_t)((n + kBlockSize - 1) / kBlockSize);
    int rc = c->reserve(nb, true);
    if (rc) return rc;
    // the dynamic-LDS limit is a per-device attribute of the function: once per device the process uses
    static std::atomic<uint64_t> attr_devices{0};
    const uint64_t dev_bit = 1ull << (c->device & 63);
    if (!(attr_devices.load() & dev_bit)) {
        const void* fns[12] = {reinterpret_cast<const void*>(enc_stage_kernel<true, true>), reinterpret_cast<const void*>(enc_stage_kernel<false, true>),
                              reinterpret_cast<const void*>(enc_stage_kernel<true, false>), reinterpret_cast<const void*>(enc_stage_kernel<false, false>),
                              reinterpret_cast<const void*>(enc_tile_kernel<true>), reinterpret_cast<const void*>(enc_tile_kernel<false>),
                              reinterpret_cast<const void*>(enc_fast_kernel<true>), reinterpret_cast<const void*>(enc_fast_kernel<false>),
                              reinterpret_cast<const void*>(enc_orbit_kernel<true>), reinterpret_cast<const void*>(enc_orbit_kernel<false>),
                              reinterpret_cast<const void*>(enc_pipe_kernel<true>), reinterpret_cast<const void*>(enc_pipe_kernel<false>)};
        for (const void* fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEncMaxLds) != hipSuccess) {
                c->set_error("cannot reserve %u B of LDS", kEncMaxLds);
                return TSQA_ERR_HIP;
            }
        attr_devices.fetch_or(dev_bit);
    }
    if (c->enc_variant == 2) {          // the windowed scalar walk (kept for A/B)
        if (ext) hipLaunchKernelGGL(enc_fast_kernel<true>, dim3(nb), dim3(64), kEncLds, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
        else     hipLaunchKernelGGL(enc_fast_kernel<false>, dim3(nb), dim3(64), kEncLds, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
    } else if (c->enc_variant == 3) {   // single-wave orbit encoder (kept for A/B)
        if (ext) hipLaunchKernelGGL(enc_orbit_kernel<true>, dim3(nb), dim3(64), kOrbLds, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
        else     hipLaunchKernelGGL(enc_orbit_kernel<false>, dim3(nb), dim3(64), kOrbLds, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
    } else if (c->enc_variant == 4) {   // two-wave pipeline: parser + builder (kept for A/B)
        if (ext) hipLaunchKernelGGL(enc_pipe_kernel<true>, dim3(nb), dim3(128), PipeCfg::total, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
        else     hipLaunchKernelGGL(enc_pipe_kernel<false>, dim3(nb), dim3(128), PipeCfg::total, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
    } else if (c->enc_variant == 5) {   // three-wave tile pipeline: front + parser + builder (kept for A/B)
        if (ext) hipLaunchKernelGGL(enc_tile_kernel<true>, dim3(nb), dim3(192), TileCfg::total, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
        else     hipLaunchKernelGGL(enc_tile_kernel<false>, dim3(nb), dim3(192), TileCfg::total, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
    } else {                            // five-wave staged pipeline: scan + match + orbit + parser + builder
        // More blocks than CUs: the lean layout (no input window in LDS, candidate bytes from L2) lets two blocks share a CU;
        // each is a little slower, together they are faster than one after the other.
        const bool lean = c->enc_variant == 6 || (c->enc_variant == 0 && nb > (uint32_t)c->n_cus);
        if (lean) {
            if (ext) hipLaunchKernelGGL((enc_stage_kernel<true, false>), dim3(nb), dim3(320), StageCfg::total_lean, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
            else     hipLaunchKernelGGL((enc_stage_kernel<false, false>), dim3(nb), dim3(320), StageCfg::total_lean, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
        } else {
            if (ext) hipLaunchKernelGGL((enc_stage_kernel<true, true>), dim3(nb), dim3(320), StageCfg::total, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
            else     hipLaunchKernelGGL((enc_stage_kernel<false, true>), dim3(nb), dim3(320), StageCfg::total, s, in, (uint64_t)n, (uint64_t)readable, c->slots, c->sizes, c->tables, status);
        }
    }
    return 0;
}

inline int launch_decode_fast(tsqa_ctx* c, const uint8_t* container, uint32_t n_blocks, uint8_t* out, int32_t* status, hipStream_t s)
{
    // the dynamic-LDS limit is a per-device attribute of the function: once per device the process uses
    static std::atomic<uint64_t> attr_devices{0};
    const uint64_t dev_bit = 1ull << (c->device & 63);
    if (!(attr_devices.load() & dev_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DecLds::total) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(dec_ring_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RingLds::total) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(dec_ring_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LeanLds::total) != hipSuccess) {
            c->set_error("cannot reserve LDS for the decoder");
            return TSQA_ERR_HIP;
        }
        attr_devices.fetch_or(dev_bit);
    }
    if (c->dec_variant == 2)   // the first parallel decoder (history gathered from L2), kept for A/B
        hipLaunchKernelGGL(dec_fast_kernel, dim3(n_blocks), dim3(DecCfg::T), DecLds::total, s, container, c->frames, out, status);
    else
        // more blocks than CUs: the lean layout lets two blocks share a CU (variant 6 forces it, 7 never uses it)
        if (c->dec_variant == 6 || (c->dec_variant == 0 && n_blocks > (uint32_t)c->n_cus))
            hipLaunchKernelGGL(dec_ring_kernel<false>, dim3(n_blocks), dim3(LeanCfg::T), LeanLds::total, s, container, c->frames, out, status);
        else
            hipLaunchKernelGGL(dec_ring_kernel<true>, dim3(n_blocks), dim3(RingCfg::T), RingLds::total, s, container, c->frames, out, status);
    return 0;
}

}  // namespace tsq
