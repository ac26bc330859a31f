// tsq_launch.cuh -- kernel selection and launch for the device context.
//
// The product library carries three kernel families: the staged encoder (tsq_enc_stage.cuh: fourteen working wavefronts per block, twelve in the lean layout; standard and lean
// layouts, with and without extensions), the byte-lane decoder (tsq_dec_sym.cuh, tsq_dec_duo.cuh) and the serial correctness
// baselines (tsq_serial.cuh, variant 1).  The previous round's production encoder (ab/tsq_enc_stage_r05.cuh, encoder variant 5) is
// compiled only into the A/B library (`make ab`, -DTSQ_AB_VARIANTS), which tests/test_gpu_parity.py holds against the same oracle.
#pragma once

#include <atomic>

#include "tsq_common.cuh"
#include "tsq_emit.cuh"
#include "tsq_internal.h"
#include "tsq_serial.cuh"
#include "tsq_dec_sym.cuh"
#include "tsq_dec_duo.cuh"
#include "tsq_enc_stage.cuh"
#ifdef TSQ_AB_VARIANTS
#include "ab/tsq_enc_stage_r05.cuh"
#endif

namespace tsq {

// the dynamic-LDS limit is a per-device attribute of a kernel function: raised once per device the process uses
template <size_t N>
inline int raise_lds_limit(tsqa_ctx* c, std::atomic<uint64_t>& done, const void* const (&fns)[N], const uint32_t (&bytes)[N])
{
    const uint64_t dev_bit = 1ull << (c->device & 63);
    if (done.load() & dev_bit) return 0;
    for (size_t k = 0; k < N; ++k)
        if (hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes[k]) != hipSuccess) {
            c->set_error("cannot reserve %u B of LDS", bytes[k]);
            return TSQA_ERR_HIP;
        }
    done.fetch_or(dev_bit);
    return 0;
}

#define TSQ_LAUNCH_ENC(KERNEL, THREADS, LDS)                                                                                        \
    hipLaunchKernelGGL((KERNEL), dim3(nb), dim3(THREADS), (LDS), s, in, (uint64_t)n, (uint64_t)readable, (uint64_t)stride, slots, sizes, c->tables, status)

// Encode nb = ceil(n / 4 MiB) blocks; block b is read at in + b * stride, streams land in slots[b], sizes in sizes[b].
inline int launch_encode_kernels(tsqa_ctx* c, const uint8_t* in, size_t n, size_t readable, size_t stride, uint32_t ext,
                                 uint8_t* slots, uint32_t* sizes, int32_t* status, hipStream_t s)
{
    const uint32_t nb = (uint32_t)((n + kBlockSize - 1) / kBlockSize);
    static std::atomic<uint64_t> attr_devices{0};
    {
        const void* const fns[4] = {reinterpret_cast<const void*>(enc_stage_kernel<true, true>), reinterpret_cast<const void*>(enc_stage_kernel<false, true>),
                                    reinterpret_cast<const void*>(enc_stage_kernel<true, false>), reinterpret_cast<const void*>(enc_stage_kernel<false, false>)};
        const uint32_t bytes[4] = {StageCfgT<true>::total, StageCfgT<true>::total, StageCfgT<false>::total, StageCfgT<false>::total};
        if (int rc = raise_lds_limit(c, attr_devices, fns, bytes)) return rc;
    }
    const int v = c->enc_variant;
    if (v == 1) {                       // one lane walks the block: the correctness baseline
        if (ext) TSQ_LAUNCH_ENC(enc_serial_kernel<true>, 64, 0);
        else     TSQ_LAUNCH_ENC(enc_serial_kernel<false>, 64, 0);
        return 0;
    }
#ifdef TSQ_AB_VARIANTS
    if (v == 5) {                       // round 5's production encoder, frozen (ab/tsq_enc_stage_r05.cuh)
        static std::atomic<uint64_t> ab_devices{0};
        const void* const fns[4] = {reinterpret_cast<const void*>(r05::enc_stage_kernel<true, true>), reinterpret_cast<const void*>(r05::enc_stage_kernel<false, true>),
                                    reinterpret_cast<const void*>(r05::enc_stage_kernel<true, false>), reinterpret_cast<const void*>(r05::enc_stage_kernel<false, false>)};
        const uint32_t bytes[4] = {r05::StageCfgT<true>::total, r05::StageCfgT<true>::total, r05::StageCfgT<false>::total, r05::StageCfgT<false>::total};
        if (int rc = raise_lds_limit(c, ab_devices, fns, bytes)) return rc;
        if (nb > (uint32_t)c->n_cus) {
            if (ext) TSQ_LAUNCH_ENC((r05::enc_stage_kernel<true, false>), r05::StageCfgT<false>::THREADS_LEAN, r05::StageCfgT<false>::total);
            else     TSQ_LAUNCH_ENC((r05::enc_stage_kernel<false, false>), r05::StageCfgT<false>::THREADS_LEAN, r05::StageCfgT<false>::total);
        } else {
            if (ext) TSQ_LAUNCH_ENC((r05::enc_stage_kernel<true, true>), r05::StageCfgT<true>::THREADS, r05::StageCfgT<true>::total);
            else     TSQ_LAUNCH_ENC((r05::enc_stage_kernel<false, true>), r05::StageCfgT<true>::THREADS, r05::StageCfgT<true>::total);
        }
        return 0;
    }
#else
    if (v == 5) { c->set_error("kernel variant %d lives in the A/B library only (make ab)", v); return TSQA_ERR_ARG; }
#endif
    if (v >= 2 && v <= 4) { c->set_error("kernel variant %d is not built", v); return TSQA_ERR_ARG; }
    // staged pipeline (tsq_enc_stage.cuh).  More blocks than CUs: the lean layout (no input window in LDS, candidate
    // bytes from L2) lets several blocks share a CU; each is a little slower, together they are faster.
    const bool lean = v == 6 || (v == 0 && nb > (uint32_t)c->n_cus);
    if (lean) {
        if (ext) TSQ_LAUNCH_ENC((enc_stage_kernel<true, false>), StageCfgT<false>::THREADS_LEAN, StageCfgT<false>::total);
        else     TSQ_LAUNCH_ENC((enc_stage_kernel<false, false>), StageCfgT<false>::THREADS_LEAN, StageCfgT<false>::total);
    } else {
        if (ext) TSQ_LAUNCH_ENC((enc_stage_kernel<true, true>), StageCfgT<true>::THREADS, StageCfgT<true>::total);
        else     TSQ_LAUNCH_ENC((enc_stage_kernel<false, true>), StageCfgT<true>::THREADS, StageCfgT<true>::total);
    }
    return 0;
}
#undef TSQ_LAUNCH_ENC

inline int launch_decode_kernels(tsqa_ctx* c, const uint8_t* container, const FrameInfo* frames, uint32_t n_blocks, uint8_t* out,
                                 int32_t* status, hipStream_t s, int variant = -1)
{
    static std::atomic<uint64_t> attr_devices{0};
    {
        const void* const fns[1] = {reinterpret_cast<const void*>(dec_sym_kernel)};
        const uint32_t bytes[1] = {SymLds::total};
        if (int rc = raise_lds_limit(c, attr_devices, fns, bytes)) return rc;
    }
    const int v = variant >= 0 ? variant : c->dec_variant;
    if (v == 1) { hipLaunchKernelGGL(dec_serial_kernel, dim3(n_blocks), dim3(64), 0, s, container, frames, out, status); return 0; }
    if (v == 8 || v == 9) { c->set_error("kernel variant %d is not built", v); return TSQA_ERR_ARG; }
    // Few blocks (every GPU of a multi-GPU job on enwik9): several workgroups per block on different CUs of one XCD -- the block's
    // copy chain on one, its parse on one (at most half as many blocks as CUs) or two (at most a third) (tsq_dec_duo.cuh).
    auto launch_multi = [&](const FrameInfo* fr, uint32_t nblk, bool three) -> int {
        static std::atomic<uint64_t> duo_devices{0};
        const void* const fns[2] = {reinterpret_cast<const void*>(dec_duo_kernel<1>), reinterpret_cast<const void*>(dec_duo_kernel<2>)};
        const uint32_t lds_bytes = DuoCopyLds::total > SymLds::total ? DuoCopyLds::total : SymLds::total;
        const uint32_t bytes[2] = {lds_bytes, lds_bytes};
        if (int rc = raise_lds_limit(c, duo_devices, fns, bytes)) return rc;
        if (int rc = c->reserve_duo(nblk)) return rc;
        if (hipMemsetAsync(c->duo_flags, 0, (size_t)nblk * DuoCfg::FLAG_STRIDE * sizeof(uint32_t), s) != hipSuccess) { c->set_error("hipMemsetAsync failed"); return TSQA_ERR_HIP; }
        const uint32_t groups = (nblk + 7u) / 8u;
        if (three) hipLaunchKernelGGL(dec_duo_kernel<2>, dim3(24u * groups), dim3(SymCfg::T), lds_bytes, s, container, fr, nblk, out, status, c->duo_ring, c->duo_flags, c->decode_wait_limit);
        else hipLaunchKernelGGL(dec_duo_kernel<1>, dim3(16u * groups), dim3(SymCfg::T), lds_bytes, s, container, fr, nblk, out, status, c->duo_ring, c->duo_flags, c->decode_wait_limit);
        return 0;
    };
    const uint32_t cus = (uint32_t)c->n_cus;
    const bool trio = v == 5 || ((v == 0 || v == 3) && 3u * n_blocks <= cus);
    if (trio || v == 3 || v == 6 || (v == 0 && 2u * n_blocks <= cus)) return launch_multi(frames, n_blocks, trio && v != 6);
    // More blocks than CUs, one workgroup per block: the blocks run in rounds of n_cus, and a last round with few blocks would
    // leave most of the chip idle for a whole block latency (321 blocks: 256 + 65).  The blocks of such a partial round get several
    // workgroups each, in a launch of their own behind the full rounds (tsq_threads.cpp:71 deals blocks to workers the same way:
    // whoever is free takes the next).
    if (v == 0 && n_blocks > cus) {
        const uint32_t tail = n_blocks % cus, full = n_blocks - tail;
        if (tail != 0u && 2u * tail <= cus) {
            hipLaunchKernelGGL(dec_sym_kernel, dim3(full), dim3(SymCfg::T), SymLds::total, s, container, frames, out, status);
            return launch_multi(frames + full, tail, 3u * tail <= cus);
        }
    }
    // one workgroup per block at any block count: with more blocks than CUs the blocks simply queue (the decoder needs its 150 KB
    // of LDS; the two-per-CU layout of the byte-granular decoder was 1.85x slower per byte, tools/config5_sweep.py)
    hipLaunchKernelGGL(dec_sym_kernel, dim3(n_blocks), dim3(SymCfg::T), SymLds::total, s, container, frames, out, status);
    return 0;
}

}  // namespace tsq
