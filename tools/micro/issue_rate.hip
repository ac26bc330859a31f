// Single-wave issue-rate microbenchmarks for gfx950 (what a lone wavefront pays per instruction kind).
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k_bench(unsigned long long* out, int iters, int which)
{
    __shared__ uint32_t lds[512];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0, t1;
    uint32_t s = iters, v = threadIdx.x, w = 0; uint32_t a = 1, b = 2, c = 3, d = 4; uint32_t idx = threadIdx.x; unsigned long long q = iters; unsigned long long m = 0;
    // 0: dependent SALU chain
    if (which == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[0] = t1 - t0;
    }
    // 1: independent SALU (4 chains)
    if (which == 1) {
    
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t") : "+s"(a), "+s"(b), "+s"(c), "+s"(d) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[1] = t1 - t0;
    }
    // 2: dependent VALU chain
    if (which == 2) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP64("v_add_u32 %0, %0, 1\n\t") : "+v"(v) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[2] = t1 - t0;
    }
    // 3: alternating VALU / SALU (independent)
    if (which == 3) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("v_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(v), "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[3] = t1 - t0;
    }
    // 4: v_readlane -> SALU use -> (64 pairs)
    if (which == 4) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("v_readlane_b32 %1, %0, 3\n\ts_add_u32 %2, %2, %1\n\tv_readlane_b32 %1, %0, 5\n\ts_add_u32 %2, %2, %1\n\t") : "+v"(v), "+s"(a), "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[4] = t1 - t0;
    }
    // 5: taken branches: 64 x (s_cmp + s_cbranch to next label)
    if (which == 5) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP64("s_cmp_lg_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") : "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[5] = t1 - t0;
    }
    // 6: not-taken branches
    if (which == 6) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP64("s_cmp_eq_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") : "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[6] = t1 - t0;
    }
    // 7: s_memtime back to back
    if (which == 7) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { REP16(w += (uint32_t)__builtin_amdgcn_s_memtime();) }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[7] = t1 - t0;
    }
    // 8: LDS load round trip (dependent chain of 64)
    if (which == 8) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { REP64(idx = ((volatile uint32_t*)lds)[idx & 255];) }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[8] = t1 - t0;
    }
    // 9: far taken branch (jump over 64 instructions) x 16
    if (which == 9) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("s_cmp_lg_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\t" REP64("s_nop 0\n\t") "1:\n\t") : "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[9] = t1 - t0;
    }
    // 10: 64-bit SALU dependent
    if (which == 10) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP64("s_lshl_b64 %0, %0, 1\n\t") : "+s"(q) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[10] = t1 - t0;
    }
    // 11: v_cmp -> s_and (VALU writes SGPR, SALU reads)
    if (which == 11) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("v_cmp_ne_u32 %1, 0, %0\n\ts_and_b64 %1, %1, exec\n\tv_cmp_ne_u32 %1, 1, %0\n\ts_and_b64 %1, %1, exec\n\t") : "+v"(v), "+s"(m) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[11] = t1 - t0;
    }
    // 12: SALU -> VALU (sgpr operand) -> SALU round trip: s_add ; v_and v,s ; v_readfirstlane ; (x32)
    if (which == 12) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("s_add_u32 %1, %1, 1\n\tv_and_b32 %0, %1, %0\n\tv_readfirstlane_b32 %2, %0\n\ts_add_u32 %1, %1, %2\n\ts_add_u32 %1, %1, 1\n\tv_and_b32 %0, %1, %0\n\tv_readfirstlane_b32 %2, %0\n\ts_add_u32 %1, %1, %2\n\t") : "+v"(v), "+s"(s), "+s"(a) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[12] = t1 - t0;
    }
    // 13: SALU writes sgpr, VALU reads it (no way back): s_add ; v_add v, s, v  (x64 pairs... 32 pairs)
    if (which == 13) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("s_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[13] = t1 - t0;
    }
    // 14: 8 v_writelane (sgpr source) + ds_write, x8
    if (which == 14) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP4("v_writelane_b32 %0, %1, 1\n\tv_writelane_b32 %0, %1, 2\n\tv_writelane_b32 %0, %1, 3\n\tv_writelane_b32 %0, %1, 4\n\tv_writelane_b32 %0, %1, 5\n\tv_writelane_b32 %0, %1, 6\n\tv_writelane_b32 %0, %1, 7\n\tv_writelane_b32 %0, %1, 8\n\tds_write_b32 %2, %0\n\t") : "+v"(v) : "s"(s), "v"((uint32_t)(threadIdx.x * 4)) : "memory"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[14] = t1 - t0;
    }
    // 15: dependent 64-bit bit-scan chain: s_ff1_i32_b64 ; s_lshl_b64 ; s_bcnt1 ; s_flbit
    if (which == 15) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("s_ff1_i32_b64 %1, %0\n\ts_lshl_b64 %0, %0, %1\n\ts_bcnt1_i32_b64 %1, %0\n\ts_flbit_i32_b64 %1, %0\n\t") : "+s"(q), "+s"(a) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[15] = t1 - t0;
    }
    // 16: v_readlane with an SGPR lane select, then use (x32 pairs)
    if (which == 16) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP16("s_and_b32 %2, %2, 63\n\tv_readlane_b32 %1, %0, %2\n\ts_add_u32 %2, %2, %1\n\ts_and_b32 %2, %2, 63\n\tv_readlane_b32 %1, %0, %2\n\ts_add_u32 %2, %2, %1\n\t") : "+v"(v), "+s"(a), "+s"(s) :: "scc"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[16] = t1 - t0;
    }
    // 17: 11 independent LDS loads + one wait (x4)
    if (which == 17) {
    uint32_t r0 = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { asm volatile(REP4("ds_read_b32 %0, %1\n\tds_read_b32 %0, %1 offset:256\n\tds_read_b32 %0, %1 offset:512\n\tds_read_b32 %0, %1 offset:768\n\tds_read_b32 %0, %1\n\tds_read_b32 %0, %1 offset:256\n\tds_read_b32 %0, %1 offset:512\n\tds_read_b32 %0, %1 offset:768\n\tds_read_b32 %0, %1\n\tds_read_b32 %0, %1 offset:256\n\tds_read_b32 %0, %1 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") : "=&v"(r0) : "v"((uint32_t)(threadIdx.x * 4)) : "memory"); }
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[17] = t1 - t0;
    w += r0;
    }
    if (threadIdx.x == 0) out[23] = s + a + b + c + d + v + w + idx + (uint32_t)q + (uint32_t)m;
}

int main()
{
    unsigned long long* d; unsigned long long h[24];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    const int iters = 2000;
    const char* names[18] = {"dependent s_add (per instr)", "independent s_add x4 (per instr)", "dependent v_add (per instr)", "alternating v_add/s_add (per instr)",
                             "v_readlane + dependent s_add (per pair)", "taken short branch (cmp+branch)", "not-taken branch (cmp+branch+nop)", "s_memtime (each)",
                             "dependent LDS load (each)", "taken far branch (cmp+branch over 64 instrs)", "dependent s_lshl_b64 (per instr)", "v_cmp -> s_and_b64 (per pair)",
                             "s_add > v_and(sgpr) > v_readfirstlane > s_add (per round trip)", "s_add > v_add(sgpr) (per pair)", "8 v_writelane + ds_write (per group of 9)",
                             "ff1/lshl/bcnt/flbit b64 chain (per instr)", "v_readlane with SGPR lane + use (per pair)", "11 LDS loads + wait (per group)"};
    const double per[18] = {64, 64, 64, 64, 32, 64, 64, 16, 64, 16, 64, 32, 32, 32, 4, 64, 32, 4};
    for (int k = 0; k < 18; ++k) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, d, iters, k);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, d, iters, k);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-48s %8.2f ticks   (kernel %.3f ms -> %.0f ticks/us)\n", names[k], (double)h[k] / iters / per[k], ms, h[k] / (ms * 1e3));
        fflush(stdout);
    }
    return 0;
}
