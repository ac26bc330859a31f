// Does IB_STS.VM_CNT (s_getreg_b32) follow a wavefront's outstanding stores without waiting?  One wavefront issues k scattered 2-byte
// stores, then (a) s_waitcnt vmcnt(0), (b) polls IB_STS until VM_CNT == 0; cycles for both, and the VM_CNT values seen right after issue.
// hipcc --offload-arch=gfx950 -O2 -o ibsts_probe ibsts_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t vm_now() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(x) :: "memory"); return (x & 15u) | ((x >> 18) & 0x30u); }
__global__ void probe(uint16_t* tab, uint32_t* out, int k, int mode)
{
    const uint32_t lane = threadIdx.x;
    uint32_t h = (lane * 2654435761u + blockIdx.x * 40503u) & 0x1FFFFu;
    unsigned long long acc = 0; uint32_t seen_after = 0, polls = 0;
    for (int rep = 0; rep < 64; ++rep) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int j = 0; j < k; ++j) { tab[h] = (uint16_t)(rep + j); h = (h * 1664525u + 1013904223u) & 0x1FFFFu; }
        const uint32_t a = vm_now();
        if (mode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else { while (vm_now() != 0u) { ++polls; } }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc += t1 - t0; seen_after += a;
    }
    if (lane == 0 && blockIdx.x == 0) { out[0] = (uint32_t)(acc / 64); out[1] = seen_after; out[2] = polls; }
}
int main()
{
    uint16_t* tab; uint32_t* out; hipMalloc(&tab, 256 * 262144); hipMalloc(&out, 64);
    for (int k : {1, 2, 4}) for (int mode : {0, 1}) {
        uint32_t h[3];
        hipLaunchKernelGGL(probe, dim3(240), dim3(64), 0, 0, tab, out, k, mode); hipDeviceSynchronize();
        hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
        printf("k=%d %s: %u cycles per round (issue + acknowledgement), VM_CNT right after issue summed over 64 rounds = %u, polls = %u\n", k, mode ? "IB_STS poll   " : "s_waitcnt     ", h[0], h[1], h[2]);
    }
    return 0;
}
