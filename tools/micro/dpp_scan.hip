// DPP wave scans on gfx950: check against a serial scan.  hipcc --offload-arch=gfx950 -O2 -o dpp_scan dpp_scan.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define STEP(op, ctrl, rows) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false); v = op(v, t_); }
__device__ uint32_t addf(uint32_t a, uint32_t b) { return a + b; }
__device__ uint32_t maxf(uint32_t a, uint32_t b) { return a > b ? a : b; }
__global__ void k(const uint32_t* in, uint32_t* out)
{
    uint32_t v = in[threadIdx.x];
    STEP(addf, 0x111, 0xF) STEP(addf, 0x112, 0xF) STEP(addf, 0x114, 0xF) STEP(addf, 0x118, 0xF) STEP(addf, 0x142, 0xA) STEP(addf, 0x143, 0xC)
    out[threadIdx.x] = v;
    v = in[threadIdx.x];
    STEP(maxf, 0x111, 0xF) STEP(maxf, 0x112, 0xF) STEP(maxf, 0x114, 0xF) STEP(maxf, 0x118, 0xF) STEP(maxf, 0x142, 0xA) STEP(maxf, 0x143, 0xC)
    out[64 + threadIdx.x] = v;
    out[128 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)in[threadIdx.x], 0x138, 0xF, 0xF, false);
}
int main()
{
    uint32_t h[64], o[192], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (i * 2654435761u >> 20) % 1000;
    hipMalloc(&di, 256); hipMalloc(&dout, 768);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, 768, hipMemcpyDeviceToHost);
    uint32_t s = 0, m = 0; int bad = 0;
    for (int i = 0; i < 64; ++i) {
        s += h[i]; m = h[i] > m ? h[i] : m;
        if (o[i] != s) { if (bad < 8) printf("add lane %d: got %u want %u\n", i, o[i], s); bad++; }
        if (o[64 + i] != m) { if (bad < 8) printf("max lane %d: got %u want %u\n", i, o[64 + i], m); bad++; }
        if (o[128 + i] != (i ? h[i - 1] : 0)) { if (bad < 8) printf("shr lane %d: got %u want %u\n", i, o[128 + i], i ? h[i - 1] : 0); bad++; }
    }
    printf(bad ? "WRONG (%d)\n" : "ALL CORRECT\n", bad);
    return bad != 0;
}
