// What does FETCH_SIZE count for random 2-byte gathers (the encoder's position-table lookups)?  A known number of gathers, each to a
// different, cold cache line of a 2 GiB region, against a streaming 16-byte read of the same region.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/micro/gather_calib      (tools/gather_calib.sh prints the bytes per gather)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void gather2(const uint16_t* t, uint64_t mask, uint32_t per_thread, uint32_t* sink)
{
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        acc += t[x & mask];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ void stream16(const uint4* t, size_t words, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = t[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
int main()
{
    const size_t bytes = size_t(2) << 30;
    uint16_t* t; uint32_t* sink;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(t, 1, bytes);
    (void)hipDeviceSynchronize();
    const uint32_t blocks = 4096, threads = 256, per_thread = 64;          // 67 108 864 gathers over 16 777 216 lines of 128 B
    hipLaunchKernelGGL(gather2, dim3(blocks), dim3(threads), 0, 0, t, (bytes / 2) - 1, per_thread, sink);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4*>(t), bytes / 16, sink);
    (void)hipDeviceSynchronize();
    printf("gathers %llu, streamed bytes %zu\n", (unsigned long long)blocks * threads * per_thread, bytes);
    return 0;
}
