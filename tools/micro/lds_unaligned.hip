// Unaligned LDS accesses on gfx950: are ds_read/ds_write _b32/_b64/_b128 at arbitrary byte addresses correct, and what do
// they cost next to byte accesses?  (hipcc emits them for __builtin_memcpy on LDS pointers: the amdhsa target assumes the
// unaligned access mode.)  Build: hipcc --offload-arch=gfx950 -O2 -o lds_unaligned lds_unaligned.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

constexpr uint32_t N = 32768;

template <int W>
__device__ __forceinline__ void copy_w(uint8_t* dst, const uint8_t* src)
{
    if (W == 1) { dst[0] = src[0]; }
    else if (W == 2) { uint16_t v; __builtin_memcpy(&v, src, 2); __builtin_memcpy(dst, &v, 2); }
    else if (W == 4) { uint32_t v; __builtin_memcpy(&v, src, 4); __builtin_memcpy(dst, &v, 4); }
    else if (W == 8) { uint64_t v; __builtin_memcpy(&v, src, 8); __builtin_memcpy(dst, &v, 8); }
    else { uint4 v; __builtin_memcpy(&v, src, 16); __builtin_memcpy(dst, &v, 16); }
}

// every lane copies W bytes from src_off[lane] to dst_off[lane] inside LDS, `iters` times (offsets advance by a stride)
template <int W>
__global__ void k_copy(const uint32_t* src_off, const uint32_t* dst_off, uint8_t* image, unsigned long long* cycles, int iters, uint32_t step)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * N + 64];
    for (uint32_t i = threadIdx.x; i < 2 * N + 64; i += blockDim.x) lds[i] = (uint8_t)(i * 7u + (i >> 8));
    __syncthreads();
    uint32_t s = src_off[threadIdx.x], d = dst_off[threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; ++k) {
        copy_w<W>(lds + N + d, lds + s);
        s = (s + step) & (N - 1u) & ~0u; d = (d + step) & (N - 32u);
        if (s > N - 16) s -= 16;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
    for (uint32_t i = threadIdx.x; i < 2 * N; i += blockDim.x) image[i] = lds[i];
}

template <int W>
bool run(const char* name, int threads, bool random_offsets, uint32_t align = 1, bool dense = false)
{
    std::vector<uint32_t> so(threads), dof(threads);
    uint32_t x = 12345;
    for (int t = 0; t < threads; ++t) {
        x = x * 1664525u + 1013904223u;
        so[t] = random_offsets ? (x >> 8) % (N - 16) : (uint32_t)t * 17u + 3u;      // arbitrary byte phases either way
        dof[t] = (uint32_t)t * 20u + 1u;                                           // distinct, non-overlapping destinations (20 >= 16)
        if (align > 1) { so[t] &= ~(align - 1u); dof[t] = (uint32_t)t * 32u; }
        if (dense) { so[t] = (uint32_t)t * W; dof[t] = (uint32_t)t * W; }          // consecutive lanes, consecutive items: no bank conflicts
    }
    uint32_t *d_so, *d_do; uint8_t* d_img; unsigned long long* d_cyc;
    hipMalloc(&d_so, threads * 4); hipMalloc(&d_do, threads * 4); hipMalloc(&d_img, 2 * N); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_so, so.data(), threads * 4, hipMemcpyHostToDevice); hipMemcpy(d_do, dof.data(), threads * 4, hipMemcpyHostToDevice);
    // correctness: one iteration
    k_copy<W><<<1, threads>>>(d_so, d_do, d_img, d_cyc, 1, 0);
    std::vector<uint8_t> img(2 * N), want(2 * N);
    hipMemcpy(img.data(), d_img, 2 * N, hipMemcpyDeviceToHost);
    for (uint32_t i = 0; i < 2 * N; ++i) want[i] = (uint8_t)(i * 7u + (i >> 8));
    for (int t = 0; t < threads; ++t) for (int b = 0; b < W; ++b) want[N + dof[t] + b] = (uint8_t)((so[t] + b) * 7u + ((so[t] + b) >> 8));
    const bool ok = memcmp(img.data(), want.data(), 2 * N) == 0;
    // cost
    const int iters = 2000;
    k_copy<W><<<1, threads>>>(d_so, d_do, d_img, d_cyc, iters, 64);   // (a step of 64 keeps every offset's alignment)
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-6s width %2d  align %2u  threads %4d  %-10s  %s   %.1f cycles per copy per workgroup  (%.2f bytes/cycle/CU)\n", name, W, align, threads,
           dense ? "dense" : random_offsets ? "random" : "strided", ok ? "CORRECT" : "WRONG", (double)cyc / iters, (double)W * threads * iters / (double)cyc);
    (void)0;
    hipFree(d_so); hipFree(d_do); hipFree(d_img); hipFree(d_cyc);
    return ok;
}

int main()
{
    bool ok = true;
    for (int threads : {64, 1024}) for (bool rnd : {false, true}) {
        ok &= run<1>("b8", threads, rnd);
        ok &= run<2>("b16", threads, rnd);
        ok &= run<4>("b32", threads, rnd);
        ok &= run<8>("b64", threads, rnd);
        ok &= run<16>("b128", threads, rnd);
    }
    for (bool rnd : {false, true}) {
        ok &= run<2>("b16", 1024, rnd, 2);
        ok &= run<4>("b32", 1024, rnd, 4);
        ok &= run<8>("b64", 1024, rnd, 4);
        ok &= run<8>("b64", 1024, rnd, 8);
        ok &= run<16>("b128", 1024, rnd, 4);
        ok &= run<16>("b128", 1024, rnd, 8);
        ok &= run<16>("b128", 1024, rnd, 16);
    }
    ok &= run<1>("b8", 1024, false, 1, true);
    ok &= run<2>("b16", 1024, false, 2, true);
    ok &= run<4>("b32", 1024, false, 4, true);
    ok &= run<8>("b64", 1024, false, 8, true);
    ok &= run<16>("b128", 1024, false, 16, true);
    printf(ok ? "ALL CORRECT\n" : "FAILURES\n");
    return ok ? 0 : 1;
}
