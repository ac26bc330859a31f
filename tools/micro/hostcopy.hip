// Host <-> device transfer microbenchmarks (what the reference-named host API has to live with).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_memcpy(uint8_t* d, const uint8_t* s, size_t n, int threads)
{
    std::vector<std::thread> th;
    size_t per = (n / threads + 4095) & ~size_t(4095);
    for (int k = 0; k < threads; ++k) { size_t a = per * k; if (a >= n) break; size_t len = n - a < per ? n - a : per; th.emplace_back([=] { memcpy(d + a, s + a, len); }); }
    for (auto& t : th) t.join();
}
int main()
{
    const size_t n = 1000000000;
    uint8_t* d; hipMalloc(&d, n);
    uint8_t* pageable = (uint8_t*)malloc(n); memset(pageable, 1, n);
    double t0 = now(); uint8_t* pinned; hipHostMalloc(&pinned, n, hipHostMallocPortable); double t1 = now();
    printf("hipHostMalloc 1e9 B: %.1f ms\n", (t1 - t0) * 1e3);
    memset(pinned, 2, n);
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now(); hipMemcpy(d, pageable, n, hipMemcpyHostToDevice); hipDeviceSynchronize(); t1 = now();
        printf("H2D pageable: %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); hipMemcpy(d, pinned, n, hipMemcpyHostToDevice); hipDeviceSynchronize(); t1 = now();
        printf("H2D pinned:   %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); hipMemcpy(pageable, d, n, hipMemcpyDeviceToHost); hipDeviceSynchronize(); t1 = now();
        printf("D2H pageable: %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); hipMemcpy(pinned, d, n, hipMemcpyDeviceToHost); hipDeviceSynchronize(); t1 = now();
        printf("D2H pinned:   %.1f GB/s\n", n / (t1 - t0) / 1e9);
    }
    for (int th : {1, 2, 4, 8, 16}) {
        t0 = now(); par_memcpy(pinned, pageable, n, th); t1 = now();
        printf("memcpy pageable->pinned, %2d threads: %.1f GB/s\n", th, n / (t1 - t0) / 1e9);
    }
    t0 = now(); uint8_t* fresh = (uint8_t*)malloc(n); par_memcpy(fresh, pinned, n, 1); t1 = now();
    printf("malloc + first-touch memcpy from pinned, 1 thread: %.1f GB/s\n", n / (t1 - t0) / 1e9);
    free(fresh);
    t0 = now(); fresh = (uint8_t*)malloc(n); par_memcpy(fresh, pinned, n, 8); t1 = now();
    printf("malloc + first-touch memcpy from pinned, 8 threads: %.1f GB/s\n", n / (t1 - t0) / 1e9);
    t0 = now(); hipError_t e = hipHostRegister(fresh, n, hipHostRegisterDefault); t1 = now();
    printf("hipHostRegister 1e9 B: %.1f ms (%s)\n", (t1 - t0) * 1e3, hipGetErrorString(e));
    if (e == hipSuccess) {
        t0 = now(); hipMemcpy(d, fresh, n, hipMemcpyHostToDevice); hipDeviceSynchronize(); t1 = now();
        printf("H2D registered: %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); hipHostUnregister(fresh); t1 = now();
        printf("hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
    }
    // both directions at once, from two threads on two streams (what the _MT scheduler's feeder and drainer do): pageable against pinned
    {
        uint8_t* d2; hipMalloc(&d2, n);
        hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        uint8_t* pageable2 = (uint8_t*)malloc(n); memset(pageable2, 3, n);
        uint8_t* pinned2; hipHostMalloc(&pinned2, n, hipHostMallocPortable); memset(pinned2, 4, n);
        auto both = [&](uint8_t* up, uint8_t* down, const char* what) {
            const size_t piece = n / 4;
            for (int rep = 0; rep < 2; ++rep) {
                double a0 = now();
                std::thread ta([&] { for (int k = 0; k < 4; ++k) hipMemcpyAsync(d + k * piece, up + k * piece, piece, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); });
                std::thread tb([&] { for (int k = 0; k < 4; ++k) hipMemcpyAsync(down + k * piece, d2 + k * piece, piece, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); });
                ta.join(); tb.join();
                double a1 = now();
                printf("H2D + D2H at once, %s: %.1f ms for 1e9 B each way = %.1f GB/s per direction\n", what, (a1 - a0) * 1e3, n / (a1 - a0) / 1e9);
            }
        };
        both(pageable, pageable2, "pageable / pageable");
        both(pinned, pinned2, "pinned / pinned");
        both(pageable, pinned2, "pageable up / pinned down");
        both(pinned, pageable2, "pinned up / pageable down");
        hipHostRegister(pageable2, n, hipHostRegisterDefault);
        both(pageable, pageable2, "pageable up / registered down");
        hipHostUnregister(pageable2);
    }
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
