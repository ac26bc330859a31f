// How fast can a fresh malloc() of 1 GB be made resident?  (host side of the reference-named API)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void par(size_t n, int threads, F f)
{
    std::vector<std::thread> th;
    size_t per = ((n / threads) + (2u << 20) - 1) & ~size_t((2u << 20) - 1);
    for (int k = 0; k < threads; ++k) { size_t a = per * k; if (a >= n) break; size_t len = n - a < per ? n - a : per; th.emplace_back([=] { f(a, len); }); }
    for (auto& t : th) t.join();
}
int main()
{
    const size_t n = 1000000000;
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char buf[128] = {0}; if (f) { fgets(buf, 127, f); fclose(f); } printf("THP: %s", buf);
    for (int threads : {1, 4, 16, 64}) {
        uint8_t* p = (uint8_t*)malloc(n); double t0 = now();
        par(n, threads, [=](size_t a, size_t len) { for (size_t o = 0; o < len; o += 4096) p[a + o] = 1; });
        printf("touch 4K stride, %2d threads: %.1f ms\n", threads, (now() - t0) * 1e3); free(p);
    }
    for (int threads : {1, 16}) {
        uint8_t* p = (uint8_t*)malloc(n); double t0 = now();
        uintptr_t a0 = ((uintptr_t)p + 4095) & ~uintptr_t(4095);
        int rc = madvise((void*)a0, n - 4096, MADV_HUGEPAGE);
        par(n, threads, [=](size_t a, size_t len) { for (size_t o = 0; o < len; o += 4096) p[a + o] = 1; });
        printf("MADV_HUGEPAGE (rc %d) + touch, %2d threads: %.1f ms\n", rc, threads, (now() - t0) * 1e3); free(p);
    }
    for (int threads : {1, 16}) {
        uint8_t* p = (uint8_t*)malloc(n); double t0 = now();
        uintptr_t a0 = ((uintptr_t)p + 4095) & ~uintptr_t(4095);
        int rc = 0;
        par(n - 8192, threads, [&](size_t a, size_t len) { int r = madvise((void*)(a0 + a), len & ~size_t(4095), MADV_POPULATE_WRITE); if (r) rc = r; });
        printf("MADV_POPULATE_WRITE (rc %d), %2d threads: %.1f ms\n", rc, threads, (now() - t0) * 1e3); free(p);
    }
    {   // what a recycled buffer costs: memset of resident memory
        uint8_t* p = (uint8_t*)malloc(n); memset(p, 1, n); double t0 = now(); memset(p, 2, n); printf("memset resident 1 thread: %.1f ms\n", (now() - t0) * 1e3); free(p);
    }
    return 0;
}
