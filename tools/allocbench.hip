// tools/allocbench.hip -- what obtaining device memory costs on this box (not part of the product):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/allocbench tools/allocbench.hip
//   /tmp/allocbench sync|async [buffers, default 4] [GB each, default 48]        (one process per line of the table)
// hipMalloc / hipMallocAsync of n buffers, a memset over them (first touch), a second memset, the frees.
// Measured (profiles/r04/allocbench.log, processes one after the other on one box): the FIRST process gets 4 x 48 GB in
// 1 ms; every process after it waits 3.6 - 5.6 s for the same 192 GB, whichever call it uses -- the driver clears memory
// that another process has just given back (15 - 40 ms per GB), a clean box hands it out at once.  First touch costs
// nothing either way (192 GB memset in 34 ms).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const bool async = argc > 1 && !strcmp(argv[1], "async");
    const int n = argc > 2 ? atoi(argv[2]) : 4;
    const size_t gb = argc > 3 ? (size_t)atoi(argv[3]) : 48;
    (void)hipFree(nullptr);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    if (async) {
        hipMemPool_t pool;
        (void)hipDeviceGetDefaultMemPool(&pool, 0);
        uint64_t thr = ~0ull;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    }
    std::vector<void*> v;
    double t0 = now();
    for (int i = 0; i < n; i++) {
        void* p = nullptr;
        hipError_t e = async ? hipMallocAsync(&p, gb << 30, s) : hipMalloc(&p, gb << 30);
        if (e != hipSuccess) { printf("alloc %d failed: %s\n", i, hipGetErrorString(e)); break; }
        v.push_back(p);
    }
    (void)hipStreamSynchronize(s);
    double t1 = now();
    for (void* p : v) (void)hipMemsetAsync(p, 1, gb << 30, s);
    (void)hipStreamSynchronize(s);
    double t2 = now();
    for (void* p : v) (void)hipMemsetAsync(p, 2, gb << 30, s);
    (void)hipStreamSynchronize(s);
    double t3 = now();
    for (void* p : v) { if (async) (void)hipFreeAsync(p, s); else (void)hipFree(p); }
    (void)hipStreamSynchronize(s);
    double t4 = now();
    printf("%s: %zu x %zu GB: alloc %.3f s, first memset %.3f s, second memset %.3f s, free %.3f s\n", async ? "hipMallocAsync" : "hipMalloc", v.size(), gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    return 0;
}
