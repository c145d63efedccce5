// what a pinned block costs against copying from pageable memory (one-shot ingest of a 50 MB text): hipcc -O2 -o tools/pinbench tools/pinbench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 52 << 20;
    void* d; hipMalloc(&d, n);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); void* p; hipHostMalloc(&p, n, hipHostMallocDefault); double t1 = now();
        memset(p, 1, n); double t2 = now();
        hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t3 = now();
        hipHostFree(p); double t4 = now();
        void* q = malloc(n); memset(q, 1, n); double t5 = now();
        hipMemcpyAsync(d, q, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t6 = now();
        hipHostRegister(q, n, hipHostRegisterDefault); double t7 = now();
        hipMemcpyAsync(d, q, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t8 = now();
        hipHostUnregister(q); double t9 = now();
        free(q);
        printf("52 MB: hipHostMalloc %.1f ms, fill %.1f, H2D pinned %.1f, hipHostFree %.1f | malloc+fill %.1f, H2D pageable %.1f | hipHostRegister %.1f, H2D registered %.1f, unregister %.1f\n",
               t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8);
    }
    return 0;
}
