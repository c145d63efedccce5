// tools/allocbench.hip -- what hipMalloc / hipFree of large buffers cost on this box (not part of the product):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/allocbench tools/allocbench.hip && /tmp/allocbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x * 4096) p[i * 0 + (i & ~(size_t)4095)] = 1;
}
int main() {
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 2; rep++)
        for (size_t gb : {1, 4, 16, 48}) {
            void* p = nullptr;
            double t0 = now();
            if (hipMalloc(&p, gb << 30) != hipSuccess) { printf("hipMalloc %zu GB failed\n", gb); continue; }
            double t1 = now();
            hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (char*)p, gb << 30);
            (void)hipDeviceSynchronize();
            double t2 = now();
            (void)hipFree(p);
            double t3 = now();
            printf("rep %d: %2zu GB  hipMalloc %.3f s  first kernel over it %.3f s  hipFree %.3f s\n", rep, gb, t1 - t0, t2 - t1, t3 - t2);
        }
    // many at once, as a context holds them
    std::vector<void*> v;
    double t0 = now();
    for (int i = 0; i < 4; i++) { void* p = nullptr; if (hipMalloc(&p, (size_t)48 << 30) == hipSuccess) v.push_back(p); }
    double t1 = now();
    for (void* p : v) (void)hipFree(p);
    printf("%zu x 48 GB held together: hipMalloc %.3f s, hipFree %.3f s\n", v.size(), t1 - t0, now() - t1);
    return 0;
}
