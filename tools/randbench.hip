// tools/randbench.hip -- random look-ups into a large table: rate against the table's footprint (address
// translation reach, cache reach) and against the bytes fetched per look-up.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
// MODE 0: 8 bytes per look-up; 1: 64 bytes (4 x 16) of one aligned 64-byte slot; 2: 8 bytes, then a DEPENDENT
// second 8-byte read at a place derived from the first (index + keys); 3: 32 bytes (2 x 16);
// 4: 32 bytes, runs of 8 consecutive lanes read the SAME record (neighbouring windows that share a minimizer);
// 5: as 4 with a dependent second 32-byte read shared by the same 8 lanes (bucket index, then bucket)
template <int MODE>
__global__ __launch_bounds__(256) void k_rand(const uint4* __restrict__ t, u64 nslots, u32 per, u32* out) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 acc = 0;
    for (u32 it = 0; it < per; it++) {
        const u64 h = mix((MODE >= 4 ? tid >> 3 : tid) * 0x9E3779B97F4A7C15ull + it);
        const u64 s = h & (nslots - 1);          // (nslots is a power of two: no 64-bit division in the loop)
        const uint4* p = t + 4 * s;
        if (MODE == 0) {
            acc += ((const u32*)p)[0];
        } else if (MODE == 1) {
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
        } else if (MODE == 3 || MODE == 4) {
            const uint4 a = p[0], b = p[1];
            acc += a.x ^ b.y;
        } else if (MODE == 5) {
            const uint4 a = p[0];
            const u64 s2 = mix(h + a.x) & (nslots - 1);
            const uint4 c = t[4 * s2], d = t[4 * s2 + 1];
            acc += c.x ^ d.y;
        } else {
            const u32 v = ((const u32*)p)[0];
            const u64 s2 = mix(h + v) & (nslots - 1);
            acc += ((const u32*)(t + 4 * s2))[1];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const u64 maxbytes = 64ull << 30;
    uint4* t;
    u32* out;
    CHECK(hipMalloc(&t, maxbytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(t, 0, maxbytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const u32 grid = 256 * 32, per = 64;
    const double looks = (double)grid * 256 * per;
    for (u64 gb4 : {1ull, 16ull, 64ull, 128ull}) {     // quarters of a GiB
        const u64 bytes = gb4 << 28, nslots = bytes / 64;
        printf("footprint %6.2f GiB:", bytes / double(1ull << 30));
        for (int mode = 0; mode < 6; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(a));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_rand<0>, dim3(grid), dim3(256), 0, 0, t, nslots, per, out); break;
                case 1: hipLaunchKernelGGL(k_rand<1>, dim3(grid), dim3(256), 0, 0, t, nslots, per, out); break;
                case 2: hipLaunchKernelGGL(k_rand<2>, dim3(grid), dim3(256), 0, 0, t, nslots, per, out); break;
                case 3: hipLaunchKernelGGL(k_rand<3>, dim3(grid), dim3(256), 0, 0, t, nslots, per, out); break;
                case 4: hipLaunchKernelGGL(k_rand<4>, dim3(grid), dim3(256), 0, 0, t, nslots, per, out); break;
                default: hipLaunchKernelGGL(k_rand<5>, dim3(grid), dim3(256), 0, 0, t, nslots, per, out); break;
                }
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("  %s %6.2f G/s", mode == 0 ? "8B" : mode == 1 ? "64B" : mode == 2 ? "8B+dep8B" : mode == 3 ? "32B" : mode == 4 ? "32B/8lanes" : "32B+dep32B/8lanes", looks / best / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
