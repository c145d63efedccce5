// tools/icbench.hip -- round 5: does blocking pass 1 -> pass 2 of the radix partition by the 256 MB Infinity Cache pay?
// Pass 1 writes 1e8 keys (800 MB) as 256-byte runs off line boundaries into 256 buckets, pass 2 reads them back
// tile by tile and writes 512-byte runs elsewhere.  (A) whole array: pass 1 over everything, then pass 2 (the keys make
// a round trip through HBM); (B) in NC chunks through ONE chunk buffer of 800 MB / NC: pass 1 of a chunk, pass 2 of the
// chunk, next chunk -- a chunk that fits the cache never reaches HBM between the two.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pass 1: ntiles tiles of 8192 keys generated in registers, written as 256 runs of 32 keys (256 B) at +8 B;
// run r of tile t -> key offset r * bucket_keys + t * 32 + 1
__global__ __launch_bounds__(512) void k_p1(u64* __restrict__ dst, u32 ntiles, u64 bucket_keys) {
    for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const u32 p = q * 512 + threadIdx.x;            // key of the tile in digit order
            const u32 r = p >> 5, o = p & 31;
            dst[(u64)r * bucket_keys + (u64)t * 32 + o + 1] = ((u64)t << 32) | p;
        }
    }
}
// pass 2: tiles of 16384 keys read contiguously from src (16-byte loads), written as 256 runs of 64 keys (512 B) at
// +8 B: run r of tile t -> key offset (r * out_bucket_keys + (t0 + t) * 64 + 1)
__global__ __launch_bounds__(1024) void k_p2(const uint4* __restrict__ src, u64* __restrict__ dst, u32 ntiles, u32 t0,
                                             u64 out_bucket_keys) {
    for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint4* s = src + (u64)t * 8192;
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = s[q * 1024 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const u32 i = q * 1024 + (threadIdx.x & ~63u), lane = threadIdx.x & 63;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 k8 = 2 * i + h * 64 + lane;
                const u32 r = k8 >> 6, o = k8 & 63;
                dst[(u64)r * out_bucket_keys + (u64)(t0 + t) * 64 + o + 1] =
                    h ? (((u64)v[q].w << 32) | v[q].z) : (((u64)v[q].y << 32) | v[q].x);
            }
        }
    }
}

int main(int argc, char** argv) {
    const u64 N = 100000000ull / 16384 * 16384;         // keys
    u64 *P, *F;
    CHECK(hipMalloc(&P, N * 8 + (64 << 20)));
    CHECK(hipMalloc(&F, N * 8 + (64 << 20)));
    CHECK(hipMemset(P, 0, N * 8));
    CHECK(hipMemset(F, 0, N * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const u32 nt2_all = (u32)(N / 16384);
    for (int rep = 0; rep < 2; rep++)
        for (int nc : {1, 2, 4, 8, 16, 32}) {
            const u64 ck = N / nc / 16384 * 16384;      // keys per chunk
            const u32 nt1 = (u32)(ck / 8192), nt2 = (u32)(ck / 16384);
            float ms1 = 0, ms2 = 0, ms;
            // warm
            hipLaunchKernelGGL(k_p1, dim3(512), dim3(512), 0, 0, P, nt1, ck / 256);
            CHECK(hipDeviceSynchronize());
            const int reps = 5;
            CHECK(hipEventRecord(a));
            for (int r = 0; r < reps; r++)
                for (int c = 0; c < nc; c++) {
                    hipLaunchKernelGGL(k_p1, dim3(512), dim3(512), 0, 0, P, nt1, ck / 256);
                    hipLaunchKernelGGL(k_p2, dim3(256), dim3(1024), 0, 0, (const uint4*)P, F, nt2, (u32)c * nt2, (u64)nt2_all * 64);
                }
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            CHECK(hipEventElapsedTime(&ms, a, b));
            // the two passes by themselves at this chunk size (back to back launches of the same pass)
            CHECK(hipEventRecord(a));
            for (int r = 0; r < reps; r++)
                for (int c = 0; c < nc; c++) hipLaunchKernelGGL(k_p1, dim3(512), dim3(512), 0, 0, P, nt1, ck / 256);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            CHECK(hipEventElapsedTime(&ms1, a, b));
            CHECK(hipEventRecord(a));
            for (int r = 0; r < reps; r++)
                for (int c = 0; c < nc; c++)
                    hipLaunchKernelGGL(k_p2, dim3(256), dim3(1024), 0, 0, (const uint4*)P, F, nt2, (u32)c * nt2, (u64)nt2_all * 64);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            CHECK(hipEventElapsedTime(&ms2, a, b));
            printf("%2d chunk(s) of %6.1f MB: pass 1 + pass 2 interleaved %.3f ms per 1e8 keys (pass 1 alone %.3f, pass 2 alone %.3f, sum %.3f)\n",
                   nc, ck * 8 / 1e6, ms / reps, ms1 / reps, ms2 / reps, (ms1 + ms2) / reps);
        }
    return 0;
}
