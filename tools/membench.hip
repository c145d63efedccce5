// tools/membench.hip -- what the memory system of this box gives the access patterns of the sort
// (not part of the product; built and run by tools/membench.sh on the GPU box).
//   copy / read / write : streaming, 16 bytes per lane, U loads in flight per thread
//   scatter             : the write pattern of a radix partition pass -- every workgroup reads a
//                         contiguous tile and writes it as NB runs of RUN bytes, run r of tile t
//                         going to bucket r's region (base r * bucket_bytes) at the tile's slot.
//                         bucket_bytes = N / NB (pass 1: buckets span the whole array) or a small
//                         window (pass 2: the 256 fine buckets of one pass-1 bucket are 12 KB apart).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int U>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, u64 n16) {
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int q = 0; q < U; q++) v[q] = src[i + q * stride];
#pragma unroll
        for (int q = 0; q < U; q++) dst[i + q * stride] = v[q];
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, u32* __restrict__ out, u64 n16) {
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int q = 0; q < U; q++) v[q] = src[i + q * stride];
#pragma unroll
        for (int q = 0; q < U; q++) acc += v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    }
    if (acc == 0x12345) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ dst, u64 n16) {
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const uint4 v = make_uint4((u32)i, 2, 3, 4);
    for (; i + (U - 1) * stride < n16; i += U * stride) {
#pragma unroll
        for (int q = 0; q < U; q++) dst[i + q * stride] = v;
    }
}

// tile = 256 runs of RUN16 x 16 bytes.  Persistent workgroups of 1024 threads; a tile is read
// contiguously and written as runs.  shift8 = offset of every run start in 8-byte units (real
// cursors fall on any 8-byte boundary); W8 = 1: 8-byte stores (one key per lane, as the sort
// kernels store), 0: 16-byte stores; blocked = 1: a workgroup takes CONSECUTIVE tiles (adjacent
// runs of a bucket then come from the same CU, one after the other), 0: tiles strided over the grid.
template <int RUN16, int W8, int NOREAD = 0>
__global__ __launch_bounds__(1024) void k_scatter(const uint4* __restrict__ src, uint4* __restrict__ dst, u32 ntiles,
                                                  u64 bucket16, u32 shift8, u32 blocked) {
    constexpr u32 tile16 = 256 * RUN16;
    constexpr int IT = tile16 / 1024;            // 16-byte items per thread and tile
    constexpr int UN = IT < 8 ? IT : 8;
    const u32 per = (ntiles + gridDim.x - 1) / gridDim.x;
    const u32 tb = blocked ? blockIdx.x * per : blockIdx.x;
    const u32 te = blocked ? min(ntiles, tb + per) : ntiles;
    const u32 ts = blocked ? 1 : gridDim.x;
    for (u32 t = tb; t < te; t += ts) {
        const uint4* s = src + (u64)t * tile16;
#pragma unroll 1
        for (int i0 = 0; i0 < IT; i0 += UN) {
            uint4 v[UN];
#pragma unroll
            for (int q = 0; q < UN; q++) v[q] = NOREAD ? make_uint4(t, i0 + q, threadIdx.x, 7) : s[(i0 + q) * 1024 + threadIdx.x];
            if (W8) {
                // lane l of a wave stores key 2l and key 2l+1 of its 16 bytes in two 8-byte store
                // instructions that each cover the wave's span with stride 8 (as `dst[p] = key`)
#pragma unroll
                for (int q = 0; q < UN; q++) {
                    const u32 i = (i0 + q) * 1024 + (threadIdx.x & ~63u);       // first 16-byte item of the wave
                    const u32 lane = threadIdx.x & 63;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const u32 k8 = 2 * i + h * 64 + lane;                    // key index inside the tile
                        const u32 r = k8 / (2 * RUN16), o = k8 % (2 * RUN16);
                        u64* p = (u64*)dst + 2 * ((u64)r * bucket16 + (u64)t * RUN16) + o + shift8;
                        *p = h ? (((u64)v[q].w << 32) | v[q].z) : (((u64)v[q].y << 32) | v[q].x);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < UN; q++) {
                    const u32 i = (i0 + q) * 1024 + threadIdx.x;
                    const u32 r = i / RUN16, o = i % RUN16;
                    const u64 d = (u64)r * bucket16 + (u64)t * RUN16 + o;
                    uint4* p = (uint4*)((u64*)(dst + d) + shift8);
                    *p = v[q];
                }
            }
        }
    }
}

// ---- round 4: what would PACKED 5-byte keys cost / give?  (at k = 28 and fan-out 2^16 the sorted key of a fine bucket
// has 40 significant bits: VERDICT r3 item 10.)  One array of 5-byte elements, runs of 64 keys = 320 bytes on ANY byte
// boundary, a key stored as an unaligned dword + a byte; the reader loads 8 unaligned bytes per key.
typedef u32 __attribute__((aligned(1))) u32u;
typedef u64 __attribute__((aligned(1))) u64u;
template <int NOREAD>
__global__ __launch_bounds__(1024) void k_scatter5(const uint4* __restrict__ src, unsigned char* __restrict__ dst, u32 ntiles,
                                                   u64 bucket_bytes, u32 shift) {
    constexpr u32 TK = 16384, RUNK = 64;            // keys per tile, keys per run (256 runs)
    for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint4* s = src + (u64)t * (TK / 2);
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = NOREAD ? make_uint4(t, q, threadIdx.x, 7) : s[q * 1024 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const u32 i = q * 1024 + (threadIdx.x & ~63u), lane = threadIdx.x & 63;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 k8 = 2 * i + h * 64 + lane;
                const u32 r = k8 / RUNK, o = k8 % RUNK;
                unsigned char* p = dst + (u64)r * bucket_bytes + (u64)t * (RUNK * 5) + o * 5 + shift;
                const u64 key = h ? (((u64)v[q].w << 32) | v[q].z) : (((u64)v[q].y << 32) | v[q].x);
                *(u32u*)p = (u32)key;
                p[4] = (unsigned char)(key >> 32);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_read5(const unsigned char* __restrict__ src, u32* __restrict__ out, u64 nkeys) {
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 acc = 0;
    for (; i + 3 * stride < nkeys; i += 4 * stride) {
        const u64 a = *(const u64u*)(src + 5 * i), b = *(const u64u*)(src + 5 * (i + stride)),
                  c = *(const u64u*)(src + 5 * (i + 2 * stride)), d = *(const u64u*)(src + 5 * (i + 3 * stride));
        acc += (a & 0xFFFFFFFFFFull) ^ (b & 0xFFFFFFFFFFull) ^ (c & 0xFFFFFFFFFFull) ^ (d & 0xFFFFFFFFFFull);
    }
    if (acc == 0x12345) out[0] = (u32)acc;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static double timeit(F f, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    f();
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const u64 bytes = 1ull << 30;       // 1 GiB per array
    const u64 n16 = bytes / 16;
    uint4 *a, *b;
    u32* out;
    CHECK(hipMalloc(&a, bytes + (1 << 20)));
    CHECK(hipMalloc(&b, bytes + (64 << 20)));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    for (int grid : {1024, 2048, 4096, 8192}) {
        double t1 = timeit([&] { hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 10);
        double t4 = timeit([&] { hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 10);
        double t8 = timeit([&] { hipLaunchKernelGGL(k_copy<8>, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 10);
        printf("copy  grid %5d: U1 %.0f  U4 %.0f  U8 %.0f GB/s (read+write)\n", grid, 2 * bytes / t1 / 1e6,
               2 * bytes / t4 / 1e6, 2 * bytes / t8 / 1e6);
    }
    for (int grid : {2048, 8192}) {
        double r4 = timeit([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, out, n16); }, 10);
        double r8 = timeit([&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, a, out, n16); }, 10);
        double w4 = timeit([&] { hipLaunchKernelGGL(k_write<4>, dim3(grid), dim3(256), 0, 0, b, n16); }, 10);
        printf("grid %5d: read U4 %.0f U8 %.0f  write U4 %.0f GB/s\n", grid, bytes / r4 / 1e6, bytes / r8 / 1e6,
               bytes / w4 / 1e6);
    }
    // scatter: 800 MB (1e8 keys) in, 256 runs per tile; bucket r = the r-th 1/256 of the output
    const u64 N16 = 50000000;       // 1e8 keys = 5e7 x 16 bytes
    auto run = [&](auto tag, u32 run_bytes) {
        constexpr int RUN16 = decltype(tag)::value;
        const u32 ntiles = (u32)(N16 / (256 * RUN16));
        const u64 bucket16 = (u64)ntiles * RUN16;
        for (u32 blocked = 0; blocked < 2; blocked++)
            for (u32 w8 = 0; w8 < 2; w8++) {
                printf("runs %4u B %s %s stores:", run_bytes, blocked ? "blocked" : "strided", w8 ? " 8-B" : "16-B");
                for (u32 shift8 : {0u, 1u, 2u, 4u, 8u}) {
                    double t = timeit([&] {
                        if (w8) hipLaunchKernelGGL((k_scatter<RUN16, 1>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, bucket16, shift8, blocked);
                        else hipLaunchKernelGGL((k_scatter<RUN16, 0>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, bucket16, shift8, blocked);
                    }, 5);
                    printf("  +%2u B: %4.0f", shift8 * 8, 2.0 * N16 * 16 / t / 1e6);
                }
                printf("  GB/s\n");
            }
    };
    // the same scatter on smaller arrays: does the 256 MB memory-side cache (which merges partial lines before
    // they reach DRAM) lift the rate of unaligned runs?  512-byte runs, 8-byte stores, read + write footprint 2 x bytes
    for (u64 n16 : {50000000ull, 12500000ull, 6250000ull, 3125000ull, 1562500ull}) {
        constexpr int RUN16 = 32;
        const u32 ntiles = (u32)(n16 / (256 * RUN16));
        const u64 bucket16 = (u64)ntiles * RUN16;
        printf("array %6.1f MB, runs 512 B:", n16 * 16 / 1e6);
        for (u32 shift8 : {0u, 1u}) {
            double t = timeit([&] { hipLaunchKernelGGL((k_scatter<RUN16, 1>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, bucket16, shift8, 0u); }, 20);
            printf("  +%u B: %4.0f", shift8 * 8, 2.0 * (u64)ntiles * 256 * RUN16 * 16 / t / 1e6);
        }
        double tc = timeit([&] { hipLaunchKernelGGL(k_copy<1>, dim3(1024), dim3(256), 0, 0, a, b, n16); }, 20);
        printf("   copy %4.0f GB/s\n", 2.0 * n16 * 16 / tc / 1e6);
    }
    // write-only scatter (pass 1 generates its keys): 256-byte runs, 8-byte stores, by alignment of the run starts
    {
        constexpr int RUN16 = 16;
        const u32 ntiles = (u32)(N16 / (256 * RUN16));
        const u64 bucket16 = (u64)ntiles * RUN16;
        printf("write-only, runs 256 B, 8-B stores:");
        for (u32 shift8 : {0u, 1u, 2u, 4u, 8u}) {
            double t = timeit([&] { hipLaunchKernelGGL((k_scatter<RUN16, 1, 1>), dim3(512), dim3(1024), 0, 0, a, b, ntiles, bucket16, shift8, 0u); }, 5);
            printf("  +%2u B: %4.0f", shift8 * 8, 1.0 * N16 * 16 / t / 1e6);
        }
        printf("  GB/s\n");
    }
    // round 5: the same write-only pattern with longer runs (what a pass 1 with 16384- / 32768-key tiles would write)
    {
        auto wo = [&](auto tag, u32 run_bytes) {
            constexpr int RUN16 = decltype(tag)::value;
            const u32 ntiles = (u32)(N16 / (256 * RUN16));
            const u64 bucket16 = (u64)ntiles * RUN16;
            for (int grid : {256, 512}) {
                printf("write-only, runs %4u B, 8-B stores, %d workgroups:", run_bytes, grid);
                for (u32 shift8 : {0u, 1u, 4u, 8u}) {
                    double t = timeit([&] { hipLaunchKernelGGL((k_scatter<RUN16, 1, 1>), dim3(grid), dim3(1024), 0, 0, a, b, ntiles, bucket16, shift8, 0u); }, 5);
                    printf("  +%2u B: %4.0f", shift8 * 8, 1.0 * (u64)ntiles * 256 * RUN16 * 16 / t / 1e6);
                }
                printf("  GB/s\n");
            }
        };
        wo(std::integral_constant<int, 16>{}, 256);
        wo(std::integral_constant<int, 32>{}, 512);
        wo(std::integral_constant<int, 64>{}, 1024);
        wo(std::integral_constant<int, 128>{}, 2048);
    }
    // packed 5-byte keys: pass 2's pattern (16384-key tiles, 256 runs of 64 keys) written as 320-byte runs on any byte
    // boundary against the 512-byte runs of 8-byte keys above; and a streaming read of the packed array
    {
        const u32 ntiles = (u32)(2 * N16 / 16384);
        const u64 bucket_bytes = (u64)ntiles * 320;
        for (u32 shift : {0u, 1u, 3u, 5u}) {
            double t = timeit([&] { hipLaunchKernelGGL(k_scatter5<0>, dim3(256), dim3(1024), 0, 0, a, (unsigned char*)b, ntiles, bucket_bytes, shift); }, 5);
            printf("packed 5-byte keys, runs 320 B, +%u B: %5.1f G keys/s = %4.0f GB/s moved (8-byte form at 512-B runs: see below)\n", shift,
                   (double)ntiles * 16384 / t / 1e6, (double)ntiles * 16384 * 13 / t / 1e6);
        }
        const u64 nk = (u64)ntiles * 16384;
        for (int grid : {2048, 8192}) {
            double t = timeit([&] { hipLaunchKernelGGL(k_read5, dim3(grid), dim3(256), 0, 0, (const unsigned char*)b, out, nk); }, 10);
            printf("packed 5-byte keys, streaming read, grid %d: %5.1f G keys/s = %4.0f GB/s of packed bytes\n", grid, nk / t / 1e6, nk * 5 / t / 1e6);
        }
        double t8 = timeit([&] { hipLaunchKernelGGL((k_scatter<32, 1>), dim3(256), dim3(1024), 0, 0, a, b, (u32)(N16 / (256 * 32)), (u64)(N16 / (256 * 32)) * 32, 1u, 0u); }, 5);
        printf("8-byte keys, runs 512 B, +8 B: %5.1f G keys/s = %4.0f GB/s moved\n", 2.0 * N16 / t8 / 1e6, 2.0 * N16 * 16 / t8 / 1e6);
    }
    run(std::integral_constant<int, 8>{}, 128);
    run(std::integral_constant<int, 16>{}, 256);
    run(std::integral_constant<int, 32>{}, 512);
    run(std::integral_constant<int, 64>{}, 1024);
    return 0;
}
