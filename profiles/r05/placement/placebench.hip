// tools/placebench.hip -- round 5: is the "placement class" of a pass-1 output buffer a matter of how 524288 write
// cursors at near-regular strides fall onto the memory channels?  Several plain allocations and one physically
// contiguous one, each probed with the library's pattern (k_probe_scatter: 2048 workgroups x 256 regions, 256-byte runs
// off a line boundary) and with variants that move the cursors: regions walked in a per-workgroup order, region bases
// padded, per-workgroup slices padded, runs on line boundaries.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// reg / per in keys; rot: workgroup w starts its walk at region group (w * rot) % 16; shift: run start off the line (keys)
__global__ __launch_bounds__(512) void k_probe(u64* __restrict__ dst, u64 reg, u64 per, u64 per_len, u32 rot, u32 shift) {
    const u32 tid = threadIdx.x;
    const u32 r0 = (blockIdx.x * rot) & 15;
    for (u64 off = 0; off + 33 <= per_len; off += 32)
        for (u32 q = 0; q < 16; q++) {
            const u32 r = ((q + r0) & 15) * 16 + tid / 32;
            dst[(u64)r * reg + (u64)blockIdx.x * per + off + (tid & 31) + shift] = off;
        }
}

static double probe(u64* p, u64 reg, u64 per, u64 per_len, u32 rot, u32 shift) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe, dim3(2048), dim3(512), 0, 0, p, reg, per, per_len, rot, shift);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_probe, dim3(2048), dim3(512), 0, 0, p, reg, per, per_len, rot, shift);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms / 3;
}

int main(int argc, char** argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 10;
    const u64 nkeys = 100000000ull;
    const u64 bytes = nkeys * 8 + 256 * (10ull << 20);      // room for the padded layouts (10 MB per region)
    std::vector<u64*> bufs;
    for (int i = 0; i < nbuf; i++) {
        u64* p;
        CHECK(hipMalloc(&p, bytes));
        CHECK(hipMemset(p, 0, bytes));
        bufs.push_back(p);
    }
    u64* pc = nullptr;
    if (hipExtMallocWithFlags((void**)&pc, bytes, hipDeviceMallocContiguous) == hipSuccess) {
        CHECK(hipMemset(pc, 0, bytes));
        bufs.push_back(pc);
    }
    const u64 reg = nkeys / 256, per = reg / 2048;
    printf("%d plain allocations%s of %.0f MB; probe = ms per launch (800 MB written)\n", nbuf, pc ? " + 1 contiguous (last)" : "", bytes / 1e6);
    printf("%-46s", "variant");
    for (size_t i = 0; i < bufs.size(); i++) printf(" %6zu", i);
    printf("\n");
    struct V { const char* name; u64 reg, per; u32 rot, shift; };
    const V vs[] = {
        {"library pattern (+8 B)", reg, per, 0, 1},
        {"runs on line boundaries (+0)", reg, per - per % 16, 0, 0},
        {"+64 B", reg, per, 0, 8},
        {"regions walked in per-workgroup order", reg, per, 1, 1},
        {"per-workgroup order, odd rotation 5", reg, per, 5, 1},
        {"region bases padded by 4 KB + 128 B", reg + 528, per, 0, 1},
        {"region bases padded by 68 KB", reg + 8704, per, 0, 1},
        {"region bases padded by 1 MB + 4 KB", reg + 131584, per, 0, 1},
        {"slices padded to a multiple of 256 B", reg + 2048 * 32, per + (32 - per % 32), 0, 1},
        {"slices padded by 128 B", reg + 2048 * 16, per + 16, 0, 1},
        {"slices padded by 4 KB + 128 B", reg + 2048 * 528, per + 528, 0, 1},
        {"regions 13.2 MB apart (whole 3.4 GB touched)", 1650000, per, 0, 1},
        {"regions 13.2 MB apart, slices 6 KB apart", 1650000, 4 * per, 0, 1},
        {"regions 6.6 MB apart", 825000, per, 0, 1},
        {"library pattern, upper part of the buffer", reg, per, 0, 1 + 300000000},
        {"library pattern again", reg, per, 0, 1},
    };
    for (const V& v : vs) {
        printf("%-46s", v.name);
        for (u64* p : bufs) printf(" %6.3f", probe(p, v.reg, v.per, per, v.rot, v.shift));
        printf("\n");
    }
    return 0;
}
