// tools/p2bench.hip -- k_scatter2 by itself on the real pass-1 output of a synthetic genome, whole or with parts taken
// out (-DP2_ABL=n, k_sort.inc): what its time is made of.  Built with the library's translation unit included, so the
// kernel is the product's; a part that decides where keys go is only ever taken out together with the global stores
// (static_assert in k_sort.inc), so no variant writes anywhere the product would not.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-DP2_ABL=n] -o tools/p2bench_n tools/p2bench.hip -lrccl -lz -ldl
//   ./tools/p2bench_n [bases, default 50000000]
#include "../krisp_amd/csrc/krisp_hip.hip"

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 50000000;
    std::vector<uint8_t> bases(n);
    u64 x = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; bases[i] = "ACGT"[(x >> 33) & 3]; }
    kr_ctx* c = kr_create(0, 0);
    if (!c) { printf("no context: %s\n", kr_last_error(nullptr)); return 1; }
    if (kr_set_params(c, 25, 1, 2, 0, n) || kr_genome_upload(c, 0, bases.data(), n) || kr_genome_sort(c, 0) || kr_sync(c)) {
        printf("set-up failed: %s\n", kr_last_error(c));
        return 1;
    }
    Genome& G = c->genomes.find(0)->second;
    Lane& ln = c->lanes[c->last_lane];
    Slice& S = G.sl[0];
    const int b = c->g.b;
    const u32 nb = 1u << b;
    const u32 tile2 = 2 * P2_TILE;
    const u32 ntmax = (u32)(S.nmax / tile2) + 257;
    if (c->nslices != 1 || b <= 8 || !ln.cursor.p) { printf("not the reserving pass 2 (b = %d, slices %d)\n", b, c->nslices); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; r++) {
        // the cursors as k_hist16_off leaves them: the fine offsets
        hipMemcpyAsync(ln.cursor.p, S.off.p, (size_t)nb * 4, hipMemcpyDeviceToDevice, c->stream);
        hipEventRecord(e0, c->stream);
        launch_scatter2(c->stream, (const u64*)ln.tmpkeys.p, (u64*)S.keys.p, (const uint2*)ln.tiledesc.p, ntmax, (const u32*)nullptr,
                        (u32*)ln.cursor.p, b);
        hipEventRecord(e1, c->stream);
        hipStreamSynchronize(c->stream);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { best = std::min(best, ms); sum += ms; }
    }
    const double keys = (double)S.nmax;
    printf("P2_ABL=%d  b=%d  keys=%.0f  tiles<=%u  k_scatter2: best %.4f ms, mean %.4f ms  (%.2f TB/s of 16 B per key at the mean)\n",
           (int)P2_ABL, b, keys, ntmax, best, sum / reps, keys * 16 / (sum / reps * 1e-3) / 1e12);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(err));
    kr_destroy(c);
    return 0;
}
