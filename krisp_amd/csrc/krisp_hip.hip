// krisp_hip.hip -- hand-written HIP (gfx950 / MI355X) for krisp_fasta's hot path:
// 2-bit pack -> both-strand keys -> MSD radix partition (LDS digit histograms,
// per-workgroup private cursors) -> LDS bucket sort -> n-way intersection with
// diagnostic-column masks -> candidate compaction -> record collection.
// C ABI: include/krisp_hip.h (each entry point cites the reference seam it replaces).
//
// Integer / byte work, HBM-bound: no MFMA anywhere (DESIGN.md "kernels").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "krisp_hip.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// ----------------------------------------------------------------------------
// geometry shared by every kernel
// ----------------------------------------------------------------------------
struct Geom {
    // --- absolute key geometry (key generation)
    int k, L, D, R;
    int sR, sD;     // layout shifts: right part << sR (= 2D), diag part >> sD (= 2R)
    u64 topmask;    // top 2k bits
    u64 mL, mR, mD; // destination masks of left / right / diag in [left|right|diag]
    int omit;       // soft-mask rule
    int strands;    // 0 both strands of every window (krisp_fasta), 1 forward only, 2 canonical = the
                    // smaller of window / reverse complement (kstream.py:679-694); 1 and 2: kernels <2>
    // --- key-space slice: genomes too large for one sort unit are sorted in 4^sb slices, one
    // per value of the first sb bases of `left`; inside a slice keys are stored RELATIVE
    // (absolute key << sbits): the slice is the same problem with geometry (L - sb, D, R)
    int sbits;      // 2 * sb
    u32 slice;      // value of the top sbits of the keys of this slice
    // --- relative geometry (everything after key generation)
    int b;          // radix fan-out bits of the two MSD passes (8 .. 18)
    int rb;         // 64 - b
    int LRrel;      // L - sb + R: bases of the (left,right) prefix of a relative key
    u64 pmask;      // top 2 * LRrel bits of a relative key
    // --- wide windows (amplicon longer than one key; see the "wide path" section).  The key
    // generators of k_hist8<1> / k_scatter1<1> read the window from global memory instead:
    int wmode;      // 1 = sub-window spectrum, 2 = dictionary composite
    int wk;         // window length: no bad base in [pos, pos + wk)
    int wfo, wro;   // mode 1: forward key = wlen bases at window offset wfo, reverse key = the
    int wlen;       //         reverse complement of the wlen bases at window offset wro
    int wL, wR;     // mode 2: key = rank of `left` in dictL << wshL | rank of `right` in dictR << wshR
    int wshL, wshR;
    int wibL, wibR; // index bits of the dictionaries (idx[top ib bits] = lower bound)
    const u64 *wdictL, *wdictR;
    const u32 *widxL, *widxR;
    // mode 2: the composite keys of a genome are generated once per phase (4 dictionary lookups
    // per window) and kept for the other passes / slices: wcache[2 * pos + strand], ~0 = none
    u64* wcache;
    int wcmode;     // 0 no cache, 1 generate + store, 2 load
};

// absolute key -> relative key of the current slice; false when the key is not in the slice
__device__ __forceinline__ bool slice_key(u64& key, const Geom& g) {
    if (g.sbits == 0) return true;
    if ((u32)(key >> (64 - g.sbits)) != g.slice) return false;
    key <<= g.sbits;
    return true;
}

// tuning constants (overridable with -D for A/B builds)
#ifndef NWG
#define NWG 2048           // persistent workgroups of the histogram / pass-1 kernels (A/B: 2048 beats 1024 by ~7 % on k_scatter1)
#endif
#ifndef P1_T
#define P1_T 512           // threads of those workgroups
#endif
#ifndef P1_WORDS
#define P1_WORDS 128       // code words per pass-1 tile (4096 positions, <= 8192 keys): a tile's 256 digit runs are
#endif                     // 256 bytes each; with 64 words / 256 threads (128-byte runs) k_scatter1 took 30 % longer
#ifndef P1_OCC
#define P1_OCC 4           // waves per SIMD k_scatter1 is compiled for: 2 workgroups of 512 threads per CU (128 VGPRs)
#endif
#define P1_TPW (P1_T / P1_WORDS)        // threads per code word
#define P1_PPT (32 / P1_TPW)            // window positions per thread
#define P1_KPT (2 * P1_PPT)             // keys per thread per tile: positions x 2 strands
#define P1_STAGE (P1_WORDS * 64)
#ifndef P2_TILE
#define P2_TILE 8192u      // keys per pass-2 tile (A/B: 8192 beats 4096 by ~20 % on k_scatter2: half the barriers per key, 256-byte runs)
#endif
#ifndef P2_T
#define P2_T 512           // threads of a pass-2 workgroup
#endif
#define P2_KPT (P2_TILE / P2_T)
#ifndef LS_T
#define LS_T 2048u         // local-sort chunk window (keys)
#endif
#ifndef LS_CAP
#define LS_CAP 4096u       // local-sort capacity (keys in LDS)
#endif
#ifndef LS_THREADS
#define LS_THREADS 512
#endif
#define LS_PER (LS_CAP / LS_THREADS)
#ifndef LS_NB
#define LS_NB 4096u        // sub-bins of the LDS bucket sort
#endif
#define LS_WPT (LS_NB / 2 / LS_THREADS)   // packed counter words per thread in the scan
#define LS_NB_LOG (LS_NB == 4096u ? 12 : (LS_NB == 2048u ? 11 : 13))
#define LS_BIN_LIMIT 48u   // a fuller sub-bin switches the chunk to the bitonic network
#ifndef BUCKET_AVG
#define BUCKET_AVG 1600ull // fan-out policy: largest average fine bucket (the limit is LS_CAP - LS_T)
#endif
#define OVF_MAX 4096       // oversized-bucket list capacity
#define PAD_WORDS 6         // all-bad code words after a genome (2 for packed windows, 6 for KR_WIDE_MAX_K bases)
#ifndef IS_SUB
#define IS_SUB 2048u       // anchor sub-tile of the intersect kernel
#endif
#ifndef IS_THREADS
#define IS_THREADS 512
#endif
#ifndef IS_NB
#define IS_NB 4096u        // sub-bins over the sub-tile's prefix span
#endif

__device__ __forceinline__ u64 layout_key(u64 w, const Geom& g) {
    return (w & g.mL) | ((w << g.sR) & g.mR) | ((w >> g.sD) & g.mD);
}

// window j (0..31) of the 32 bases of word c0 (continuing into c1): both-strand keys
__device__ __forceinline__ bool window_keys(u64 c0, u64 c1, u32 b0, u32 b1, int j, const Geom& g,
                                            u64& kf, u64& kr) {
    u64 x = j ? ((c0 << (2 * j)) | (c1 >> (64 - 2 * j))) : c0;
    u32 bm = j ? ((b0 << j) | (b1 >> (32 - j))) : b0;
    if ((bm >> (32 - g.k)) != 0) return false;
    u64 wf = x & g.topmask;
    u64 y = ~(x >> (64 - 2 * g.k));
    y = __brevll(y);
    y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
    u64 wr = y & g.topmask;
    kf = layout_key(wf, g);
    kr = layout_key(wr, g);
    return true;
}

// one key per window: the forward strand, or the canonical one (compared as plain strings,
// before the column layout, as kstream compares them before _split)
__device__ __forceinline__ bool window_key_single(u64 c0, u64 c1, u32 b0, u32 b1, int j, const Geom& g, u64& key) {
    u64 x = j ? ((c0 << (2 * j)) | (c1 >> (64 - 2 * j))) : c0;
    u32 bm = j ? ((b0 << j) | (b1 >> (32 - j))) : b0;
    if ((bm >> (32 - g.k)) != 0) return false;
    u64 w = x & g.topmask;
    if (g.strands == 2) {
        u64 y = ~(x >> (64 - 2 * g.k));
        y = __brevll(y);
        y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
        const u64 wr = y & g.topmask;
        w = wr < w ? wr : w;
    }
    key = layout_key(w, g);
    return true;
}

// ---- wide windows: the window does not fit two registers, read it from the codes array ----
// 32 bases starting at base position pos, MSB first
__device__ __forceinline__ u64 read32(const u64* __restrict__ codes, u64 pos) {
    const u64 w = pos >> 5;
    const int j = (int)(pos & 31);
    const u64 c0 = codes[w];
    return j ? ((c0 << (2 * j)) | (codes[w + 1] >> (64 - 2 * j))) : c0;
}
// reverse complement of all 32 bases of x
__device__ __forceinline__ u64 revcomp32(u64 x) {
    u64 y = __brevll(~x);
    return ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
}
// no bad base in [pos, pos + k)
__device__ __forceinline__ bool range_valid(const u32* __restrict__ bad, u64 pos, int k) {
    u64 w = pos >> 5;
    int j = (int)(pos & 31);
    while (k > 0) {
        const int take = min(32 - j, k);
        const u32 m = (take == 32 ? 0xFFFFFFFFu : ((1u << take) - 1u)) << (32 - j - take);
        if (bad[w] & m) return false;
        k -= take;
        j = 0;
        w++;
    }
    return true;
}
// rank of key in a sorted dictionary of distinct keys (idx: lower bounds by the top ib bits)
__device__ __forceinline__ bool dict_rank(const u64* __restrict__ dict, const u32* __restrict__ idx, int ib, u64 key,
                                          u32& rank) {
    const u32 bkt = (u32)(key >> (64 - ib));
    u32 lo = idx[bkt];
    const u32 end = idx[bkt + 1];
    u32 hi = end;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (dict[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo >= end || dict[lo] != key) return false;
    rank = lo;
    return true;
}
// both-strand keys of the window at base position pos; bit 0 / 1 of the result = forward /
// reverse key present
__device__ __forceinline__ u32 wide_keys(const u64* __restrict__ codes, const u32* __restrict__ bad, u64 pos,
                                         const Geom& g, u64& kf, u64& kr) {
    if (g.wmode == 1) {
        if (!range_valid(bad, pos, g.wk)) return 0;
        const int sh = 64 - 2 * g.wlen;
        kf = (read32(codes, pos + g.wfo) >> sh) << sh;
        kr = revcomp32(read32(codes, pos + g.wro)) << sh;
        return 3;
    }
    if (g.wcmode == 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(g.wcache + 2 * pos);
        kf = v.x;
        kr = v.y;
        return (kf != ~0ull ? 1u : 0u) | (kr != ~0ull ? 2u : 0u);
    }
    u32 m = 0;
    if (range_valid(bad, pos, g.wk)) {
        const int shL = 64 - 2 * g.wL, shR = 64 - 2 * g.wR;
        const u64 head = read32(codes, pos);
        u32 a, b;
        // forward strand: left = first wL bases, right = last wR bases
        if (dict_rank(g.wdictL, g.widxL, g.wibL, (head >> shL) << shL, a) &&
            dict_rank(g.wdictR, g.widxR, g.wibR, (read32(codes, pos + g.wk - g.wR) >> shR) << shR, b)) {
            kf = ((u64)a << g.wshL) | ((u64)b << g.wshR);
            m |= 1;
        }
        // reverse strand: left = rc(last wL bases), right = rc(first wR bases)
        if (dict_rank(g.wdictL, g.widxL, g.wibL, revcomp32(read32(codes, pos + g.wk - g.wL)) << shL, a) &&
            dict_rank(g.wdictR, g.widxR, g.wibR, revcomp32(head) << shR, b)) {
            kr = ((u64)a << g.wshL) | ((u64)b << g.wshR);
            m |= 2;
        }
    }
    if (g.wcmode == 1) {
        ulonglong2 v;
        v.x = (m & 1) ? kf : ~0ull;      // (rank(left) < 2^bits - 1: a composite key is never all ones)
        v.y = (m & 2) ? kr : ~0ull;
        *reinterpret_cast<ulonglong2*>(g.wcache + 2 * pos) = v;
    }
    return m;
}

// ----------------------------------------------------------------------------
// K1  ASCII -> 2-bit codes (32 bases / u64, MSB first) + bad bits (32 / u32, MSB first)
// ----------------------------------------------------------------------------
__global__ void k_pack(const uint8_t* __restrict__ bases, u64 n, u64* __restrict__ codes,
                       u32* __restrict__ bad, u64 nwords_padded, int omit) {
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; w < nwords_padded; w += stride) {
        u64 base = w * 32;
        u64 c = 0;
        u32 bd = 0;
        if (base + 32 <= n) {
            const uint4* p = reinterpret_cast<const uint4*>(bases + base);
            uint4 v0 = p[0], v1 = p[1];
            u32 wd[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int q = 0; q < 8; q++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    u32 ch = (wd[q] >> (8 * r)) & 0xFF;
                    u32 up = ch & 0xDF;
                    u32 code = (up >> 1) & 3;
                    code ^= code >> 1;
                    bool ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
                    if (omit) ok = ok & (ch == up);
                    c = (c << 2) | (ok ? code : 0);
                    bd = (bd << 1) | (ok ? 0u : 1u);
                }
            }
        } else {
            for (int q = 0; q < 32; q++) {
                u32 code = 0;
                bool ok = false;
                if (base + q < n) {
                    u32 ch = bases[base + q];
                    u32 up = ch & 0xDF;
                    code = (up >> 1) & 3;
                    code ^= code >> 1;
                    ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
                    if (omit) ok = ok & (ch == up);
                }
                c = (c << 2) | (ok ? code : 0);
                bd = (bd << 1) | (ok ? 0u : 1u);
            }
        }
        codes[w] = c;
        bad[w] = bd;
    }
}

// ----------------------------------------------------------------------------
// K2  per-workgroup histogram of the top key byte, straight from the codes.
// (The finer digit of pass 2 is counted by the pass-2 workgroup itself.)
// ----------------------------------------------------------------------------
// top key byte of both strands of window j without building the keys (valid when L >= 4:
// the first four bases of `left` are the first four bases of the strand's window)
// top 16 key bits of both strands of window j without building the keys (valid when L >= 8:
// the first eight bases of `left` are the first eight bases of the strand's window)
__device__ __forceinline__ bool window_top16(u64 c0, u64 c1, u32 b0, u32 b1, int j, int k, u32& tf, u32& tr) {
    u64 x = j ? ((c0 << (2 * j)) | (c1 >> (64 - 2 * j))) : c0;
    u32 bm = j ? ((b0 << j) | (b1 >> (32 - j))) : b0;
    if ((bm >> (32 - k)) != 0) return false;
    tf = (u32)(x >> 48);
    u32 t = ~(u32)(x >> (64 - 2 * k)) & 0xFFFFu;   // complement of the last eight bases of the window
    t = __brev(t) >> 16;                            // reverse the 16 bits ...
    tr = ((t & 0xAAAAu) >> 1) | ((t & 0x5555u) << 1);   // ... and restore the order inside each base
    return true;
}

// the slice test and the pass-1 digit from those 16 bits
__device__ __forceinline__ bool top16_in_slice(u32 t16, const Geom& g, u32& d1) {
    if (g.sbits && (t16 >> (16 - g.sbits)) != g.slice) return false;
    d1 = (t16 >> (8 - g.sbits)) & 0xFFu;
    return true;
}

template <int WIDE>
__global__ __launch_bounds__(P1_T) void k_hist8(const u64* __restrict__ codes, const u32* __restrict__ bad,
                                               u64 nwords, u32* __restrict__ partial8, Geom g) {
    __shared__ u32 lhist[256];
    if (threadIdx.x < 256) lhist[threadIdx.x] = 0;
    __syncthreads();
    u64 wpw = (nwords + NWG - 1) / NWG;
    u64 w0 = (u64)blockIdx.x * wpw;
    u64 w1 = w0 + wpw < nwords ? w0 + wpw : nwords;
    const bool cheap = g.L >= 8;
    for (u64 w = w0 + threadIdx.x; w < w1; w += P1_T) {
        u32 b0 = bad[w], b1 = bad[w + 1];
        if (b0 == 0xFFFFFFFFu) continue;
        if (WIDE == 1) {
            for (int j = 0; j < 32; j++) {
                u64 kf, kr;
                const u32 m = wide_keys(codes, bad, w * 32 + j, g, kf, kr);
                if ((m & 1) && slice_key(kf, g)) atomicAdd(&lhist[(u32)(kf >> 56)], 1u);
                if ((m & 2) && slice_key(kr, g)) atomicAdd(&lhist[(u32)(kr >> 56)], 1u);
            }
            continue;
        }
        u64 c0 = codes[w], c1 = codes[w + 1];
        if (WIDE == 2) {
#pragma unroll 4
            for (int j = 0; j < 32; j++) {
                u64 k1;
                if (window_key_single(c0, c1, b0, b1, j, g, k1) && slice_key(k1, g))
                    atomicAdd(&lhist[(u32)(k1 >> 56)], 1u);
            }
            continue;
        }
        if (cheap) {
#pragma unroll 8
            for (int j = 0; j < 32; j++) {
                u32 tf, tr, d1;
                if (!window_top16(c0, c1, b0, b1, j, g.k, tf, tr)) continue;
                if (top16_in_slice(tf, g, d1)) atomicAdd(&lhist[d1], 1u);
                if (top16_in_slice(tr, g, d1)) atomicAdd(&lhist[d1], 1u);
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < 32; j++) {
                u64 kf, kr;
                if (!window_keys(c0, c1, b0, b1, j, g, kf, kr)) continue;
                if (slice_key(kf, g)) atomicAdd(&lhist[(u32)(kf >> 56)], 1u);
                if (slice_key(kr, g)) atomicAdd(&lhist[(u32)(kr >> 56)], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) partial8[(u64)blockIdx.x * 256 + threadIdx.x] = lhist[threadIdx.x];
}

// ----------------------------------------------------------------------------
// block-wide exclusive scan helper (blockDim.x <= 1024, multiple of 64)
// ----------------------------------------------------------------------------
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* lds_waves /* >= 17 u32 */, u32& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) lds_waves[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int i = 0; i < nw; i++) { u32 t = lds_waves[i]; lds_waves[i] = run; run += t; }
        lds_waves[16] = run;
    }
    __syncthreads();
    u32 res = lds_waves[wave] + x - v;
    total = lds_waves[16];
    __syncthreads();
    return res;
}

// ----------------------------------------------------------------------------
// K2b  (single workgroup, 256 threads) thread d walks column d of partial8[NWG][256]:
// in place exclusive prefix over the workgroups (= that workgroup's private pass-1
// cursor offset inside bucket d), then a scan over d gives the bucket bases.
// ----------------------------------------------------------------------------
// (a) one workgroup per top-byte column d: exclusive prefix over the NWG persistent
//     workgroups, in place (= each workgroup's private pass-1 cursor offset), column total out
__global__ __launch_bounds__(256) void k_reduce8a(u32* __restrict__ partial8, u32* __restrict__ tot) {
    __shared__ u32 waves[17];
    const u32 d = blockIdx.x;
    const u32 PER = NWG / 256;
    u32 v[PER];
    u32 sum = 0;
#pragma unroll
    for (u32 q = 0; q < PER; q++) {
        v[q] = partial8[(u64)(threadIdx.x * PER + q) * 256 + d];
        sum += v[q];
    }
    u32 total;
    u32 ex = block_excl_scan(sum, waves, total);
#pragma unroll
    for (u32 q = 0; q < PER; q++) {
        partial8[(u64)(threadIdx.x * PER + q) * 256 + d] = ex;
        ex += v[q];
    }
    if (threadIdx.x == 0) tot[d] = total;
}

// (b) bucket bases and the pass-2 tile table (tiles never straddle a bucket: tp[d] = first
//     tile of bucket d, tiledesc[tile] = its key range; pre-zeroed, unused tiles are empty)
__global__ __launch_bounds__(1024) void k_reduce8b(const u32* __restrict__ tot, u32* __restrict__ base1,
                                                   u32* __restrict__ tp, uint2* __restrict__ tiledesc) {
    __shared__ u32 waves[17];
    __shared__ u32 sb[3][256];
    const u32 d = threadIdx.x & 255, part = threadIdx.x >> 8;
    const u32 colsum = part == 0 ? tot[d] : 0;
    u32 total;
    u32 ex = block_excl_scan(colsum, waves, total);
    const u32 ntile = part == 0 ? (colsum + P2_TILE - 1) / P2_TILE : 0;
    u32 ttotal;
    u32 t0 = block_excl_scan(ntile, waves, ttotal);
    if (part == 0) {
        base1[d] = ex;
        if (d == 0) base1[256] = total;
        tp[d] = t0;
        if (d == 0) tp[256] = ttotal;
        sb[0][d] = ex;
        sb[1][d] = colsum;
        sb[2][d] = t0;
    }
    __syncthreads();
    for (u32 dd = part; dd < 256; dd += 4) {
        const u32 bex = sb[0][dd], brun = sb[1][dd], bt0 = sb[2][dd];
        const u32 nt = (brun + P2_TILE - 1) / P2_TILE;
        for (u32 t = d; t < nt; t += 256) {
            u32 ts = bex + t * P2_TILE;
            tiledesc[bt0 + t] = make_uint2(ts, min(bex + brun, ts + P2_TILE));
        }
    }
}

// ----------------------------------------------------------------------------
// generic single-workgroup exclusive scan: out[0..n] (n+1 entries), out[n] = total
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
    __shared__ u32 waves[17];
    // thread-contiguous slices of a multiple of 4 elements (in / out are 16-byte aligned)
    const u32 per = ((n + 1023) / 1024 + 3) & ~3u;
    const u32 s0 = threadIdx.x * per;
    const u32 s1 = s0 + per < n ? s0 + per : n;
    u32 s = 0;
    for (u32 i = s0; i < s1; i += 4) {
        if (i + 4 <= n) {
            uint4 v = *reinterpret_cast<const uint4*>(in + i);
            s += v.x + v.y + v.z + v.w;
        } else {
            for (u32 q = i; q < s1; q++) s += in[q];
        }
    }
    u32 total;
    u32 ex = block_excl_scan(s, waves, total);
    for (u32 i = s0; i < s1; i += 4) {
        if (i + 4 <= n) {
            uint4 v = *reinterpret_cast<const uint4*>(in + i);
            uint4 o;
            o.x = ex; ex += v.x;
            o.y = ex; ex += v.y;
            o.z = ex; ex += v.z;
            o.w = ex; ex += v.w;
            *reinterpret_cast<uint4*>(out + i) = o;
        } else {
            for (u32 q = i; q < s1; q++) { u32 v = in[q]; out[q] = ex; ex += v; }
        }
    }
    if (threadIdx.x == 0) out[n] = total;
}

// ----------------------------------------------------------------------------
// K3  pass 1: partition by the top 8 bits, LDS-staged so that global stores are
// coalesced runs.  Same word ranges as k_hist8, so the workgroup's private cursors
// (base1[d] + its column prefix) are exact: no global atomics.  Per tile of 64 code
// words: generate <= 4096 keys, rank them per digit with LDS atomics, scan the 256
// digit counts, stage the keys digit-sorted in LDS, copy the runs out.
// ----------------------------------------------------------------------------
template <int WIDE>
__global__ __launch_bounds__(P1_T, P1_OCC) void k_scatter1(const u64* __restrict__ codes, const u32* __restrict__ bad,
                                                  u64 nwords, const u32* __restrict__ base1,
                                                  const u64* __restrict__ base64,
                                                  const u32* __restrict__ rowoff, u64* __restrict__ dst, Geom g) {
    __shared__ __attribute__((aligned(16))) u64 stage[P1_STAGE];
    __shared__ u64 cur[256];      // (64 bit: the slice pre-partition of a >= 2 Gbp genome exceeds 2^32 keys)
    __shared__ u32 cnt[256];
    __shared__ u64 delta[256];
    __shared__ u32 waves[17];
    const u32 tid = threadIdx.x;
    if (tid < 256) cur[tid] = (base64 ? base64[tid] : (u64)base1[tid]) + rowoff[(u64)blockIdx.x * 256 + tid];
    u64 wpw = (nwords + NWG - 1) / NWG;
    u64 w0 = (u64)blockIdx.x * wpw;
    u64 w1 = w0 + wpw < nwords ? w0 + wpw : nwords;
    for (u64 wt = w0; wt < w1; wt += P1_WORDS) {
        const u64 w = wt + tid / P1_TPW;
        const int j0 = (tid % P1_TPW) * P1_PPT;
        u64 key[P1_KPT];
        u32 r[P1_KPT];
        u32 vm = 0;
        if (tid < 256) cnt[tid] = 0;
        if (w < w1) {
            u32 b0 = bad[w], b1 = bad[w + 1];
            if (b0 != 0xFFFFFFFFu) {
                if (WIDE == 1) {
#pragma unroll
                    for (int jj = 0; jj < P1_PPT; jj++) {
                        u64 kf, kr;
                        const u32 m = wide_keys(codes, bad, w * 32 + j0 + jj, g, kf, kr);
                        if ((m & 1) && slice_key(kf, g)) { key[2 * jj] = kf; vm |= 1u << (2 * jj); }
                        if ((m & 2) && slice_key(kr, g)) { key[2 * jj + 1] = kr; vm |= 2u << (2 * jj); }
                    }
                } else if (WIDE == 2) {
                    u64 c0 = codes[w], c1 = codes[w + 1];
#pragma unroll
                    for (int jj = 0; jj < P1_PPT; jj++) {
                        u64 k1;
                        if (window_key_single(c0, c1, b0, b1, j0 + jj, g, k1) && slice_key(k1, g)) {
                            key[2 * jj] = k1;
                            vm |= 1u << (2 * jj);
                        }
                    }
                } else {
                    u64 c0 = codes[w], c1 = codes[w + 1];
#pragma unroll
                    for (int jj = 0; jj < P1_PPT; jj++) {
                        u64 kf, kr;
                        if (window_keys(c0, c1, b0, b1, j0 + jj, g, kf, kr)) {
                            if (slice_key(kf, g)) { key[2 * jj] = kf; vm |= 1u << (2 * jj); }
                            if (slice_key(kr, g)) { key[2 * jj + 1] = kr; vm |= 2u << (2 * jj); }
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P1_KPT; q++)
            if ((vm >> q) & 1) r[q] = atomicAdd(&cnt[(u32)(key[q] >> 56)], 1u);
        __syncthreads();
        u32 c = tid < 256 ? cnt[tid] : 0, total;
        u32 ex = block_excl_scan(c, waves, total);
        if (tid < 256) {
            cnt[tid] = ex;
            delta[tid] = cur[tid] - ex;
            cur[tid] += c;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P1_KPT; q++)
            if ((vm >> q) & 1) stage[cnt[(u32)(key[q] >> 56)] + r[q]] = key[q];
        __syncthreads();
        for (u32 p = tid; p < total; p += P1_T) {
            u64 k2 = stage[p];
            dst[p + delta[(u32)(k2 >> 56)]] = k2;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------
// K3k  sliced genomes: "pass 0" partitions ALL keys of a genome once by their top byte
// (k_hist8 / k_scatter1 on absolute keys, 64-bit bucket bases from k_bases64) -- a slice is a
// run of 2^(8 - sbits) such buckets, i.e. one contiguous region -- and every slice then takes
// its pass 1 from that region instead of regenerating all windows of the genome:
// k_hist8k / k_scatter1k = k_hist8 / k_scatter1 with keys read (and made relative) instead of
// generated.  Same workgroup ranges in both, so the private cursors stay exact.
// ----------------------------------------------------------------------------
__global__ void k_bases64(const u32* __restrict__ tot, u64* __restrict__ base64) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        u64 run = 0;
        for (int d = 0; d < 256; d++) { base64[d] = run; run += tot[d]; }
        base64[256] = run;
    }
}

#define P1K_TILE (P1_T * P1_KPT)      // keys per tile of the from-keys kernels (= P1_STAGE)

__global__ __launch_bounds__(P1_T) void k_hist8k(const u64* __restrict__ src, u64 n, int sbits,
                                                u32* __restrict__ partial8) {
    __shared__ u32 lhist[256];
    if (threadIdx.x < 256) lhist[threadIdx.x] = 0;
    __syncthreads();
    const u64 per = ((n + NWG - 1) / NWG + P1K_TILE - 1) / P1K_TILE * P1K_TILE;
    const u64 k0 = (u64)blockIdx.x * per;
    const u64 k1 = k0 + per < n ? k0 + per : n;
    for (u64 i = k0 + threadIdx.x; i < k1; i += P1_T) atomicAdd(&lhist[(u32)((src[i] << sbits) >> 56)], 1u);
    __syncthreads();
    if (threadIdx.x < 256) partial8[(u64)blockIdx.x * 256 + threadIdx.x] = lhist[threadIdx.x];
}

__global__ __launch_bounds__(P1_T) void k_scatter1k(const u64* __restrict__ src, u64 n, int sbits,
                                                   const u32* __restrict__ base1, const u32* __restrict__ rowoff,
                                                   u64* __restrict__ dst) {
    __shared__ __attribute__((aligned(16))) u64 stage[P1_STAGE];
    __shared__ u32 cur[256];
    __shared__ u32 cnt[256];
    __shared__ u32 delta[256];
    __shared__ u32 waves[17];
    const u32 tid = threadIdx.x;
    if (tid < 256) cur[tid] = base1[tid] + rowoff[(u64)blockIdx.x * 256 + tid];
    const u64 per = ((n + NWG - 1) / NWG + P1K_TILE - 1) / P1K_TILE * P1K_TILE;
    const u64 k0 = (u64)blockIdx.x * per;
    const u64 k1 = k0 + per < n ? k0 + per : n;
    for (u64 t0 = k0; t0 < k1; t0 += P1K_TILE) {
        u64 key[P1_KPT];
        u32 r[P1_KPT];
        u32 vm = 0;
        if (tid < 256) cnt[tid] = 0;
#pragma unroll
        for (int q = 0; q < P1_KPT; q++) {
            const u64 i = t0 + (u64)q * P1_T + tid;
            if (i < k1) { key[q] = src[i] << sbits; vm |= 1u << q; }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P1_KPT; q++)
            if ((vm >> q) & 1) r[q] = atomicAdd(&cnt[(u32)(key[q] >> 56)], 1u);
        __syncthreads();
        u32 c = tid < 256 ? cnt[tid] : 0, total;
        u32 ex = block_excl_scan(c, waves, total);
        if (tid < 256) {
            cnt[tid] = ex;
            delta[tid] = cur[tid] - ex;
            cur[tid] += c;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P1_KPT; q++)
            if ((vm >> q) & 1) stage[cnt[(u32)(key[q] >> 56)] + r[q]] = key[q];
        __syncthreads();
        for (u32 p = tid; p < total; p += P1_T) {
            u64 k2 = stage[p];
            dst[p + delta[(u32)(k2 >> 56)]] = k2;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------
// K4  pass 2 = segmented partition of the pass-1 buckets by the next b-8 bits, in
// tiles of P2_TILE keys that never straddle a bucket (tp / tiledesc from k_reduce8b):
//   k_hist2    per-tile digit counts (LDS histogram)           -> tilehist[tile][bin]
//   k_scan2    per bucket: bin totals -> fine offsets off[], per-tile bin bases (in place)
//   k_scatter2 per tile: rank, scan, stage digit-sorted in LDS, coalesced runs out
// No global atomics, any number of workgroups per CU.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(P2_T) void k_hist2(const u64* __restrict__ src, const uint2* __restrict__ tiledesc,
                                               u32* __restrict__ tilehist, int b) {
    __shared__ u32 h[1024];
    const u32 tile = blockIdx.x;
    const uint2 td = tiledesc[tile];
    const u32 s = td.x, e = td.y;
    if (s >= e) return;
    const u32 nb2 = 1u << (b - 8);
    const int rb = 64 - b;
    for (u32 i = threadIdx.x; i < nb2; i += P2_T) h[i] = 0;
    __syncthreads();
    u64 key[P2_KPT];
#pragma unroll
    for (int q = 0; q < (int)P2_KPT; q++) {
        u32 i = s + q * P2_T + threadIdx.x;
        key[q] = i < e ? src[i] : 0;
    }
#pragma unroll
    for (int q = 0; q < (int)P2_KPT; q++) {
        u32 i = s + q * P2_T + threadIdx.x;
        if (i < e) atomicAdd(&h[(u32)(key[q] >> rb) & (nb2 - 1)], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nb2; i += P2_T) tilehist[(u64)tile * nb2 + i] = h[i];
}

__global__ __launch_bounds__(1024) void k_scan2(u32* __restrict__ tilehist, const u32* __restrict__ base1,
                                                const u32* __restrict__ tp, u32* __restrict__ off, int b) {
    __shared__ u32 waves[17];
    const u32 nb2 = 1u << (b - 8);
    const u32 d1 = blockIdx.x, bin = threadIdx.x;
    const u32 t0 = tp[d1], t1 = tp[d1 + 1];
    u32 tot = 0;
    if (bin < nb2) {
        u32 t = t0;
        for (; t + 8 <= t1; t += 8) {
            u32 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = tilehist[(u64)(t + q) * nb2 + bin];
#pragma unroll
            for (int q = 0; q < 8; q++) tot += v[q];
        }
        for (; t < t1; t++) tot += tilehist[(u64)t * nb2 + bin];
    }
    u32 total;
    u32 ex = block_excl_scan(tot, waves, total);
    u32 run = base1[d1] + ex;
    if (bin < nb2) {
        off[d1 * nb2 + bin] = run;
        u32 t = t0;
        for (; t + 8 <= t1; t += 8) {
            u32 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = tilehist[(u64)(t + q) * nb2 + bin];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                tilehist[(u64)(t + q) * nb2 + bin] = run;
                run += v[q];
            }
        }
        for (; t < t1; t++) {
            u32 v = tilehist[(u64)t * nb2 + bin];
            tilehist[(u64)t * nb2 + bin] = run;
            run += v;
        }
    }
    if (d1 == 255 && bin == 0) off[256u * nb2] = base1[256];
}

__global__ __launch_bounds__(P2_T) void k_scatter2(const u64* __restrict__ src, u64* __restrict__ dst,
                                                   const uint2* __restrict__ tiledesc,
                                                   const u32* __restrict__ tilehist, int b) {
    __shared__ __attribute__((aligned(16))) u64 stage[P2_TILE];
    __shared__ u32 cnt[1024];
    __shared__ u32 delta[1024];
    __shared__ u32 waves[17];
    // tiles are walked from the last to the first: k_hist2 has just streamed the same keys in
    // ascending order, so its tail is what the 256 MiB Infinity Cache still holds; and the
    // fine buckets written last (low ones) are the ones k_localsort reads first
    const u32 tile = gridDim.x - 1 - blockIdx.x;
    const uint2 td = tiledesc[tile];
    const u32 s = td.x, e = td.y;
    if (s >= e) return;
    const u32 tid = threadIdx.x;
    const u32 nb2 = 1u << (b - 8);
    const int rb = 64 - b;
    u64 key[P2_KPT];
    u32 r[P2_KPT];
#pragma unroll
    for (int q = 0; q < (int)P2_KPT; q++) {
        u32 i = s + q * P2_T + tid;
        key[q] = i < e ? src[i] : 0;
    }
    // this thread scans bins [2 tid, 2 tid + 2) (nb2 <= 1024 = 2 * P2_T)
    const u32 bin0 = 2 * tid, bin1 = 2 * tid + 1;
    u32 tb0 = bin0 < nb2 ? tilehist[(u64)tile * nb2 + bin0] : 0;
    u32 tb1 = bin1 < nb2 ? tilehist[(u64)tile * nb2 + bin1] : 0;
    cnt[bin0] = 0;
    cnt[bin1] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (int)P2_KPT; q++) {
        u32 i = s + q * P2_T + tid;
        if (i < e) r[q] = atomicAdd(&cnt[(u32)(key[q] >> rb) & (nb2 - 1)], 1u);
    }
    __syncthreads();
    u32 c0 = cnt[bin0], c1 = cnt[bin1], total;
    u32 ex = block_excl_scan(c0 + c1, waves, total);
    cnt[bin0] = ex;
    cnt[bin1] = ex + c0;
    delta[bin0] = tb0 - ex;
    delta[bin1] = tb1 - (ex + c0);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (int)P2_KPT; q++) {
        u32 i = s + q * P2_T + tid;
        if (i < e) stage[cnt[(u32)(key[q] >> rb) & (nb2 - 1)] + r[q]] = key[q];
    }
    __syncthreads();
    const u32 nt = e - s;
#pragma unroll
    for (int q = 0; q < (int)P2_KPT; q++) {
        u32 p = q * P2_T + tid;
        if (p < nt) {
            u64 k2 = stage[p];
            dst[p + delta[(u32)(k2 >> rb) & (nb2 - 1)]] = k2;
        }
    }
}

// ----------------------------------------------------------------------------
// K4b  chunk table: chunkstart[j] = first bucket whose start offset is >= j*T
// (pre-filled with nbuckets).  Chunk j = buckets [chunkstart[j], chunkstart[j+1]).
// ----------------------------------------------------------------------------
__global__ void k_chunk_bounds(const u32* __restrict__ off, u32 nb, u32* __restrict__ chunkstart) {
    u32 f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > nb) return;
    u32 cur = off[f];
    u32 jlo = f == 0 ? 0 : off[f - 1] / LS_T + 1;
    u32 jhi = cur / LS_T;
    for (u32 j = jlo; j <= jhi; j++) chunkstart[j] = f;
}

// chunk descriptors {first key, end key, first bucket, end bucket}: one 16-byte load per chunk
// instead of a chain of three dependent ones
__global__ void k_chunk_desc(const u32* __restrict__ off, const u32* __restrict__ chunkstart, u32 nchunks,
                             uint4* __restrict__ desc) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nchunks) return;
    u32 lo = chunkstart[j], hi = chunkstart[j + 1];
    uint4 d = make_uint4(0, 0, 0, 0);
    if (lo < hi) d = make_uint4(off[lo], off[hi], lo, hi);
    desc[j] = d;
}

// property check for full-size runs: number of adjacent pairs out of order
__global__ void k_count_inversions(const u64* __restrict__ keys, u64 n, u64* __restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    u32 bad = 0;
    for (; i + 1 < n; i += stride) bad += keys[i] > keys[i + 1];
    if (bad) atomicAdd(out, (u64)bad);
}

// fine bucket offsets of an already sorted key array: off[f] = lower_bound(f << rb)
__global__ void k_offsets_from_sorted(const u64* __restrict__ keys, u32 n, u32 nb, int rb, u32* __restrict__ off) {
    u32 f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > nb) return;
    if (f == nb) { off[f] = n; return; }
    const u64 target = (u64)f << rb;
    u32 l = 0, r = n;
    while (l < r) {
        u32 mid = l + ((r - l) >> 1);
        if (keys[mid] < target) l = mid + 1; else r = mid;
    }
    off[f] = l;
}

// ----------------------------------------------------------------------------
// K5  local sort of one chunk (<= LS_CAP keys) in LDS, in place in global memory:
// count into 4096 order-preserving sub-bins (16-bit LDS counters), scan, place,
// then rank every key inside its sub-bin by counting; a chunk with a crowded
// sub-bin (duplicates, skew) runs a bitonic network instead.  A chunk that does
// not fit reports its (single) oversized bucket for the global fallback.
// ----------------------------------------------------------------------------
__device__ __forceinline__ u32 cnt16_get(const u32* c, u32 i) {
    u32 w = c[i >> 1];
    return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
}

__device__ __forceinline__ void ls_load(const u64* __restrict__ keys, uint4 d, u64 (&k)[LS_PER]) {
    const u32 m = min(d.y - d.x, LS_CAP);
#pragma unroll
    for (int i = 0; i < (int)LS_PER; i++) {
        u32 p = threadIdx.x + i * LS_THREADS;
        k[i] = p < m ? keys[d.x + p] : 0;
    }
}

// Persistent: grid = resident workgroups; each walks chunks j, j+G, ... with the NEXT chunk's
// keys (and the descriptor after that) already in flight while the current one is sorted.
__global__ __launch_bounds__(LS_THREADS) void k_localsort(u64* __restrict__ keys,
                                                          const u32* __restrict__ off,
                                                          const uint4* __restrict__ desc, u32 nchunks, int b,
                                                          u32* __restrict__ ovf_count,
                                                          uint4* __restrict__ ovf_list, int dbg) {
    __shared__ __attribute__((aligned(16))) u64 S[LS_CAP];
    __shared__ u32 cnt[LS_NB / 2];
    __shared__ u32 waves[17];
    __shared__ u32 s_maxbin;
    const u32 tid = threadIdx.x;
    const u32 G = gridDim.x;
    const int rb = 64 - b;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    u32 j = blockIdx.x;
    uint4 dc = j < nchunks ? desc[j] : zero4;
    uint4 dn = j + G < nchunks ? desc[j + G] : zero4;
    u64 key[LS_PER], nkey[LS_PER];
    ls_load(keys, dc, key);
    for (; j < nchunks; j += G) {
        const uint4 dnn = j + 2 * G < nchunks ? desc[j + 2 * G] : zero4;
        ls_load(keys, dn, nkey);
        u32 s = dc.x, e = dc.y, lo = dc.z, hi = dc.w;
        u32 m = e - s;
        if (m > LS_CAP) {           // the chunk's last bucket is oversized: global fallback sorts it
            u32 last = hi - 1;
            u32 ls = off[last];
            if (tid == 0) {
                u32 idx = atomicAdd(ovf_count, 1u);
                if (idx < OVF_MAX) ovf_list[idx] = make_uint4(ls, e, last, 0);
            }
            hi = last;
            e = ls;
            m = e - s;
        }
        if (m > 0 && (dbg & 64)) {          // kr_debug_localsort: stream the chunk through, nothing else
#pragma unroll
            for (int i = 0; i < (int)LS_PER; i++) {
                u32 p = tid + i * LS_THREADS;
                if (p < m) keys[s + p] = key[i];
            }
        } else if (m > 0) {
            const u32 nbk = hi - lo;
            const int clog = nbk <= 1 ? 0 : 32 - __clz((int)(nbk - 1));
            const int sh = rb + clog - LS_NB_LOG;
            const u64 keylo = (u64)lo << rb;
            for (u32 i = tid; i < LS_NB / 2; i += LS_THREADS) cnt[i] = 0;
            if (tid == 0) s_maxbin = 0;
            __syncthreads();
            u32 sub[LS_PER], r[LS_PER];
#pragma unroll
            for (int i = 0; i < (int)LS_PER; i++) {
                u32 p = tid + i * LS_THREADS;
                if (p < m) {
                    sub[i] = (u32)((key[i] - keylo) >> sh);
                    u32 old = atomicAdd(&cnt[sub[i] >> 1], (sub[i] & 1) ? 0x10000u : 1u);
                    r[i] = (sub[i] & 1) ? (old >> 16) : (old & 0xFFFFu);
                }
            }
            __syncthreads();
            // exclusive scan of the 4096 16-bit counters: thread t owns LS_WPT consecutive words
            {
                u32 w[LS_WPT];
                u32 sum = 0, mx = 0;
#pragma unroll
                for (int q = 0; q < LS_WPT; q++) {
                    w[q] = cnt[tid * LS_WPT + q];
                    u32 a = w[q] & 0xFFFFu, c2 = w[q] >> 16;
                    sum += a + c2;
                    mx = max(mx, max(a, c2));
                }
                if (mx > LS_BIN_LIMIT) atomicMax(&s_maxbin, mx);
                u32 total;
                u32 ex = block_excl_scan(sum, waves, total);
#pragma unroll
                for (int q = 0; q < LS_WPT; q++) {
                    u32 a = w[q] & 0xFFFFu, c2 = w[q] >> 16;
                    u32 lo16 = ex;
                    ex += a;
                    u32 hi16 = ex;
                    ex += c2;
                    cnt[tid * LS_WPT + q] = lo16 | (hi16 << 16);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < (int)LS_PER; i++) {
                u32 p = tid + i * LS_THREADS;
                if (p < m) S[cnt16_get(cnt, sub[i]) + r[i]] = key[i];
            }
            __syncthreads();
            if (dbg & 128) {                // kr_debug_localsort: binned but not ranked
                for (u32 p = tid; p < m; p += LS_THREADS) keys[s + p] = S[p];
            } else if (s_maxbin <= LS_BIN_LIMIT) {
                // keys are walked in PLACED order: lanes of a wave rank neighbours, so their LDS reads
                // are adjacent and their global stores fall into one or two 512-byte runs.  (Measured,
                // tools/ls_ablate.py: load + store alone 0.34 ms per 10^8 keys, binning adds nothing,
                // this step 0.17 ms -- all of it the divergent loop below (storing in placed order
                // instead costs the same); staging the ranked keys through LDS, fewer barriers,
                // unconditional neighbour reads, ranking through wave shuffles (halo lanes), 8192
                // sub-bins, 1024 threads / higher occupancy and non-temporal stores were all
                // A/B-tested: no gain.)
#pragma unroll 2
                for (int i = 0; i < (int)LS_PER; i++) {
                    u32 p = tid + i * LS_THREADS;
                    if (p < m) {
                        u64 kk = S[p];
                        u32 sb = (u32)((kk - keylo) >> sh);
                        u32 b0 = cnt16_get(cnt, sb);
                        u32 b1 = sb + 1 < LS_NB ? cnt16_get(cnt, sb + 1) : m;
                        u32 rank = b0;
                        for (u32 q = b0; q < b1; q++) {
                            u64 kq = S[q];
                            rank += (kq < kk) || (kq == kk && q < p);
                        }
                        keys[s + ((dbg & 256) ? p : rank)] = kk;    // (256: kr_debug_localsort, ranked but stored in place)
                    }
                }
            } else {
                u32 np = 1;
                while (np < m) np <<= 1;
                for (u32 p = m + tid; p < np; p += LS_THREADS) S[p] = ~0ull;
                __syncthreads();
                for (u32 kk = 2; kk <= np; kk <<= 1) {
                    for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                        for (u32 t = tid; t < np / 2; t += LS_THREADS) {
                            u32 i0 = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                            u32 i1 = i0 | jj;
                            bool up = (i0 & kk) == 0;
                            u64 a = S[i0], c2 = S[i1];
                            if ((a > c2) == up) { S[i0] = c2; S[i1] = a; }
                        }
                        __syncthreads();
                    }
                }
                for (u32 p = tid; p < m; p += LS_THREADS) keys[s + p] = S[p];
            }
            __syncthreads();      // S / cnt are reused by the next chunk
        }
        dc = dn;
        dn = dnn;
#pragma unroll
        for (int i = 0; i < (int)LS_PER; i++) key[i] = nkey[i];
    }
}

// ----------------------------------------------------------------------------
// Kb  fallback for fine buckets that do not fit the LDS sort (poly-A, satellites, tandem
// repeats): their 4096-key tiles are sorted by k_localsort itself (descriptor list of
// tiles), then merged pairwise -- run length doubling each round -- by a merge-path kernel:
// one workgroup per 2048 output keys finds its two input ranges by binary search, stages
// them in LDS, every thread merges 8 outputs.  Source -> scratch, then copied back.
// segs[i] = {start, end, first global tile of the segment, 0}
// ----------------------------------------------------------------------------
#define MG_T 256
#define MG_VT 8
#define MG_TILE (MG_T * MG_VT)

__device__ __forceinline__ u32 merge_path(const u64* A, u32 na, const u64* B, u32 nb, u32 diag) {
    u32 lo = diag > nb ? diag - nb : 0, hi = diag < na ? diag : na;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (A[mid] <= B[diag - mid - 1]) lo = mid + 1; else hi = mid;
    }
    return lo;      // number of A elements among the first `diag` merged (ties: A first)
}

__global__ __launch_bounds__(MG_T) void k_seg_merge(const u64* __restrict__ src, u64* __restrict__ dst,
                                                    const uint4* __restrict__ segs, u32 nseg, u32 run) {
    __shared__ __attribute__((aligned(16))) u64 L[MG_TILE];
    __shared__ u32 sh[4];
    const u32 tile = blockIdx.x;
    u32 lo = 0, hi = nseg;                      // last segment whose first tile is <= tile
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (segs[mid].z <= tile) lo = mid; else hi = mid;
    }
    const uint4 sg = segs[lo];
    const u32 len = sg.y - sg.x;
    const u32 o0 = (tile - sg.z) * MG_TILE;
    if (o0 >= len) return;
    const u32 o1 = min(o0 + MG_TILE, len);
    const u32 base = (o0 / (2 * run)) * (2 * run);
    const u32 amid = min(base + run, len), bend = min(base + 2 * run, len);
    const u64* A = src + sg.x + base;
    const u64* B = src + sg.x + amid;
    const u32 na = amid - base, nb = bend - amid;
    if (threadIdx.x == 0) {
        sh[0] = merge_path(A, na, B, nb, o0 - base);
        sh[1] = merge_path(A, na, B, nb, o1 - base);
    }
    __syncthreads();
    const u32 a0 = sh[0], a1 = sh[1];
    const u32 b0 = (o0 - base) - a0, b1 = (o1 - base) - a1;
    const u32 ca = a1 - a0, cb = b1 - b0;
    for (u32 i = threadIdx.x; i < ca; i += MG_T) L[i] = A[a0 + i];
    for (u32 i = threadIdx.x; i < cb; i += MG_T) L[ca + i] = B[b0 + i];
    __syncthreads();
    const u64* LA = L;
    const u64* LB = L + ca;
    const u32 total = ca + cb;
    const u32 d = min(threadIdx.x * MG_VT, total);
    u32 ia = merge_path(LA, ca, LB, cb, d);
    u32 ib = d - ia;
    u64* out = dst + sg.x + o0;
#pragma unroll
    for (int q = 0; q < MG_VT; q++) {
        u32 o = d + q;
        if (o < total) {
            bool takeA = ib >= cb || (ia < ca && LA[ia] <= LB[ib]);
            out[o] = takeA ? LA[ia] : LB[ib];
            ia += takeA;
            ib += !takeA;
        }
    }
}

__global__ void k_seg_copy(const u64* __restrict__ src, u64* __restrict__ dst, const uint4* __restrict__ segs,
                           u32 nseg) {
    const u32 tile = blockIdx.x;
    u32 lo = 0, hi = nseg;
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (segs[mid].z <= tile) lo = mid; else hi = mid;
    }
    const uint4 sg = segs[lo];
    const u32 len = sg.y - sg.x;
    const u32 o0 = (tile - sg.z) * MG_TILE;
    for (u32 i = o0 + threadIdx.x; i < min(o0 + MG_TILE, len); i += blockDim.x) dst[sg.x + i] = src[sg.x + i];
}

// ----------------------------------------------------------------------------
// ordered compaction inside a workgroup: position of this thread's flagged item
// ----------------------------------------------------------------------------
__device__ __forceinline__ u32 block_compact(bool flag, u32* lds_waves, u32& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u64 mask = __ballot(flag);
    u32 pos = __popcll(mask & ((1ull << lane) - 1));
    if (lane == 0) lds_waves[wave] = __popcll(mask);
    __syncthreads();
    u32 basev = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        u32 t = lds_waves[i];
        if (i < wave) basev += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return basev + pos;
}

__device__ __forceinline__ u64 diag_mask(u64 key, int LR, int D) {
    u64 m = 0;
    for (int c = 0; c < D; c++) {
        u32 bb = (u32)(key >> (62 - 2 * (LR + c))) & 3u;
        m |= 1ull << (4 * c + bb);
    }
    return m;
}

__device__ __forceinline__ bool passes_filter(u64 im, u64 om, int D) {
    u64 x = im & om;
    for (int c = 0; c < D; c++)
        if (((x >> (4 * c)) & 15ull) == 0) return true;
    return false;
}

// ----------------------------------------------------------------------------
// K6  n-way intersection.  One workgroup per chunk of the anchor genome: the
// distinct (left,right) prefixes of the chunk go to LDS; every genome (the
// anchor included) streams its keys of the same bucket range past them
// (order-preserving sub-bins in LDS: ~2 probes per key), OR-ing a presence bit and the diagnostic-column masks.
// Survivors are written in order to tmp[anchor offset ...]; chunkcnt[j] = count.
// ----------------------------------------------------------------------------
#define MAXG 32
struct IsectArgs {
    const u64* keys[MAXG];
    const u32* off[MAXG];
    u32 ingroup_bits;
    int n;
    int anchor;
    const uint4* chunkdesc;
    kr_cand* tmp;
    u32* chunkcnt;
    int apply_filter;
    int dbg;   // KR_DBG env: 32 = always take the generic sub-tile path (tests cover both)
};

// per-head state in LDS, by mask format FMT:
//   0 compact (D <= 4)  one u64: present << 32 | out << 16 | in     -> one atomic per matching key
//   1 narrow  (D <= 8)  present u32 + one u64: out << 32 | in
//   2 wide    (D <= 16) present u32 + in u64 + out u64
template <int FMT>
__device__ __forceinline__ void isect_probe(u64 key, int gi, bool ing, u64 first, u64 last, int sh,
                                            const u64* heads, const unsigned short* binstart, u32* present,
                                            u64* inm, u64* outm, const Geom& g, int LR) {
    u64 pre = key & g.pmask;
    if (pre < first || pre > last) return;
    u32 sb = (u32)((pre - first) >> sh);
    u32 h = binstart[sb];
    const u32 hend = binstart[sb + 1];
    for (; h < hend; h++) {
        if (heads[h] == pre) {
            const u64 dm = g.D > 0 ? diag_mask(key, LR, g.D) : 0;
            if (FMT == 0) {
                atomicOr((u64*)&inm[h], ((u64)1 << (32 + gi)) | (ing ? dm : (dm << 16)));
            } else {
                atomicOr(&present[h], 1u << gi);
                if (g.D > 0) {
                    if (FMT == 2) atomicOr((u64*)(ing ? &inm[h] : &outm[h]), dm);
                    else atomicOr((u64*)&inm[h], ing ? dm : (dm << 32));
                }
            }
            return;
        }
    }
}

#define IS_APT (IS_SUB / IS_THREADS)    // anchor keys (and stream keys per batch) per thread
#define IS_NW (IS_THREADS / 64)

// ordered compaction of IS_APT flags per thread (element p = q * IS_THREADS + tid):
// ballots go to LDS, one wave turns the IS_APT * IS_NW group counts into prefixes.
// Returns the total; pos[q] = output slot of element q (valid where flag[q]).
__device__ __forceinline__ u32 compact_flags(const bool (&flag)[IS_APT], u32 (&pos)[IS_APT], u64* masks,
                                             u32* mpref) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 m[IS_APT];
#pragma unroll
    for (int q = 0; q < (int)IS_APT; q++) {
        m[q] = __ballot(flag[q]);
        if (lane == 0) masks[q * IS_NW + wave] = m[q];
    }
    __syncthreads();
    if (wave == 0) {
        const int ng = IS_APT * IS_NW;     // <= 64 groups
        u32 c = lane < ng ? (u32)__popcll(masks[lane]) : 0;
        u32 x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u32 y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane < ng) mpref[lane] = x - c;
        if (lane == 63) mpref[64] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (int)IS_APT; q++)
        pos[q] = mpref[q * IS_NW + wave] + (u32)__popcll(m[q] & ((1ull << lane) - 1));
    return mpref[64];
}

template <int FMT>
__global__ __launch_bounds__(IS_THREADS, FMT == 2 ? 4 : (FMT == 1 ? 6 : 8)) void k_intersect(IsectArgs a, Geom g) {
    constexpr bool WIDE = FMT == 2;
    constexpr u32 NBINS = FMT == 0 ? IS_NB / 2 : IS_NB;      // compact: 4 workgroups per CU fit
    constexpr int NB_LOG = FMT == 0 ? 11 : 12;
    __shared__ __attribute__((aligned(16))) u64 heads[IS_SUB];
    __shared__ __attribute__((aligned(16))) u64 inm[IS_SUB];
    __shared__ __attribute__((aligned(16))) u64 outm[WIDE ? IS_SUB : 1];
    __shared__ u32 present[FMT == 0 ? 1 : IS_SUB];
    __shared__ unsigned short binstart[NBINS + 2];
    __shared__ u64 masks[64];
    __shared__ u32 mpref[65];
    __shared__ u32 sstart[MAXG], send[MAXG];
    const u32 tid = threadIdx.x;
    const uint4 cd = a.chunkdesc[blockIdx.x];
    const u32 sa = cd.x, ea = cd.y, chi = cd.w;
    if (sa >= ea) {
        if (tid == 0) a.chunkcnt[blockIdx.x] = 0;
        return;
    }
    const u64* KA = a.keys[a.anchor];
    const u32* offA = a.off[a.anchor];
    const u32 full = a.n >= 32 ? 0xFFFFFFFFu : ((1u << a.n) - 1);
    const int LR = g.LRrel;
    const bool prefix_in_bucket = 2 * LR >= g.b;      // a (left,right) group never spans fine buckets
    const bool anchor_in = (a.ingroup_bits >> a.anchor) & 1;
    u32 nout = 0;
    u32 fcur = cd.z;                                   // first bucket of the current sub-tile
    bool start_ok = true;                              // the sub-tile starts on a bucket boundary
    for (u32 sub = sa; sub < ea;) {
        // ---- sub-tile = whole buckets, at most IS_SUB anchor keys (key-count split only for
        //      a single bucket larger than that)
        u32 fend = chi, subend = ea;
        bool aligned = true;
        if (ea - sub > IS_SUB) {
            u32 l = fcur, r = chi;                     // largest f with offA[f] <= sub + IS_SUB
            while (l < r) {
                u32 mid = (l + r + 1) >> 1;
                if (offA[mid] <= sub + IS_SUB) l = mid; else r = mid - 1;
            }
            if (l > fcur && offA[l] > sub) {
                fend = l;
                subend = offA[l];
            } else {
                aligned = false;
                fend = fcur;
                subend = sub + IS_SUB;
            }
        }
        const u32 cnt = subend - sub;
        const bool fast = start_ok && aligned && prefix_in_bucket && !(a.dbg & 32);
        // ---- every load we can issue now: anchor keys, and (fast) the stream ranges
        u64 ak[IS_APT], pk[IS_APT];
#pragma unroll
        for (int q = 0; q < (int)IS_APT; q++) {
            u32 p = q * IS_THREADS + tid;
            u32 gi = sub + p;
            ak[q] = p < cnt ? KA[gi] : 0;
            pk[q] = (p < cnt && gi > 0) ? KA[gi - 1] : 0;
        }
        if (fast && tid < (u32)a.n) {
            u32 s0 = a.off[tid][fcur], e0 = a.off[tid][fend];
            if ((int)tid == a.anchor) e0 = s0;                     // the anchor's own keys are in registers
            sstart[tid] = s0;
            send[tid] = e0;
        }
        // ---- distinct prefixes, in order
        bool flag[IS_APT];
        u32 pos[IS_APT];
#pragma unroll
        for (int q = 0; q < (int)IS_APT; q++) {
            u32 p = q * IS_THREADS + tid;
            flag[q] = p < cnt && ((sub + p == 0) || ((pk[q] & g.pmask) != (ak[q] & g.pmask)));
        }
        const u32 nheads = compact_flags(flag, pos, masks, mpref);
#pragma unroll
        for (int q = 0; q < (int)IS_APT; q++) {
            if (flag[q]) {
                heads[pos[q]] = ak[q] & g.pmask;
                if (FMT == 0) {
                    inm[pos[q]] = fast ? ((u64)1 << (32 + a.anchor)) : 0ull;
                } else {
                    present[pos[q]] = fast ? (1u << a.anchor) : 0u;
                    inm[pos[q]] = 0;
                    if (WIDE) outm[pos[q]] = 0;
                }
            }
        }
        __syncthreads();
        if (nheads > 0) {
            const u64 first = heads[0], last = heads[nheads - 1];
            const u64 span = last - first;
            const int sh = span < NBINS ? 0 : (64 - __clzll((long long)span) - NB_LOG);
            if (!fast && tid < (u32)a.n) {
                const u32 fl = (u32)(first >> g.rb);
                const u32 fh = (u32)((last | ~g.pmask) >> g.rb) + 1;
                u32 s0 = a.off[tid][fl];
                sstart[tid] = s0;
                send[tid] = a.off[tid][fh];
            }
            // order-preserving sub-bins over [first, last]: binstart[sb] = #heads with bin < sb
            for (u32 h = tid; h < nheads; h += IS_THREADS) {
                u32 sb = (u32)((heads[h] - first) >> sh);
                u32 prev = h ? (u32)((heads[h - 1] - first) >> sh) + 1 : 0;
                for (u32 q = prev; q <= sb; q++) binstart[q] = (unsigned short)h;
            }
            {
                u32 lastbin = (u32)(span >> sh);
                for (u32 q = lastbin + 1 + tid; q <= NBINS; q += IS_THREADS) binstart[q] = (unsigned short)nheads;
            }
            if (fast && g.D > 0) {
                // the anchor's own diagnostic columns, straight from registers
#pragma unroll
                for (int q = 0; q < (int)IS_APT; q++) {
                    u32 p = q * IS_THREADS + tid;
                    if (p < cnt) {
                        u32 h = flag[q] ? pos[q] : pos[q] - 1;
                        u64 dm = diag_mask(ak[q], LR, g.D);
                        if (FMT == 2) atomicOr((u64*)(anchor_in ? &inm[h] : &outm[h]), dm);
                        else if (FMT == 1) atomicOr((u64*)&inm[h], anchor_in ? dm : (dm << 32));
                        else atomicOr((u64*)&inm[h], anchor_in ? dm : (dm << 16));
                    }
                }
            }
            __syncthreads();
            // ---- stream the genomes past the heads; first batch of genome g+1 in flight while g is probed
            u64 cur[IS_APT], nxt[IS_APT];
            {
                const u64* K = a.keys[0];
                const u32 s0 = sstart[0], e0 = send[0];
#pragma unroll
                for (int q = 0; q < (int)IS_APT; q++) {
                    u32 i = s0 + q * IS_THREADS + tid;
                    cur[q] = i < e0 ? K[i] : 0;
                }
            }
            for (int gi = 0; gi < a.n; gi++) {
                const u64* K = a.keys[gi];
                const u32 s = sstart[gi], e = send[gi];
                const bool ing = (a.ingroup_bits >> gi) & 1;
                if (gi + 1 < a.n) {
                    const u64* K2 = a.keys[gi + 1];
                    const u32 s2 = sstart[gi + 1], e2 = send[gi + 1];
#pragma unroll
                    for (int q = 0; q < (int)IS_APT; q++) {
                        u32 i = s2 + q * IS_THREADS + tid;
                        nxt[q] = i < e2 ? K2[i] : 0;
                    }
                }
                for (u32 i0 = s; i0 < e; i0 += IS_SUB) {
                    if (i0 != s) {
#pragma unroll
                        for (int q = 0; q < (int)IS_APT; q++) {
                            u32 i = i0 + q * IS_THREADS + tid;
                            cur[q] = i < e ? K[i] : 0;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < (int)IS_APT; q++) {
                        u32 i = i0 + q * IS_THREADS + tid;
                        if (i < e)
                            isect_probe<FMT>(cur[q], gi, ing, first, last, sh, heads, binstart, present, inm, outm,
                                              g, LR);
                    }
                }
#pragma unroll
                for (int q = 0; q < (int)IS_APT; q++) cur[q] = nxt[q];
            }
            __syncthreads();
            // ---- survivors, in order
            u64 im[IS_APT], om[IS_APT];
#pragma unroll
            for (int q = 0; q < (int)IS_APT; q++) {
                u32 h = q * IS_THREADS + tid;
                flag[q] = false;
                im[q] = om[q] = 0;
                if (h < nheads) {
                    const u64 st = inm[h];
                    flag[q] = (FMT == 0 ? (u32)(st >> 32) : present[h]) == full;
                    im[q] = FMT == 2 ? st : (FMT == 1 ? (st & 0xFFFFFFFFull) : (st & 0xFFFFull));
                    om[q] = FMT == 2 ? outm[h] : (FMT == 1 ? (st >> 32) : ((st >> 16) & 0xFFFFull));
                    if (flag[q] && a.apply_filter && g.D > 0) flag[q] = passes_filter(im[q], om[q], g.D);
                }
            }
            const u32 nsurv = compact_flags(flag, pos, masks, mpref);
#pragma unroll
            for (int q = 0; q < (int)IS_APT; q++) {
                if (flag[q]) {
                    kr_cand c;
                    c.prefix = heads[q * IS_THREADS + tid];
                    c.in_mask = im[q];
                    c.out_mask = om[q];
                    a.tmp[(u64)sa + nout + pos[q]] = c;
                }
            }
            nout += nsurv;
            __syncthreads();
        }
        sub = subend;
        fcur = fend;
        start_ok = aligned;
    }
    if (tid == 0) a.chunkcnt[blockIdx.x] = nout;
}

// ----------------------------------------------------------------------------
// K6b  dense, ordered candidate list from the per-chunk runs
__global__ void k_gather_cands(int sbits, u32 slice, const kr_cand* __restrict__ tmp,
                               const uint4* __restrict__ chunkdesc, const u32* __restrict__ chunkcnt,
                               const u32* __restrict__ chunkpos, kr_cand* __restrict__ out) {
    u32 n = chunkcnt[blockIdx.x];
    if (n == 0) return;
    u64 src = chunkdesc[blockIdx.x].x;
    u32 dst = chunkpos[blockIdx.x];
    const u64 top = sbits ? ((u64)slice << (64 - sbits)) : 0;
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        kr_cand c = tmp[src + i];
        c.prefix = top | (c.prefix >> sbits);          // relative -> absolute prefix
        out[dst + i] = c;
    }
}

// ----------------------------------------------------------------------------
// K7  records of every candidate in one genome: distinct keys + multiplicities
// ----------------------------------------------------------------------------
__global__ void k_collect(const kr_cand* __restrict__ cands, u32 ncand, const u64* __restrict__ K,
                          const u32* __restrict__ off, Geom g, u32 genome_id, kr_record* __restrict__ out,
                          u64 cap, u64* __restrict__ nrec) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncand) return;
    u64 pre = cands[c].prefix;                          // absolute
    if (!slice_key(pre, g)) return;                     // -> relative, or not in this slice
    const u64 top = g.sbits ? ((u64)g.slice << (64 - g.sbits)) : 0;
    u32 f = (u32)(pre >> g.rb);
    u32 f2 = (u32)((pre | ~g.pmask) >> g.rb) + 1;
    u32 l = off[f], r = off[f2];
    const u32 e = r;
    while (l < r) {
        u32 mid = l + ((r - l) >> 1);
        if (K[mid] < pre) l = mid + 1; else r = mid;
    }
    u32 i = l;
    while (i < e && (K[i] & g.pmask) == pre) {
        u64 key = K[i];
        u32 cnt = 1;
        while (i + cnt < e && K[i + cnt] == key) cnt++;
        u64 idx = atomicAdd(nrec, 1ull);
        if (out && idx < cap) {
            kr_record rec;
            rec.key = top | (key >> g.sbits);
            rec.genome = genome_id;
            rec.count = cnt;
            out[idx] = rec;
        }
        i += cnt;
    }
}

// ----------------------------------------------------------------------------
// K8  candidate list (x) candidate list, and/or the diagnostic filter
// ----------------------------------------------------------------------------
__global__ void k_cands_flag(kr_cand* __restrict__ cur, u32 n, const kr_cand* __restrict__ other, u32 m,
                             int have_other, int apply_filter, int D, u32* __restrict__ flags,
                             u32* __restrict__ blockcnt) {
    __shared__ u32 waves[17];
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < n) {
        kr_cand c = cur[i];
        keep = true;
        if (have_other) {
            u32 l = 0, r = m;
            while (l < r) {
                u32 mid = l + ((r - l) >> 1);
                if (other[mid].prefix < c.prefix) l = mid + 1; else r = mid;
            }
            if (l < m && other[l].prefix == c.prefix) {
                c.in_mask |= other[l].in_mask;
                c.out_mask |= other[l].out_mask;
                cur[i] = c;
            } else {
                keep = false;
            }
        }
        if (keep && apply_filter && D > 0) keep = passes_filter(c.in_mask, c.out_mask, D);
        flags[i] = keep ? 1u : 0u;
    }
    u32 tot;
    block_compact(keep, waves, tot);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = tot;
}

__global__ void k_cands_compact(const kr_cand* __restrict__ cur, u32 n, const u32* __restrict__ flags,
                                const u32* __restrict__ blockpos, kr_cand* __restrict__ out) {
    __shared__ u32 waves[17];
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = i < n && flags[i] != 0;
    u32 tot;
    u32 pos = block_compact(keep, waves, tot);
    if (keep) out[blockpos[blockIdx.x] + pos] = cur[i];
}

// ============================================================================
// host side
// ============================================================================
// ----------------------------------------------------------------------------
// wide path (amplicon longer than one key: k > 32 or D > 16).  The 64-bit pipeline above is
// reused three times per run with the key generators of wide_keys():
//   phase 1  keys = `left`  of every valid window (both strands)  -> dictL = lefts  present in all genomes
//   phase 2  keys = `right` of every valid window                 -> dictR = rights present in all genomes
//   phase 3  keys = rank(left) : rank(right)  (exact and order preserving; a window whose left
//            or right is in no dictionary cannot belong to a group of all genomes) -> the
//            (left,right) groups present in all genomes, in the reference's group order
// then the members of those groups are located in the genomes (k_wide_locate): per group and
// diagnostic column the bases seen in the ingroup / the outgroup (the filter, Amplicon.py:495-521)
// and, for the surviving groups, one hit (group, genome, position, strand) per member window.
// ----------------------------------------------------------------------------
typedef struct { u32 cand, genome, pos, strand; } wide_hit;

// idx[b] = lower bound of bucket b (top ib bits) in a sorted key array, idx[2^ib] = n.
// (1) entry i fills the buckets (bucket(i - 1), bucket(i)] -- at most the 32 nearest, the
// array was preset to WIDE_UNSET; (2) a bucket still unset (inside a long empty run, or past
// the last key) does its own binary search.  Streaming passes for evenly spread keys.
#define WIDE_UNSET 0xFFFFFFFFu
__global__ void k_index_sorted(const u64* __restrict__ keys, u32 n, int ib, u32* __restrict__ idx) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 hi = keys[i] >> (64 - ib);
    u64 lo = i == 0 ? 0 : (keys[i - 1] >> (64 - ib)) + 1;
    if (hi >= 32 && lo < hi - 31) lo = hi - 31;
    for (u64 b = lo; b <= hi; b++) idx[b] = (u32)i;
}
__global__ void k_index_fill(const u64* __restrict__ keys, u32 n, int ib, u32* __restrict__ idx) {
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 nbk = 1ull << ib;
    if (b > nbk) return;
    if (b == nbk) { idx[b] = n; return; }
    if (idx[b] != WIDE_UNSET) return;
    const u64 target = b << (64 - ib);
    u32 l = 0, r = n;
    while (l < r) {
        const u32 mid = l + ((r - l) >> 1);
        if (keys[mid] < target) l = mid + 1; else r = mid;
    }
    idx[b] = l;
}

__global__ void k_cand_prefixes(const kr_cand* __restrict__ cands, u32 n, u64* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cands[i].prefix;
}

__device__ __forceinline__ u32 code_at(const u64* __restrict__ codes, u64 p) {
    return (u32)(codes[p >> 5] >> (62 - 2 * (int)(p & 31))) & 3u;
}

// per member window: masks[(cand * 2 + side) * W + col / 16] |= 1 << (4 * (col % 16) + base),
// cnt[cand]++, and the window's group is remembered: ci[2 * pos + strand] (WIDE_NONE = no group)
#define WIDE_NONE 0xFFFFFFFFu
__global__ __launch_bounds__(256) void k_wide_locate(const u64* __restrict__ codes, const u32* __restrict__ bad,
                                                    u64 npos, Geom g, const u64* __restrict__ fin,
                                                    const u32* __restrict__ fidx, int fib, int D, int W, u32 side,
                                                    u64* __restrict__ masks, u32* __restrict__ cnt,
                                                    uint2* cibuf, u32 cistride) {
    u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; pos < npos; pos += stride) {
        uint2 found = make_uint2(WIDE_NONE, WIDE_NONE);
        if (bad[pos >> 5] != 0xFFFFFFFFu) {
            u64 key[2];
            const u32 m = wide_keys(codes, bad, pos, g, key[0], key[1]);
            for (u32 strand = 0; strand < 2; strand++) {
                if (!((m >> strand) & 1)) continue;
                u32 ci;
                if (!dict_rank(fin, fidx, fib, key[strand], ci)) continue;
                if (strand) found.y = ci; else found.x = ci;
                atomicAdd(&cnt[ci], 1u);
                u64* mrow = masks + ((u64)ci * 2 + side) * W;
                u64 acc = 0;
                for (int col = 0; col < D; col++) {
                    const u32 base = strand ? 3u - code_at(codes, pos + g.wk - 1 - g.wL - col)
                                            : code_at(codes, pos + g.wL + col);
                    acc |= 1ull << (4 * (col & 15) + base);
                    if ((col & 15) == 15 || col == D - 1) {
                        atomicOr((unsigned long long*)&mrow[col >> 4], (unsigned long long)acc);
                        acc = 0;
                    }
                }
            }
        }
        cibuf[pos * cistride] = found;     // (may alias this window's entry of g.wcache, read above)
    }
}

// hits[hitoff[cand] + cursor[cand]++] for the member windows of the groups the filter kept
__global__ __launch_bounds__(256) void k_wide_emit(const uint2* __restrict__ cibuf, u32 cistride, u64 npos, u32 gidx,
                                                  const u32* __restrict__ hitoff, u32* __restrict__ cursor,
                                                  wide_hit* __restrict__ hits) {
    u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; pos < npos; pos += stride) {
        const uint2 f = cibuf[pos * cistride];
#pragma unroll
        for (u32 strand = 0; strand < 2; strand++) {
            const u32 ci = strand ? f.y : f.x;
            if (ci == WIDE_NONE) continue;
            const u32 o = hitoff[ci];
            if (hitoff[ci + 1] == o) continue;
            hits[o + atomicAdd(&cursor[ci], 1u)] = wide_hit{ci, gidx, (u32)pos, strand};
        }
    }
}

// exclusive scan of a long u32 array in tiles of 2048: tile sums -> k_scan of the sums -> apply
__global__ __launch_bounds__(256) void k_tile_sums(const u32* __restrict__ in, u32 n, u32* __restrict__ tsum) {
    __shared__ u32 waves[17];
    const u32 base = blockIdx.x * 2048u + threadIdx.x * 8u;
    u32 sum = 0;
#pragma unroll
    for (u32 q = 0; q < 8; q++) sum += base + q < n ? in[base + q] : 0u;
    u32 total;
    (void)block_excl_scan(sum, waves, total);
    if (threadIdx.x == 0) tsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_tile_apply(const u32* __restrict__ in, u32 n, const u32* __restrict__ tpos,
                                                   u32 ntiles, u32* __restrict__ out) {
    __shared__ u32 waves[17];
    const u32 base = blockIdx.x * 2048u + threadIdx.x * 8u;
    u32 v[8], sum = 0;
#pragma unroll
    for (u32 q = 0; q < 8; q++) { v[q] = base + q < n ? in[base + q] : 0u; sum += v[q]; }
    u32 total;
    u32 ex = block_excl_scan(sum, waves, total) + tpos[blockIdx.x];
#pragma unroll
    for (u32 q = 0; q < 8; q++) {
        if (base + q < n) out[base + q] = ex;
        ex += v[q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tpos[ntiles];
}

// keep a candidate when some diagnostic column separates the groups (or no filter is asked)
__global__ void k_wide_filter(const u64* __restrict__ masks, u32* __restrict__ cnt, u32 n, int D, int W, int apply_filter) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!apply_filter) return;
    bool keep = false;
    for (int col = 0; col < D && !keep; col++) {
        const u64 x = masks[((u64)i * 2) * W + (col >> 4)] & masks[((u64)i * 2 + 1) * W + (col >> 4)];
        keep = ((x >> (4 * (col & 15))) & 15ull) == 0;
    }
    if (!keep) cnt[i] = 0;
}

// plain streaming copy (the measured-peak companion of the 8 TB/s spec figure in bench.py)
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, u64 n16) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) dst[i] = src[i];
}

static thread_local std::string g_last_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// one key-space slice of a genome (the whole genome when the context has a single slice)
struct Slice {
    DevBuf keys, off, chunkstart, chunkdesc, ovf;   // ovf: u32 count @0, uint2 segments @16
    u64 nmax = 0;            // exact key count of the slice (known after upload)
    u32 nchunks = 0;
    int64_t count = -1;      // key count confirmed by the sort
};

struct Genome {
    int id = -1;
    size_t n_bases = 0;
    u64 nwords = 0;          // ceil(n/32)
    u64 nmax = 0;            // sum of the slice counts
    DevBuf bases, wkeys;     // wkeys: wide path, this genome's composite keys (then its windows' groups)
    std::vector<Slice> sl;
    bool uploaded = false, sorted = false, finalized = false;
    int64_t count = -1;
};

// one sort "lane" = a stream + the scratch one genome sort needs; consecutive genome sorts
// go to different lanes so that their (latency / barrier bound) kernels overlap on the GPU
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false;
    DevBuf codes, bad, partial8, base1, tmpkeys, tp, tiledesc, tilehist, wkeys;
    DevBuf pass0, base64;    // sliced genomes: all keys partitioned by their top byte, the 64-bit bucket bases
};
#define MAX_LANES 8

struct kr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;     // main stream = lanes[0].stream
    Lane lanes[MAX_LANES];
    int nlanes = 1, next_lane = 0;
    int sb = 0, nslices = 1;          // key-space slices: 4^sb
    size_t budget = 0, used = 0;
    bool have_params = false;
    Geom g{};
    size_t max_bases = 0;
    std::map<int, Genome> genomes;
    int ls_grid = 768;     // resident k_localsort workgroups (set from the occupancy query)
    // candidates / records
    DevBuf candA, candB, chunkcnt, chunkpos, flags, blockcnt, blockpos, other, records, nrec, fbdesc, fbsegs;
    int64_t ncand = -1;
    int64_t nrecords = 0;
    std::string err;
    // timers
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool stage_on = false;
    unsigned stage_mask = ~0u;      // stages whose launches are bracketed by events while stage_on
    struct Pair { hipEvent_t a, b; int stage; };
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double stage_ms[KR_ST_COUNT] = {0};
    int64_t stage_n[KR_ST_COUNT] = {0};
    int64_t fallback_launches = 0, overflow_segments = 0;
    int dbg = 0;   // KR_DBG env (test switch, results unchanged): 32 = generic intersect sub-tile path
    int isect_fmt = 0;   // KR_ISECT_FMT env (A/B switch): 1 = narrow mask format also for D <= 4
    // wide windows (kr_set_params_wide / kr_wide_run)
    struct Wide {
        bool on = false;
        int L = 0, D = 0, R = 0, k = 0, omit = 0, W = 0;
        DevBuf dict[2], idx[2], fin, fidx, masks, cnt, hitoff, cursor, hits, cibuf, tsum, tpos;
        bool genome_cache = false;
        u32 ndict[2] = {0, 0}, nfin = 0;
        int ib[2] = {1, 1}, fib = 1;
        int64_t nhits = -1;
    } wide;
};

static int fail(kr_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail((c), KR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

static int ensure(kr_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes < 16) bytes = 16;
    if (b.bytes >= bytes) return KR_OK;
    if (b.p) {
        (void)hipDeviceSynchronize();
        (void)hipFree(b.p);
        c->used -= b.bytes;
        b.p = nullptr;
        b.bytes = 0;
    }
    if (c->budget && c->used + bytes > c->budget)
        return fail(c, KR_ERR_CAPACITY, "hbm budget exceeded: need %zu more bytes, %zu of %zu in use", bytes,
                    c->used, c->budget);
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    c->used += bytes;
    return KR_OK;
}

// grow a buffer that holds live data: contents are preserved
static int ensure_keep(kr_ctx* c, DevBuf& b, size_t bytes, size_t live_bytes) {
    if (b.bytes >= bytes) return KR_OK;
    size_t want = std::max(bytes, b.bytes * 2);
    DevBuf nb;
    int rc = ensure(c, nb, want);
    if (rc) return rc;
    if (b.p && live_bytes) {
        hipError_t e = hipMemcpy(nb.p, b.p, live_bytes, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) return fail(c, KR_ERR_HIP, "hipMemcpy D2D failed: %s", hipGetErrorString(e));
    }
    if (b.p) {
        (void)hipFree(b.p);
        c->used -= b.bytes;
    }
    b = nb;
    return KR_OK;
}

static void release(kr_ctx* c, DevBuf& b) {
    if (b.p) {
        (void)hipFree(b.p);
        c->used -= b.bytes;
    }
    b.p = nullptr;
    b.bytes = 0;
}

static void release_genome(kr_ctx* c, Genome& G) {
    release(c, G.bases);
    release(c, G.wkeys);
    for (Slice& S : G.sl) {
        release(c, S.keys);
        release(c, S.off);
        release(c, S.chunkstart);
        release(c, S.chunkdesc);
        release(c, S.ovf);
    }
    G.sl.clear();
}

struct StageScope {
    kr_ctx* c;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t stream;
    StageScope(kr_ctx* c_, int st, hipStream_t s_ = nullptr) : c(c_), stage(st), stream(s_ ? s_ : c_->stream) {
        c->stage_n[st]++;
        if (!c->stage_on || !((c->stage_mask >> st) & 1u)) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, stream);
    }
    ~StageScope() {
        if (!a) return;
        (void)hipEventRecord(b, stream);
        c->pending.push_back({a, b, stage});
    }
};

static void resolve_stages(kr_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0;
        (void)hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->stage_ms[p.stage] += ms;
        c->pool.push_back(p.a);
        c->pool.push_back(p.b);
    }
    c->pending.clear();
}

// make the main stream wait for every lane that has sorts in flight
static int join_lanes(kr_ctx* c) {
    for (int i = 1; i < c->nlanes; i++) {
        Lane& ln = c->lanes[i];
        if (!ln.pending) continue;
        HIPCHK(c, hipEventRecord(ln.done, ln.stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, ln.done, 0));
        ln.pending = false;
    }
    return KR_OK;
}

extern "C" {

const char* kr_last_error(kr_ctx* c) { return c ? c->err.c_str() : g_last_error.c_str(); }

kr_ctx* kr_create(int device, size_t hbm_budget_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_last_error = "no HIP device visible";
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        g_last_error = "device index out of range";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_last_error = "hipSetDevice failed";
        return nullptr;
    }
    kr_ctx* c = new kr_ctx();
    c->device = device;
    c->budget = hbm_budget_bytes;
    {
        const char* e = getenv("KR_DBG");
        c->dbg = e ? atoi(e) : 0;
        const char* e2 = getenv("KR_ISECT_FMT");
        c->isect_fmt = e2 ? atoi(e2) : 0;
        int ncu = 256, per = 3;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_localsort, LS_THREADS, 0) != hipSuccess || per < 1)
            per = 2;
        c->ls_grid = ncu * per;
    }
    {
        const char* e = getenv("KR_LANES");
        int nl = e ? atoi(e) : 1;   // > 1 overlaps consecutive genome sorts (+0-7 %, box dependent)
        c->nlanes = nl < 1 ? 1 : (nl > MAX_LANES ? MAX_LANES : nl);
    }
    bool ok = hipEventCreate(&c->t0) == hipSuccess && hipEventCreate(&c->t1) == hipSuccess;
    for (int i = 0; ok && i < c->nlanes; i++)
        ok = hipStreamCreate(&c->lanes[i].stream) == hipSuccess &&
             hipEventCreateWithFlags(&c->lanes[i].done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g_last_error = "stream / event creation failed";
        delete c;
        return nullptr;
    }
    c->stream = c->lanes[0].stream;
    return c;
}

void kr_destroy(kr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    resolve_stages(c);
    for (int i = 0; i < c->nlanes; i++) {
        Lane& ln = c->lanes[i];
        DevBuf* lb[] = {&ln.codes, &ln.bad, &ln.partial8, &ln.base1, &ln.tmpkeys, &ln.tp, &ln.tiledesc, &ln.tilehist, &ln.wkeys,
                        &ln.pass0, &ln.base64};
        for (DevBuf* b : lb) release(c, *b);
    }
    for (auto& kv : c->genomes) release_genome(c, kv.second);
    DevBuf* all[] = {&c->candA, &c->candB, &c->chunkcnt,
                     &c->chunkpos, &c->flags, &c->blockcnt, &c->blockpos, &c->other, &c->records, &c->nrec, &c->fbdesc, &c->fbsegs};
    for (DevBuf* b : all) release(c, *b);
    {
        auto& w = c->wide;
        DevBuf* wb[] = {&w.dict[0], &w.dict[1], &w.idx[0], &w.idx[1], &w.fin, &w.fidx, &w.masks, &w.cnt, &w.hitoff, &w.cursor, &w.hits, &w.cibuf, &w.tsum, &w.tpos};
        for (DevBuf* b : wb) release(c, *b);
    }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->t0);
    (void)hipEventDestroy(c->t1);
    for (int i = 0; i < c->nlanes; i++) {
        if (c->lanes[i].done) (void)hipEventDestroy(c->lanes[i].done);
        if (c->lanes[i].stream) (void)hipStreamDestroy(c->lanes[i].stream);
    }
    delete c;
}

static u64 topbits(int nbits) { return nbits <= 0 ? 0ull : (nbits >= 64 ? ~0ull : (~0ull << (64 - nbits))); }

// geometry of slice `slice` (relative part) on top of the context's absolute geometry
static Geom slice_geom(const kr_ctx* c, u32 slice) {
    Geom g = c->g;
    g.slice = slice;
    return g;
}

// absolute + relative geometry of (L, D, R) windows sorted in 4^sb slices with fan-out 2^b
static Geom make_geom(int L, int D, int R, int omit, int sb, int b) {
    Geom g{};
    const int k = L + D + R;
    g.k = k; g.L = L; g.D = D; g.R = R;
    g.sR = 2 * D;
    g.sD = 2 * R;
    g.topmask = topbits(2 * k);
    g.mL = topbits(2 * L);
    g.mR = topbits(2 * (L + R)) & ~g.mL;
    g.mD = topbits(2 * k) & ~topbits(2 * (L + R));
    g.omit = omit;
    g.sbits = 2 * sb;
    g.slice = 0;
    g.LRrel = L - sb + R;
    g.pmask = topbits(2 * g.LRrel);
    g.b = b;
    g.rb = 64 - b;
    return g;
}

// key-space slices: one sort unit holds at most ~4.2e8 keys (fine buckets of <= 1600 keys at the
// largest fan-out b = 18); larger genomes are sorted in 4^sb slices by the first sb bases of `left`.
// fan-out: average fine bucket of ~1600 keys or fewer (limit LS_CAP - LS_T = 2048), 8 <= b <= 18
static int plan_sort(kr_ctx* c, size_t max_bases, int Lmin, int& sb, int& b) {
    const u64 nmax = 2 * (u64)max_bases;
    sb = 0;
    while (sb < 4 && (nmax >> (2 * sb)) > (BUCKET_AVG << 18)) sb++;
    // once slices are needed, finer ones pay: a slice of <= 1e8 keys sorts with fan-out 2^16, whose
    // pass-2 runs are 256 bytes instead of 64 (C5: 33.0 -> 36.4 G k-mers/s with 64 instead of 16 slices)
    if (sb > 0)
        while (sb < 4 && sb < Lmin && (nmax >> (2 * sb)) > (BUCKET_AVG << 16)) sb++;
    if (const char* e = getenv("KR_SLICE_BASES")) sb = std::max(0, std::min(4, atoi(e)));
    if (sb > Lmin) {
        if (getenv("KR_SLICE_BASES")) sb = Lmin;
        else return fail(c, KR_ERR_PARAM, "a genome of %zu bases needs %d slice bases but conserved-left is %d", max_bases, sb, Lmin);
    }
    if ((nmax >> (2 * sb)) > (BUCKET_AVG << 18) * 4)
        return fail(c, KR_ERR_PARAM, "genome of %zu bases is too large for %d slice bases", max_bases, sb);
    const u64 per_slice = nmax >> (2 * sb);
    b = 8;
    while (b < 18 && (per_slice >> b) > BUCKET_AVG) b++;
    return KR_OK;
}

int kr_set_params(kr_ctx* c, int L, int D, int R, int softmask_mode, size_t max_bases) {
    if (!c) return KR_ERR_PARAM;
    const int k = L + D + R;
    if (L < 0 || D < 0 || R < 0 || k < 1 || k > 32)
        return fail(c, KR_ERR_PARAM, "need 1 <= L+D+R <= 32 on the packed path (got %d/%d/%d)", L, D, R);
    if (D > 16) return fail(c, KR_ERR_PARAM, "diagnostic length %d > 16 unsupported by the mask format", D);
    if (softmask_mode != KR_SOFT_MAP && softmask_mode != KR_SOFT_OMIT)
        return fail(c, KR_ERR_PARAM, "unknown softmask mode %d", softmask_mode);
    if (!c->genomes.empty()) return fail(c, KR_ERR_STATE, "kr_set_params after genomes were uploaded");
    if (max_bases >= (1ull << 32) - 64) return fail(c, KR_ERR_PARAM, "genomes of >= 2^32 bases are not supported");
    int sb, b, rc;
    if ((rc = plan_sort(c, max_bases, L, sb, b))) return rc;
    c->sb = sb;
    c->nslices = 1 << (2 * sb);
    c->g = make_geom(L, D, R, softmask_mode == KR_SOFT_OMIT, sb, b);
    c->max_bases = max_bases;
    c->have_params = true;
    c->wide.on = false;
    return KR_OK;
}

int kr_set_strands(kr_ctx* c, int mode) {
    if (!c || !c->have_params) return fail(c, KR_ERR_STATE, "kr_set_params first");
    if (mode < KR_STRANDS_BOTH || mode > KR_STRANDS_CANONICAL) return fail(c, KR_ERR_PARAM, "unknown strand mode %d", mode);
    if (c->wide.on) return fail(c, KR_ERR_PARAM, "the wide path emits both strands only");
    if (!c->genomes.empty()) return fail(c, KR_ERR_STATE, "kr_set_strands after genomes were uploaded");
    c->g.strands = mode;
    return KR_OK;
}

int kr_set_params_wide(kr_ctx* c, int L, int D, int R, int softmask_mode, size_t max_bases) {
    if (!c) return KR_ERR_PARAM;
    const int k = L + D + R;
    if (L < 1 || L > 32 || R < 1 || R > 32 || D < 0 || k > KR_WIDE_MAX_K)
        return fail(c, KR_ERR_PARAM, "wide path needs 1 <= L, R <= 32 and L+D+R <= %d (got %d/%d/%d)", KR_WIDE_MAX_K, L, D, R);
    if (softmask_mode != KR_SOFT_MAP && softmask_mode != KR_SOFT_OMIT)
        return fail(c, KR_ERR_PARAM, "unknown softmask mode %d", softmask_mode);
    if (!c->genomes.empty()) return fail(c, KR_ERR_STATE, "kr_set_params_wide after genomes were uploaded");
    if (max_bases >= (1ull << 32) - 256) return fail(c, KR_ERR_PARAM, "genomes of >= 2^32 bases are not supported");
    int sb, b, rc;
    if ((rc = plan_sort(c, max_bases, std::min(L, R), sb, b))) return rc;
    c->sb = sb;
    c->nslices = 1 << (2 * sb);
    auto& w = c->wide;
    w.on = true;
    w.L = L; w.D = D; w.R = R; w.k = k;
    w.omit = softmask_mode == KR_SOFT_OMIT;
    w.W = (D + 15) / 16;
    w.nhits = -1;
    c->g = make_geom(L, 0, 0, w.omit, sb, b);    // placeholder until kr_wide_run picks a phase
    c->max_bases = max_bases;
    c->have_params = true;
    return KR_OK;
}

static int alloc_slice(kr_ctx* c, Slice& S, u64 count) {
    const u32 nb = 1u << c->g.b;
    S.nmax = count;
    S.nchunks = (u32)(count / LS_T) + 1;
    S.count = -1;
    int rc;
    if ((rc = ensure(c, S.keys, (count + 2) * 8))) return rc;
    if ((rc = ensure(c, S.off, ((size_t)nb + 2) * 4))) return rc;
    if ((rc = ensure(c, S.chunkstart, ((size_t)S.nchunks + 2) * 4))) return rc;
    if ((rc = ensure(c, S.chunkdesc, ((size_t)S.nchunks + 2) * 16))) return rc;
    if ((rc = ensure(c, S.ovf, 16 + (size_t)OVF_MAX * 16))) return rc;
    return KR_OK;
}

// per-lane scratch for sorting slices of up to `maxcount` keys from genomes of up to max_bases
static int ensure_lanes(kr_ctx* c, u64 maxcount) {
    const u32 nb = 1u << c->g.b;
    const u64 mw = (c->max_bases + 31) / 32 + PAD_WORDS + 2;
    const u64 ntmax = maxcount / P2_TILE + 260;
    int rc;
    for (int i = 0; i < c->nlanes; i++) {
        Lane& ln = c->lanes[i];
        if ((rc = ensure(c, ln.codes, mw * 8))) return rc;
        if ((rc = ensure(c, ln.bad, mw * 4))) return rc;
        if ((rc = ensure(c, ln.partial8, (size_t)NWG * 256 * 4))) return rc;
        if ((rc = ensure(c, ln.base1, (260 + 256) * 4))) return rc;     // bases[257] | column totals[256]
        if ((rc = ensure(c, ln.tmpkeys, (maxcount + 2) * 8))) return rc;
        if ((rc = ensure(c, ln.tp, 260 * 4))) return rc;
        if ((rc = ensure(c, ln.base64, 260 * 8))) return rc;
        if ((rc = ensure(c, ln.tiledesc, ntmax * 8))) return rc;
        if ((rc = ensure(c, ln.tilehist, ntmax * (nb >> 8) * 4))) return rc;
    }
    return KR_OK;
}

static void launch_pack(kr_ctx* c, Genome& G, Lane& ln, hipStream_t st) {
    const u64 nwp = G.nwords + PAD_WORDS;   // pad words (all bad): a window may read past the last base
    u32 grid = (u32)std::min<u64>((nwp + 255) / 256, 4096);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, st, (const uint8_t*)G.bases.p, (u64)G.n_bases,
                       (u64*)ln.codes.p, (u32*)ln.bad.p, nwp, c->g.omit);
}

static void launch_scatter1(const Geom& g, hipStream_t st, const u64* codes, const u32* bad, u64 nwords,
                            const u32* base1, const u64* base64, const u32* rowoff, u64* dst) {
    if (g.wmode)
        hipLaunchKernelGGL(k_scatter1<1>, dim3(NWG), dim3(P1_T), 0, st, codes, bad, nwords, base1, base64, rowoff, dst, g);
    else if (g.strands)
        hipLaunchKernelGGL(k_scatter1<2>, dim3(NWG), dim3(P1_T), 0, st, codes, bad, nwords, base1, base64, rowoff, dst, g);
    else
        hipLaunchKernelGGL(k_scatter1<0>, dim3(NWG), dim3(P1_T), 0, st, codes, bad, nwords, base1, base64, rowoff, dst, g);
}

static void launch_hist8(kr_ctx* c, Genome& G, Lane& ln, hipStream_t st, const Geom& g) {
    if (g.wmode)
        hipLaunchKernelGGL(k_hist8<1>, dim3(NWG), dim3(P1_T), 0, st, (const u64*)ln.codes.p, (const u32*)ln.bad.p,
                           G.nwords, (u32*)ln.partial8.p, g);
    else if (g.strands)
        hipLaunchKernelGGL(k_hist8<2>, dim3(NWG), dim3(P1_T), 0, st, (const u64*)ln.codes.p, (const u32*)ln.bad.p,
                           G.nwords, (u32*)ln.partial8.p, g);
    else
        hipLaunchKernelGGL(k_hist8<0>, dim3(NWG), dim3(P1_T), 0, st, (const u64*)ln.codes.p, (const u32*)ln.bad.p,
                           G.nwords, (u32*)ln.partial8.p, g);
}

// exact key count of every slice under the current geometry (sizes the slice arrays):
// pack + top-byte histogram on the main stream
static int count_slices(kr_ctx* c, Genome& G) {
    int rc;
    if ((rc = ensure_lanes(c, 16))) return rc;
    hipStream_t st = c->stream;
    Lane& ln = c->lanes[0];
    launch_pack(c, G, ln, st);
    if ((int)G.sl.size() != c->nslices) {
        for (Slice& S : G.sl) {
            release(c, S.keys); release(c, S.off); release(c, S.chunkstart); release(c, S.chunkdesc); release(c, S.ovf);
        }
        G.sl.assign(c->nslices, Slice());
    }
    std::vector<u32> tot(256);
    u64 maxcount = 0;
    G.nmax = 0;
    G.sorted = G.finalized = false;
    G.count = -1;
    // ONE histogram of the top byte of the absolute keys: a slice is 2^(8 - sbits) of its buckets
    Geom g0 = c->g;
    g0.sbits = 0;
    g0.slice = 0;
    if (g0.wmode == 2 && g0.wcache) g0.wcmode = 1;
    {
        StageScope sc(c, KR_ST_HIST8, st);
        launch_hist8(c, G, ln, st, g0);
    }
    hipLaunchKernelGGL(k_reduce8a, dim3(256), dim3(256), 0, st, (u32*)ln.partial8.p, (u32*)ln.base1.p + 260);
    HIPCHK(c, hipMemcpyAsync(tot.data(), (u32*)ln.base1.p + 260, 256 * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int per = 256 >> c->g.sbits;          // buckets per slice
    for (int s = 0; s < c->nslices; s++) {
        u64 cnt = 0;
        for (int d = s * per; d < (s + 1) * per; d++) cnt += tot[d];
        if (cnt >= (1ull << 31))
            return fail(c, KR_ERR_CAPACITY, "slice %d of genome %d holds %llu keys (>= 2^31): raise KR_SLICE_BASES", s,
                        G.id, (unsigned long long)cnt);
        if ((rc = alloc_slice(c, G.sl[s], cnt))) return rc;
        G.nmax += cnt;
        maxcount = std::max(maxcount, cnt);
    }
    if (c->nslices > 1)
        for (int i = 0; i < c->nlanes; i++)
            if ((rc = ensure(c, c->lanes[i].pass0, (G.nmax + 2) * 8))) return rc;
    HIPCHK(c, hipGetLastError());
    return ensure_lanes(c, maxcount);
}

int kr_genome_upload(kr_ctx* c, int id, const uint8_t* bases, size_t n) {
    if (!c || !c->have_params) return fail(c, KR_ERR_STATE, "kr_set_params first");
    if (n > c->max_bases) return fail(c, KR_ERR_PARAM, "genome of %zu bases exceeds max_bases %zu", n, c->max_bases);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    Genome& G = c->genomes[id];
    G.id = id;
    G.n_bases = n;
    G.nwords = (n + 31) / 32;
    G.sorted = G.finalized = false;
    G.count = -1;
    int rc;
    if ((rc = ensure(c, G.bases, n + 64))) return rc;
    if (n) HIPCHK(c, hipMemcpyAsync(G.bases.p, bases, n, hipMemcpyHostToDevice, c->stream));
    if (c->wide.on) {            // the slices are counted per phase by kr_wide_run
        HIPCHK(c, hipStreamSynchronize(c->stream));
        G.uploaded = true;
        return KR_OK;
    }
    if ((rc = count_slices(c, G))) return rc;
    G.uploaded = true;
    return KR_OK;
}

static int genome_sort(kr_ctx* c, int id, bool reuse_count);
int kr_genome_sort(kr_ctx* c, int id) { return genome_sort(c, id, false); }

// reuse_count: count_slices(G) has just run on this lane (single lane): the codes, the bad
// bits and the workgroup histograms of the absolute top byte (already prefix-summed by
// k_reduce8a) are still in the lane's scratch
static int genome_sort(kr_ctx* c, int id, bool reuse_count) {
    if (!c) return KR_ERR_PARAM;
    auto it = c->genomes.find(id);
    if (it == c->genomes.end() || !it->second.uploaded) return fail(c, KR_ERR_STATE, "genome %d not uploaded", id);
    if (it->second.sl.empty()) return fail(c, KR_ERR_STATE, "genome %d has no slice plan (wide windows: use kr_wide_run)", id);
    HIPCHK(c, hipSetDevice(c->device));
    Genome& G = it->second;
    const u32 nb = 1u << c->g.b;
    Lane& ln = c->lanes[c->next_lane];
    c->next_lane = (c->next_lane + 1) % c->nlanes;
    ln.pending = true;
    hipStream_t st = ln.stream;
    u64* codes = (u64*)ln.codes.p;
    u32* bad = (u32*)ln.bad.p;
    G.sorted = G.finalized = false;
    G.count = -1;
    const bool cached_keys = reuse_count && c->nlanes == 1 && c->g.wmode == 2 && c->g.wcache;
    reuse_count = reuse_count && c->nlanes == 1;
    const bool sliced = c->nslices > 1;
    if (!reuse_count) {
        StageScope sc(c, KR_ST_PACK, st);
        launch_pack(c, G, ln, st);
    }
    if (sliced) {
        // pass 0: all keys of the genome, generated once, partitioned by their top byte; slice s
        // is the contiguous run of its 2^(8 - sbits) buckets
        if (ln.pass0.bytes < (G.nmax + 2) * 8) return fail(c, KR_ERR_STATE, "genome %d: slice plan is stale", id);
        Geom g0 = c->g;
        g0.sbits = 0;
        g0.slice = 0;
        if (!reuse_count) {
            if (g0.wmode == 2 && g0.wcache && c->nlanes == 1) g0.wcmode = 1;
            StageScope sc(c, KR_ST_HIST8, st);
            launch_hist8(c, G, ln, st, g0);
        }
        {
            StageScope sc(c, KR_ST_REDUCE8, st);
            if (!reuse_count)
                hipLaunchKernelGGL(k_reduce8a, dim3(256), dim3(256), 0, st, (u32*)ln.partial8.p, (u32*)ln.base1.p + 260);
            hipLaunchKernelGGL(k_bases64, dim3(1), dim3(64), 0, st, (const u32*)ln.base1.p + 260, (u64*)ln.base64.p);
        }
        g0.wcmode = (g0.wmode == 2 && g0.wcache && c->nlanes == 1) ? 2 : 0;
        (void)cached_keys;
        StageScope sc(c, KR_ST_SCATTER1, st);
        launch_scatter1(g0, st, (const u64*)codes, (const u32*)bad, G.nwords, (const u32*)nullptr,
                        (const u64*)ln.base64.p, (const u32*)ln.partial8.p, (u64*)ln.pass0.p);
    }
    u64 region = 0;       // start of the current slice inside the pass-0 array
    for (int s = 0; s < c->nslices; s++) {
        Slice& S = G.sl[s];
        Geom g = slice_geom(c, (u32)s);
        if (cached_keys) g.wcmode = 2;
        S.count = -1;
        const u64* src = (const u64*)ln.pass0.p + region;
        region += S.nmax;
        if (sliced) {
            StageScope sc(c, KR_ST_HIST8, st);
            hipLaunchKernelGGL(k_hist8k, dim3(NWG), dim3(P1_T), 0, st, src, (u64)S.nmax, g.sbits, (u32*)ln.partial8.p);
        } else if (!reuse_count) {
            StageScope sc(c, KR_ST_HIST8, st);
            launch_hist8(c, G, ln, st, g);
        }
        {
            StageScope sc(c, KR_ST_REDUCE8, st);
            if (g.b > 8)
                HIPCHK(c, hipMemsetAsync(ln.tiledesc.p, 0, ((size_t)(S.nmax / P2_TILE) + 257) * 8, st));
            if (sliced || !reuse_count)
                hipLaunchKernelGGL(k_reduce8a, dim3(256), dim3(256), 0, st, (u32*)ln.partial8.p, (u32*)ln.base1.p + 260);
            hipLaunchKernelGGL(k_reduce8b, dim3(1), dim3(1024), 0, st, (const u32*)ln.base1.p + 260,
                               (u32*)ln.base1.p, (u32*)ln.tp.p, (uint2*)ln.tiledesc.p);
        }
        u64* pass1_dst = g.b > 8 ? (u64*)ln.tmpkeys.p : (u64*)S.keys.p;
        {
            StageScope sc(c, KR_ST_SCATTER1, st);
            if (sliced)
                hipLaunchKernelGGL(k_scatter1k, dim3(NWG), dim3(P1_T), 0, st, src, (u64)S.nmax, g.sbits,
                                   (const u32*)ln.base1.p, (const u32*)ln.partial8.p, pass1_dst);
            else
                launch_scatter1(g, st, (const u64*)codes, (const u32*)bad, G.nwords, (const u32*)ln.base1.p,
                                (const u64*)nullptr, (const u32*)ln.partial8.p, pass1_dst);
        }
        if (g.b > 8) {
            const u32 ntmax = (u32)(S.nmax / P2_TILE) + 257;
            {
                StageScope sc(c, KR_ST_HIST2, st);
                hipLaunchKernelGGL(k_hist2, dim3(ntmax), dim3(P2_T), 0, st, (const u64*)ln.tmpkeys.p,
                                   (const uint2*)ln.tiledesc.p, (u32*)ln.tilehist.p, g.b);
            }
            {
                StageScope sc(c, KR_ST_SCAN2, st);
                hipLaunchKernelGGL(k_scan2, dim3(256), dim3(1024), 0, st, (u32*)ln.tilehist.p, (const u32*)ln.base1.p,
                                   (const u32*)ln.tp.p, (u32*)S.off.p, g.b);
            }
            {
                StageScope sc(c, KR_ST_SCATTER2, st);
                hipLaunchKernelGGL(k_scatter2, dim3(ntmax), dim3(P2_T), 0, st, (const u64*)ln.tmpkeys.p,
                                   (u64*)S.keys.p, (const uint2*)ln.tiledesc.p, (const u32*)ln.tilehist.p, g.b);
            }
        } else {
            HIPCHK(c, hipMemcpyAsync(S.off.p, ln.base1.p, 257 * 4, hipMemcpyDeviceToDevice, st));
        }
        {
            StageScope sc(c, KR_ST_CHUNKS, st);
            HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)S.chunkstart.p, (int)nb, S.nchunks + 2, st));
            hipLaunchKernelGGL(k_chunk_bounds, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, (const u32*)S.off.p, nb,
                               (u32*)S.chunkstart.p);
            hipLaunchKernelGGL(k_chunk_desc, dim3((S.nchunks + 255) / 256), dim3(256), 0, st, (const u32*)S.off.p,
                               (const u32*)S.chunkstart.p, S.nchunks, (uint4*)S.chunkdesc.p);
        }
        {
            StageScope sc(c, KR_ST_LOCALSORT, st);
            HIPCHK(c, hipMemsetAsync(S.ovf.p, 0, 16, st));
            const u32 grid = std::min<u32>(S.nchunks, (u32)c->ls_grid);
            hipLaunchKernelGGL(k_localsort, dim3(grid), dim3(LS_THREADS), 0, st, (u64*)S.keys.p,
                               (const u32*)S.off.p, (const uint4*)S.chunkdesc.p, S.nchunks, g.b, (u32*)S.ovf.p,
                               (uint4*)((char*)S.ovf.p + 16), c->dbg);
        }
    }
    G.sorted = true;      // enqueued; oversized buckets (if any) are resolved by finalize()
    return KR_OK;
}

// Resolve what the asynchronous sort left open: the key counts and -- rarely -- the
// oversized buckets the LDS sort could not take (bitonic fallback in global memory).
// One small D2H + sync for ALL listed genomes.
static int finalize(kr_ctx* c, const std::vector<Genome*>& gs) {
    std::vector<Slice*> todo;
    std::vector<Genome*> tg;
    for (Genome* G : gs)
        if (G->sorted && !G->finalized) {
            tg.push_back(G);
            for (Slice& S : G->sl) todo.push_back(&S);
        }
    if (tg.empty()) return KR_OK;
    hipStream_t st = c->stream;
    {
        int rcj = join_lanes(c);
        if (rcj) return rcj;
    }
    const u32 nb = 1u << c->g.b;
    std::vector<u32> novf(todo.size()), total(todo.size());
    for (size_t i = 0; i < todo.size(); i++) {
        HIPCHK(c, hipMemcpyAsync(&novf[i], todo[i]->ovf.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(&total[i], (u32*)todo[i]->off.p + nb, 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    for (size_t i = 0; i < todo.size(); i++) {
        Slice& S = *todo[i];
        S.count = total[i];
        if (novf[i]) {
            StageScope sc(c, KR_ST_FALLBACK);
            const u32 nb = 1u << c->g.b;
            std::vector<uint4> segs;
            uint4* dsegs = (uint4*)((char*)S.ovf.p + 16);
            if (novf[i] > OVF_MAX) {
                segs.push_back(make_uint4(0, total[i], 0, nb));      // everything: bucket range [0, nb)
            } else {
                segs.resize(novf[i]);
                HIPCHK(c, hipMemcpy(segs.data(), dsegs, (size_t)novf[i] * 16, hipMemcpyDeviceToHost));
                for (auto& sg : segs) sg.w = sg.z + 1;               // bucket range [f, f + 1)
            }
            c->overflow_segments += (int64_t)segs.size();
            // (1) sort the 4096-key tiles of every segment with the LDS sorter
            std::vector<uint4> tiles;
            u32 maxlen = 0;
            for (auto& sg : segs) {
                maxlen = std::max(maxlen, sg.y - sg.x);
                for (u32 t = sg.x; t < sg.y; t += LS_CAP) tiles.push_back(make_uint4(t, std::min(t + LS_CAP, sg.y), sg.z, sg.w));
            }
            int rc;
            if ((rc = ensure(c, c->fbdesc, tiles.size() * 16 + 16))) return rc;
            if ((rc = ensure(c, c->fbsegs, (segs.size() + 1) * 16))) return rc;
            HIPCHK(c, hipMemcpy(c->fbdesc.p, tiles.data(), tiles.size() * 16, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemsetAsync(S.ovf.p, 0, 16, st));
            hipLaunchKernelGGL(k_localsort, dim3(std::min<u32>((u32)tiles.size(), (u32)c->ls_grid)), dim3(LS_THREADS), 0,
                               st, (u64*)S.keys.p, (const u32*)S.off.p, (const uint4*)c->fbdesc.p, (u32)tiles.size(),
                               c->g.b, (u32*)S.ovf.p, dsegs, c->dbg);
            c->fallback_launches++;
            // (2) merge rounds: runs of `run` keys -> 2 * run, through the lane scratch and back
            Lane& ln = c->lanes[0];
            if ((rc = ensure(c, ln.tmpkeys, ((size_t)total[i] + 2) * 8))) return rc;
            for (u64 run = LS_CAP; run < maxlen; run *= 2) {
                std::vector<uint4> act;
                u32 ntiles = 0;
                for (auto& sg : segs) {
                    u32 len = sg.y - sg.x;
                    if (len <= run) continue;
                    act.push_back(make_uint4(sg.x, sg.y, ntiles, 0));
                    ntiles += (len + MG_TILE - 1) / MG_TILE;
                }
                if (act.empty()) break;
                HIPCHK(c, hipMemcpy(c->fbsegs.p, act.data(), act.size() * 16, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_seg_merge, dim3(ntiles), dim3(MG_T), 0, st, (const u64*)S.keys.p,
                                   (u64*)ln.tmpkeys.p, (const uint4*)c->fbsegs.p, (u32)act.size(), (u32)run);
                hipLaunchKernelGGL(k_seg_copy, dim3(ntiles), dim3(256), 0, st, (const u64*)ln.tmpkeys.p,
                                   (u64*)S.keys.p, (const uint4*)c->fbsegs.p, (u32)act.size());
                HIPCHK(c, hipStreamSynchronize(st));      // fbsegs is rewritten next round
                c->fallback_launches += 2;
            }
            HIPCHK(c, hipGetLastError());
        }
    }
    for (Genome* G : tg) {
        G->count = 0;
        for (Slice& S : G->sl) G->count += S.count;
        G->finalized = true;
    }
    return KR_OK;
}

int64_t kr_genome_load_sorted(kr_ctx* c, int id, const uint64_t* keys, size_t n) {
    if (!c || !c->have_params) return fail(c, KR_ERR_STATE, "kr_set_params first");
    if (n > 2 * c->max_bases) return fail(c, KR_ERR_PARAM, "%zu keys exceed 2 * max_bases", n);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    Genome& G = c->genomes[id];
    G.id = id;
    G.n_bases = 0;
    G.nwords = 0;
    G.nmax = n;
    for (Slice& S : G.sl) {
        release(c, S.keys); release(c, S.off); release(c, S.chunkstart); release(c, S.chunkdesc); release(c, S.ovf);
    }
    G.sl.assign(c->nslices, Slice());
    const u32 nb = 1u << c->g.b;
    const int sbits = c->g.sbits;
    hipStream_t st = c->stream;
    size_t pos = 0;
    std::vector<u64> rel;
    for (int s = 0; s < c->nslices; s++) {
        // the slice's keys are a contiguous run of the sorted input; store them relative
        size_t end = n;
        if (sbits && s + 1 < c->nslices) {
            const u64 bound = (u64)(s + 1) << (64 - sbits);
            end = std::lower_bound(keys + pos, keys + n, (uint64_t)bound) - keys;
        }
        const size_t cnt = end - pos;
        Slice& S = G.sl[s];
        int rc = alloc_slice(c, S, cnt);
        if (rc) return rc;
        if (cnt) {
            const void* src = keys + pos;
            if (sbits) {
                rel.resize(cnt);
                for (size_t i = 0; i < cnt; i++) rel[i] = (u64)keys[pos + i] << sbits;
                src = rel.data();
            }
            HIPCHK(c, hipMemcpy(S.keys.p, src, cnt * 8, hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(k_offsets_from_sorted, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, (const u64*)S.keys.p,
                           (u32)cnt, nb, c->g.rb, (u32*)S.off.p);
        HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)S.chunkstart.p, (int)nb, S.nchunks + 2, st));
        hipLaunchKernelGGL(k_chunk_bounds, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, (const u32*)S.off.p, nb,
                           (u32*)S.chunkstart.p);
        hipLaunchKernelGGL(k_chunk_desc, dim3((S.nchunks + 255) / 256), dim3(256), 0, st, (const u32*)S.off.p,
                           (const u32*)S.chunkstart.p, S.nchunks, (uint4*)S.chunkdesc.p);
        HIPCHK(c, hipMemsetAsync(S.ovf.p, 0, 16, st));
        S.count = (int64_t)cnt;
        pos = end;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    G.uploaded = false;
    G.sorted = true;
    G.finalized = true;
    G.count = (int64_t)n;
    return (int64_t)n;
}

int64_t kr_genome_count(kr_ctx* c, int id) {
    if (!c) return KR_ERR_PARAM;
    auto it = c->genomes.find(id);
    if (it == c->genomes.end() || !it->second.sorted) return fail(c, KR_ERR_STATE, "genome %d not sorted", id);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = finalize(c, {&it->second});
    if (rc) return rc;
    return it->second.count;
}

int64_t kr_genome_add(kr_ctx* c, int id, const uint8_t* bases, size_t n) {
    int rc = kr_genome_upload(c, id, bases, n);
    if (rc) return rc;
    rc = kr_genome_sort(c, id);
    if (rc) return rc;
    return kr_genome_count(c, id);
}

int64_t kr_genome_fetch_keys(kr_ctx* c, int id, uint64_t* out, size_t cap) {
    int64_t n = kr_genome_count(c, id);
    if (n < 0) return n;
    if ((size_t)n > cap) return fail(c, KR_ERR_CAPACITY, "key buffer too small: %lld > %zu", (long long)n, cap);
    Genome& G = c->genomes[id];
    HIPCHK(c, hipDeviceSynchronize());
    const int sbits = c->g.sbits;
    size_t pos = 0;
    for (int s = 0; s < c->nslices; s++) {
        Slice& S = G.sl[s];
        const size_t cnt = (size_t)S.count;
        if (cnt) HIPCHK(c, hipMemcpy(out + pos, S.keys.p, cnt * 8, hipMemcpyDeviceToHost));
        if (sbits) {      // relative -> absolute
            const u64 top = (u64)s << (64 - sbits);
            for (size_t i = 0; i < cnt; i++) out[pos + i] = top | (out[pos + i] >> sbits);
        }
        pos += cnt;
    }
    return n;
}

int kr_genome_free(kr_ctx* c, int id) {
    if (!c) return KR_ERR_PARAM;
    auto it = c->genomes.find(id);
    if (it == c->genomes.end()) return fail(c, KR_ERR_PARAM, "unknown genome %d", id);
    (void)hipDeviceSynchronize();
    release_genome(c, it->second);
    c->genomes.erase(it);
    return KR_OK;
}

int64_t kr_intersect(kr_ctx* c, const int* ids, int n, const uint8_t* is_in, int apply_filter) {
    if (!c || n < 1) return fail(c, KR_ERR_PARAM, "kr_intersect: need at least one genome");
    if (n > MAXG) return fail(c, KR_ERR_PARAM, "kr_intersect: at most %d genomes per call (cascade with kr_cands_merge)", MAXG);
    HIPCHK(c, hipSetDevice(c->device));
    int anchor = 0;
    u64 best = ~0ull;
    u32 ingroup_bits = 0;
    std::vector<Genome*> gs;
    for (int i = 0; i < n; i++) {
        auto it = c->genomes.find(ids[i]);
        if (it == c->genomes.end() || !it->second.sorted) return fail(c, KR_ERR_STATE, "genome %d not sorted", ids[i]);
        Genome& G = it->second;
        if (is_in[i]) ingroup_bits |= 1u << i;
        if (G.nmax < best) { best = G.nmax; anchor = i; }
        gs.push_back(&G);
    }
    int rc;
    if ((rc = finalize(c, gs))) return rc;
    Genome& A = *gs[anchor];
    u64 amax = 0;
    u32 cmax = 0;
    for (Slice& S : A.sl) { amax = std::max(amax, S.nmax); cmax = std::max(cmax, S.nchunks); }
    if ((rc = ensure(c, c->candA, (amax + 2) * sizeof(kr_cand)))) return rc;      // per-slice sparse runs
    if ((rc = ensure(c, c->candB, 4096 * sizeof(kr_cand)))) return rc;            // dense result: grows as needed
    if ((rc = ensure(c, c->chunkcnt, ((size_t)cmax + 2) * 4))) return rc;
    if ((rc = ensure(c, c->chunkpos, ((size_t)cmax + 2) * 4))) return rc;
    hipStream_t st = c->stream;
    u64 running = 0;
    for (int s = 0; s < c->nslices; s++) {
        const Geom g = slice_geom(c, (u32)s);
        Slice& AS = A.sl[s];
        IsectArgs a{};
        a.n = n;
        a.ingroup_bits = ingroup_bits;
        a.apply_filter = apply_filter ? 1 : 0;
        a.dbg = c->dbg;
        a.anchor = anchor;
        for (int i = 0; i < n; i++) {
            a.keys[i] = (const u64*)gs[i]->sl[s].keys.p;
            a.off[i] = (const u32*)gs[i]->sl[s].off.p;
        }
        a.chunkdesc = (const uint4*)AS.chunkdesc.p;
        a.tmp = (kr_cand*)c->candA.p;
        a.chunkcnt = (u32*)c->chunkcnt.p;
        {
            StageScope sc(c, KR_ST_INTERSECT);
            if (g.D > 8)
                hipLaunchKernelGGL(k_intersect<2>, dim3(AS.nchunks), dim3(IS_THREADS), 0, st, a, g);
            else if (g.D > 4 || c->isect_fmt == 1)
                hipLaunchKernelGGL(k_intersect<1>, dim3(AS.nchunks), dim3(IS_THREADS), 0, st, a, g);
            else
                hipLaunchKernelGGL(k_intersect<0>, dim3(AS.nchunks), dim3(IS_THREADS), 0, st, a, g);
        }
        u32 total = 0;
        {
            StageScope sc(c, KR_ST_COMPACT);
            hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, (const u32*)c->chunkcnt.p, (u32*)c->chunkpos.p,
                               AS.nchunks);
            HIPCHK(c, hipMemcpyAsync(&total, (u32*)c->chunkpos.p + AS.nchunks, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, hipGetLastError());
            size_t want = running + total + 2;
            if (s == 0 && c->nslices > 1)       // size for all slices at once instead of doubling through them
                want = (size_t)((double)total * c->nslices * 1.15) + 4096;
            if ((rc = ensure_keep(c, c->candB, want * sizeof(kr_cand), running * sizeof(kr_cand))))
                return rc;
            if (total)
                hipLaunchKernelGGL(k_gather_cands, dim3(AS.nchunks), dim3(64), 0, st, g.sbits, g.slice,
                                   (const kr_cand*)c->candA.p, (const uint4*)AS.chunkdesc.p,
                                   (const u32*)c->chunkcnt.p, (const u32*)c->chunkpos.p,
                                   (kr_cand*)c->candB.p + running);
        }
        running += total;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    c->ncand = (int64_t)running;
    return c->ncand;
}

int64_t kr_cands_count(kr_ctx* c) { return c ? c->ncand : KR_ERR_PARAM; }

int64_t kr_cands_fetch(kr_ctx* c, kr_cand* out, size_t cap) {
    if (!c || c->ncand < 0) return fail(c, KR_ERR_STATE, "no candidate set");
    if ((size_t)c->ncand > cap) return fail(c, KR_ERR_CAPACITY, "candidate buffer too small");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->ncand) HIPCHK(c, hipMemcpy(out, c->candB.p, (size_t)c->ncand * sizeof(kr_cand), hipMemcpyDeviceToHost));
    return c->ncand;
}

int64_t kr_cands_load(kr_ctx* c, const kr_cand* cands, size_t n) {
    if (!c) return KR_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->candB, (n + 2) * sizeof(kr_cand)))) return rc;
    if ((rc = ensure(c, c->candA, (n + 2) * sizeof(kr_cand)))) return rc;
    if (n) HIPCHK(c, hipMemcpy(c->candB.p, cands, n * sizeof(kr_cand), hipMemcpyHostToDevice));
    c->ncand = (int64_t)n;
    return c->ncand;
}

int64_t kr_cands_merge(kr_ctx* c, const kr_cand* other, size_t m, int have_other, int apply_filter) {
    if (!c || c->ncand < 0) return fail(c, KR_ERR_STATE, "no candidate set");
    if (!c->have_params) return fail(c, KR_ERR_STATE, "kr_set_params first");
    HIPCHK(c, hipSetDevice(c->device));
    const u32 n = (u32)c->ncand;
    if (n == 0) return 0;
    hipStream_t st = c->stream;
    int rc;
    const u32 nblk = (n + 255) / 256;
    if ((rc = ensure(c, c->flags, (size_t)n * 4))) return rc;
    if ((rc = ensure(c, c->blockcnt, ((size_t)nblk + 2) * 4))) return rc;
    if ((rc = ensure(c, c->blockpos, ((size_t)nblk + 2) * 4))) return rc;
    if ((rc = ensure(c, c->candA, ((size_t)n + 2) * sizeof(kr_cand)))) return rc;
    if (have_other) {
        if ((rc = ensure(c, c->other, (m + 2) * sizeof(kr_cand)))) return rc;
        if (m) HIPCHK(c, hipMemcpyAsync(c->other.p, other, m * sizeof(kr_cand), hipMemcpyHostToDevice, st));
    }
    {
        StageScope sc(c, KR_ST_MERGE);
        hipLaunchKernelGGL(k_cands_flag, dim3(nblk), dim3(256), 0, st, (kr_cand*)c->candB.p, n,
                           (const kr_cand*)c->other.p, (u32)m, have_other ? 1 : 0, apply_filter ? 1 : 0, c->g.D,
                           (u32*)c->flags.p, (u32*)c->blockcnt.p);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, (const u32*)c->blockcnt.p, (u32*)c->blockpos.p, nblk);
        hipLaunchKernelGGL(k_cands_compact, dim3(nblk), dim3(256), 0, st, (const kr_cand*)c->candB.p, n,
                           (const u32*)c->flags.p, (const u32*)c->blockpos.p, (kr_cand*)c->candA.p);
    }
    u32 total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, (u32*)c->blockpos.p + nblk, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    std::swap(c->candA, c->candB);
    c->ncand = total;
    return total;
}

int64_t kr_collect(kr_ctx* c, const int* ids, int n) {
    if (!c || c->ncand < 0) return fail(c, KR_ERR_STATE, "no candidate set");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    int rc;
    if ((rc = ensure(c, c->nrec, 16))) return rc;
    const u32 nc = (u32)c->ncand;
    c->nrecords = 0;
    if (nc == 0 || n == 0) return 0;
    std::vector<Genome*> gs;
    for (int i = 0; i < n; i++) {
        auto it = c->genomes.find(ids[i]);
        if (it == c->genomes.end() || !it->second.sorted)
            return fail(c, KR_ERR_STATE, "genome %d not sorted", ids[i]);
        gs.push_back(&it->second);
    }
    if ((rc = finalize(c, gs))) return rc;
    for (int pass = 0; pass < 2; pass++) {
        HIPCHK(c, hipMemsetAsync(c->nrec.p, 0, 16, st));
        StageScope sc(c, KR_ST_COLLECT);
        for (int i = 0; i < n; i++) {
            for (int s = 0; s < c->nslices; s++) {
                Slice& S = gs[i]->sl[s];
                if (S.count == 0) continue;
                hipLaunchKernelGGL(k_collect, dim3((nc + 127) / 128), dim3(128), 0, st, (const kr_cand*)c->candB.p,
                                   nc, (const u64*)S.keys.p, (const u32*)S.off.p, slice_geom(c, (u32)s), (u32)ids[i],
                                   pass ? (kr_record*)c->records.p : (kr_record*)nullptr,
                                   pass ? (u64)(c->records.bytes / sizeof(kr_record)) : 0ull, (u64*)c->nrec.p);
            }
        }
        u64 total = 0;
        HIPCHK(c, hipMemcpyAsync(&total, c->nrec.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, hipGetLastError());
        c->nrecords = (int64_t)total;
        if (pass == 0) {
            if (total == 0) return 0;
            if ((rc = ensure(c, c->records, (total + 2) * sizeof(kr_record)))) return rc;
        }
    }
    return c->nrecords;
}

int64_t kr_fetch(kr_ctx* c, kr_record* out, size_t cap) {
    if (!c) return KR_ERR_PARAM;
    if ((size_t)c->nrecords > cap) return fail(c, KR_ERR_CAPACITY, "record buffer too small");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->nrecords)
        HIPCHK(c, hipMemcpy(out, c->records.p, (size_t)c->nrecords * sizeof(kr_record), hipMemcpyDeviceToHost));
    return c->nrecords;
}

// ----------------------------------------------------------------------------
// wide path driver
// ----------------------------------------------------------------------------
static int ceil_log2(u64 n) { int b = 0; while ((1ull << b) < n) b++; return b; }

// sort every listed genome under the current c->g and intersect them (no filter): the
// candidates' prefixes, dense and sorted, go to `out` with a top-bits index
static int64_t wide_phase(kr_ctx* c, const int* ids, int n, const uint8_t* is_in, DevBuf& out, DevBuf& idx, int& ib,
                          int keybits) {
    int rc;
    for (int i = 0; i < n; i++) {
        Genome& G = c->genomes[ids[i]];
        if (c->g.wmode == 2 && c->wide.genome_cache) {
            if ((rc = ensure(c, G.wkeys, (G.nwords + PAD_WORDS) * 32 * 16))) return rc;
            c->g.wcache = (u64*)G.wkeys.p;
        }
        if ((rc = count_slices(c, G))) return rc;
        if ((rc = genome_sort(c, ids[i], true))) return rc;
    }
    // more than MAXG genomes: cascade (the running candidate list goes through the host)
    int64_t nc = -1;
    std::vector<kr_cand> prev;
    for (int o = 0; o < n; o += MAXG) {
        const int m = std::min(MAXG, n - o);
        nc = kr_intersect(c, ids + o, m, is_in + o, 0);
        if (nc < 0) return nc;
        if (o > 0) {
            nc = kr_cands_merge(c, prev.data(), prev.size(), 1, 0);
            if (nc < 0) return nc;
        }
        if (o + MAXG < n) {
            prev.resize((size_t)nc);
            if (nc) HIPCHK(c, hipMemcpy(prev.data(), c->candB.p, (size_t)nc * sizeof(kr_cand), hipMemcpyDeviceToHost));
        }
    }
    if (nc >= (1ll << 32) - 1) return fail(c, KR_ERR_CAPACITY, "wide path: %lld dictionary entries (>= 2^32)", (long long)nc);
    ib = std::max(1, std::min(std::min(27, keybits), ceil_log2((u64)nc + 1)));     // ~1 entry per index bucket
    if ((rc = ensure(c, out, ((size_t)nc + 2) * 8))) return rc;
    if ((rc = ensure(c, idx, (((size_t)1 << ib) + 2) * 4))) return rc;
    hipStream_t st = c->stream;
    if (nc)
        hipLaunchKernelGGL(k_cand_prefixes, dim3(((u32)nc + 255) / 256), dim3(256), 0, st, (const kr_cand*)c->candB.p,
                           (u32)nc, (u64*)out.p);
    HIPCHK(c, hipMemsetAsync(idx.p, 0xFF, (((size_t)1 << ib) + 1) * 4, st));
    if (nc)
        hipLaunchKernelGGL(k_index_sorted, dim3((u32)(((u64)nc + 255) / 256)), dim3(256), 0, st, (const u64*)out.p,
                           (u32)nc, ib, (u32*)idx.p);
    hipLaunchKernelGGL(k_index_fill, dim3((u32)((((u64)1 << ib) + 1 + 255) / 256)), dim3(256), 0, st,
                       (const u64*)out.p, (u32)nc, ib, (u32*)idx.p);
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    return nc;
}

int64_t kr_wide_run(kr_ctx* c, const int* ids, int n, const uint8_t* is_in, int apply_filter) {
    if (!c || !c->wide.on) return fail(c, KR_ERR_STATE, "kr_set_params_wide first");
    if (n < 1) return fail(c, KR_ERR_PARAM, "kr_wide_run: need at least one genome");
    for (int i = 0; i < n; i++) {
        auto it = c->genomes.find(ids[i]);
        if (it == c->genomes.end() || !it->second.uploaded) return fail(c, KR_ERR_STATE, "genome %d not uploaded", ids[i]);
    }
    HIPCHK(c, hipSetDevice(c->device));
    auto& w = c->wide;
    w.nhits = -1;
    w.nfin = 0;
    const int sb = c->sb, b = c->g.b;
    // phase 1 / 2: the `left` / `right` spectra present in all genomes
    for (int ph = 0; ph < 2; ph++) {
        const int len = ph == 0 ? w.L : w.R;
        Geom g = make_geom(len, 0, 0, w.omit, sb, b);
        g.wmode = 1;
        g.wk = w.k;
        g.wlen = len;
        g.wfo = ph == 0 ? 0 : w.k - w.R;
        g.wro = ph == 0 ? w.k - w.L : 0;
        c->g = g;
        int64_t nd = wide_phase(c, ids, n, is_in, w.dict[ph], w.idx[ph], w.ib[ph], 2 * len);
        if (nd < 0) return nd;
        w.ndict[ph] = (u32)nd;
        if (nd == 0) { w.nhits = 0; return 0; }
    }
    // phase 3: composite keys rank(left) : rank(right)
    {
        int bitsL = std::max(1, ceil_log2((u64)w.ndict[0] + 1)), bitsR = std::max(1, ceil_log2(w.ndict[1]));
        if ((bitsL + bitsR) & 1) bitsR++;
        while (bitsL + bitsR < 2 * sb + 2) bitsR += 2;       // room for the slice digits
        const int half = (bitsL + bitsR) / 2;
        Geom g = make_geom(half, 0, 0, w.omit, sb, b);
        g.wmode = 2;
        g.wk = w.k;
        g.wL = w.L;
        g.wR = w.R;
        g.wshL = 64 - bitsL;
        g.wshR = 64 - bitsL - bitsR;
        g.wibL = w.ib[0];
        g.wibR = w.ib[1];
        g.wdictL = (const u64*)w.dict[0].p;
        g.wdictR = (const u64*)w.dict[1].p;
        g.widxL = (const u32*)w.idx[0].p;
        g.widxR = (const u32*)w.idx[1].p;
        // composite keys are generated once per genome and phase (16 bytes per window start).
        // Kept per genome when that fits comfortably -- the locate pass then reads them too and
        // leaves each window's group in their place -- else in one lane buffer (sort passes only)
        w.genome_cache = false;
        if (c->nlanes == 1) {
            size_t need = 0, fr = 0, tot = 0;
            for (int i = 0; i < n; i++) need += (c->genomes[ids[i]].nwords + PAD_WORDS) * 32 * 16;
            if (getenv("KR_WIDE_CACHE")) w.genome_cache = atoi(getenv("KR_WIDE_CACHE")) != 0;
            else w.genome_cache = hipMemGetInfo(&fr, &tot) == hipSuccess && need <= fr / 6;
            if (!w.genome_cache) {
                int rcw = ensure(c, c->lanes[0].wkeys, ((c->max_bases + 31) / 32 + PAD_WORDS) * 32 * 16);
                if (rcw) return rcw;
                g.wcache = (u64*)c->lanes[0].wkeys.p;
            }
        }
        c->g = g;
        int64_t nf = wide_phase(c, ids, n, is_in, w.fin, w.fidx, w.fib, bitsL + bitsR);
        if (nf < 0) return nf;
        w.nfin = (u32)nf;
        if (nf == 0) { w.nhits = 0; return 0; }
    }
    // members of the groups: masks + counts, filter, hits
    int rc;
    const u32 nf = w.nfin;
    const int W = std::max(1, w.W);
    {
        // the sorted composite keys are not needed any more: give their memory back when the
        // locate buffers would not fit beside them
        size_t need = (size_t)nf * 2 * W * 8 + 3 * ((size_t)nf + 4) * 4, fr = 0, tot = 0;
        if (!w.genome_cache)
            for (int i = 0; i < n; i++) need += c->genomes[ids[i]].n_bases * sizeof(uint2);
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && need > fr / 10 * 8) {
            HIPCHK(c, hipDeviceSynchronize());
            for (int i = 0; i < n; i++) {
                Genome& G = c->genomes[ids[i]];
                for (Slice& S : G.sl) {
                    release(c, S.keys); release(c, S.off); release(c, S.chunkstart); release(c, S.chunkdesc); release(c, S.ovf);
                }
                G.sl.clear();
                G.sorted = G.finalized = false;
            }
            release(c, c->candA);
            release(c, c->candB);
            c->ncand = -1;
        }
    }
    if ((rc = ensure(c, w.masks, (size_t)nf * 2 * W * 8))) return rc;
    if ((rc = ensure(c, w.cnt, ((size_t)nf + 4) * 4))) return rc;
    if ((rc = ensure(c, w.hitoff, ((size_t)nf + 4) * 4))) return rc;
    if ((rc = ensure(c, w.cursor, ((size_t)nf + 4) * 4))) return rc;
    hipStream_t st = c->stream;
    if ((rc = join_lanes(c))) return rc;
    HIPCHK(c, hipMemsetAsync(w.masks.p, 0, (size_t)nf * 2 * W * 8, st));
    HIPCHK(c, hipMemsetAsync(w.cnt.p, 0, ((size_t)nf + 4) * 4, st));
    HIPCHK(c, hipMemsetAsync(w.cursor.p, 0, ((size_t)nf + 4) * 4, st));
    Lane& ln = c->lanes[0];
    // the group of every window of every genome (8 bytes per base): written by the locate pass,
    // streamed by the emit pass once the filter has decided
    std::vector<u64> cioff(n + 1, 0);
    for (int i = 0; i < n; i++) cioff[i + 1] = cioff[i] + c->genomes[ids[i]].n_bases;
    if (!w.genome_cache && (rc = ensure(c, w.cibuf, (cioff[n] + 2) * sizeof(uint2)))) return rc;
    for (int i = 0; i < n; i++) {
        Genome& G = c->genomes[ids[i]];
        if (G.n_bases == 0) continue;
        launch_pack(c, G, ln, st);
        const u32 grid = (u32)std::min<u64>((G.n_bases + 255) / 256, 16384);
        Geom g = c->g;
        g.wcmode = 0;
        uint2* ci = (uint2*)w.cibuf.p + cioff[i];
        if (w.genome_cache) {
            g.wcache = (u64*)G.wkeys.p;
            g.wcmode = 2;
            ci = (uint2*)G.wkeys.p;
        }
        StageScope sc(c, KR_ST_LOCATE, st);
        hipLaunchKernelGGL(k_wide_locate, dim3(grid), dim3(256), 0, st, (const u64*)ln.codes.p, (const u32*)ln.bad.p,
                           (u64)G.n_bases, g, (const u64*)w.fin.p, (const u32*)w.fidx.p, w.fib, w.D, W,
                           is_in[i] ? 0u : 1u, (u64*)w.masks.p, (u32*)w.cnt.p, ci, w.genome_cache ? 2u : 1u);
    }
    hipLaunchKernelGGL(k_wide_filter, dim3((nf + 255) / 256), dim3(256), 0, st, (const u64*)w.masks.p, (u32*)w.cnt.p,
                       nf, w.D, W, apply_filter ? 1 : 0);
    {
        const u32 ntiles = (nf + 2047) / 2048;
        if ((rc = ensure(c, w.tsum, ((size_t)ntiles + 4) * 4))) return rc;
        if ((rc = ensure(c, w.tpos, ((size_t)ntiles + 4) * 4))) return rc;
        hipLaunchKernelGGL(k_tile_sums, dim3(ntiles), dim3(256), 0, st, (const u32*)w.cnt.p, nf, (u32*)w.tsum.p);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, (const u32*)w.tsum.p, (u32*)w.tpos.p, ntiles);
        hipLaunchKernelGGL(k_tile_apply, dim3(ntiles), dim3(256), 0, st, (const u32*)w.cnt.p, nf, (const u32*)w.tpos.p,
                           ntiles, (u32*)w.hitoff.p);
    }
    u32 total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, (u32*)w.hitoff.p + nf, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    w.nhits = total;
    if (total == 0) return 0;
    if ((rc = ensure(c, w.hits, ((size_t)total + 2) * sizeof(wide_hit)))) return rc;
    for (int i = 0; i < n; i++) {
        Genome& G = c->genomes[ids[i]];
        if (G.n_bases == 0) continue;
        const u32 grid = (u32)std::min<u64>((G.n_bases + 255) / 256, 16384);
        const uint2* ci = w.genome_cache ? (const uint2*)G.wkeys.p : (const uint2*)w.cibuf.p + cioff[i];
        hipLaunchKernelGGL(k_wide_emit, dim3(grid), dim3(256), 0, st, ci, w.genome_cache ? 2u : 1u, (u64)G.n_bases,
                           (u32)i, (const u32*)w.hitoff.p, (u32*)w.cursor.p, (wide_hit*)w.hits.p);
    }
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    return w.nhits;
}

int64_t kr_wide_fetch(kr_ctx* c, int what, void* out, size_t cap_bytes) {
    if (!c || !c->wide.on) return fail(c, KR_ERR_STATE, "kr_set_params_wide first");
    HIPCHK(c, hipSetDevice(c->device));
    auto& w = c->wide;
    const void* src = nullptr;
    size_t n = 0, esz = 8;
    switch (what) {
    case KR_WIDE_DICT_LEFT: src = w.dict[0].p; n = w.ndict[0]; break;
    case KR_WIDE_DICT_RIGHT: src = w.dict[1].p; n = w.ndict[1]; break;
    case KR_WIDE_GROUPS: src = w.fin.p; n = w.nfin; break;
    case KR_WIDE_HITS: src = w.hits.p; n = w.nhits > 0 ? (size_t)w.nhits : 0; esz = sizeof(wide_hit); break;
    default: return fail(c, KR_ERR_PARAM, "kr_wide_fetch: unknown selector %d", what);
    }
    if (!out) return (int64_t)n;       // size query
    if (n * esz > cap_bytes) return fail(c, KR_ERR_CAPACITY, "kr_wide_fetch: buffer too small (%zu bytes needed)", n * esz);
    HIPCHK(c, hipDeviceSynchronize());
    if (n) HIPCHK(c, hipMemcpy(out, src, n * esz, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

// timing aid (tools/ls_ablate.py): run k_localsort `reps` times over the (already sorted) slice 0
// of a genome; mode 0 = the kernel as it is, 64 = load + store only, 128 = without the ranking
// step.  A final regular pass restores the order.  Returns the average milliseconds per launch.
double kr_debug_localsort(kr_ctx* c, int id, int reps, int mode) {
    if (!c) return -1.0;
    auto it = c->genomes.find(id);
    if (it == c->genomes.end() || !it->second.sorted || it->second.sl.empty()) return -1.0;
    if (finalize(c, {&it->second})) return -1.0;
    Slice& S = it->second.sl[0];
    hipStream_t st = c->stream;
    const u32 grid = std::min<u32>(S.nchunks, (u32)c->ls_grid);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, st);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k_localsort, dim3(grid), dim3(LS_THREADS), 0, st, (u64*)S.keys.p, (const u32*)S.off.p,
                           (const uint4*)S.chunkdesc.p, S.nchunks, c->g.b, (u32*)S.ovf.p,
                           (uint4*)((char*)S.ovf.p + 16), mode);
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    hipLaunchKernelGGL(k_localsort, dim3(grid), dim3(LS_THREADS), 0, st, (u64*)S.keys.p, (const u32*)S.off.p,
                       (const uint4*)S.chunkdesc.p, S.nchunks, c->g.b, (u32*)S.ovf.p,
                       (uint4*)((char*)S.ovf.p + 16), 0);
    (void)hipStreamSynchronize(st);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return reps > 0 ? (double)ms / reps : 0.0;
}

// measured HBM copy rate on this device: GB/s of read + write traffic of a streaming copy kernel
double kr_debug_copy_gbps(kr_ctx* c, size_t bytes, int reps) {
    if (!c || bytes < 4096 || reps < 1) return -1.0;
    if (hipSetDevice(c->device) != hipSuccess) return -1.0;
    DevBuf a, b;
    if (ensure(c, a, bytes) || ensure(c, b, bytes)) { release(c, a); release(c, b); return -1.0; }
    hipStream_t st = c->stream;
    (void)hipMemsetAsync(a.p, 1, bytes, st);
    (void)hipMemsetAsync(b.p, 2, bytes, st);
    const u64 n16 = bytes / 16;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_copy16, dim3(8192), dim3(256), 0, st, (const uint4*)a.p, (uint4*)b.p, n16);   // warm-up
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k_copy16, dim3(8192), dim3(256), 0, st, (const uint4*)a.p, (uint4*)b.p, n16);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    release(c, a);
    release(c, b);
    return ms > 0 ? 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9 : -1.0;
}

int kr_sync(kr_ctx* c) {
    if (!c) return KR_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    {
        int rcj = join_lanes(c);
        if (rcj) return rcj;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return KR_OK;
}

int kr_timer_begin(kr_ctx* c) {
    if (!c) return KR_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->t0, c->stream));
    return KR_OK;
}

double kr_timer_end_ms(kr_ctx* c) {
    if (!c) return -1.0;
    float ms = 0;
    if (hipEventRecord(c->t1, c->stream) != hipSuccess) return -1.0;
    if (hipEventSynchronize(c->t1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, c->t0, c->t1) != hipSuccess) return -1.0;
    return (double)ms;
}

int kr_stage_enable(kr_ctx* c, int on) {
    if (!c) return KR_ERR_PARAM;
    c->stage_on = on != 0;
    c->stage_mask = ~0u;
    return KR_OK;
}

int kr_stage_select(kr_ctx* c, unsigned mask) {
    if (!c) return KR_ERR_PARAM;
    c->stage_on = mask != 0;
    c->stage_mask = mask;
    return KR_OK;
}

int kr_stage_reset(kr_ctx* c) {
    if (!c) return KR_ERR_PARAM;
    (void)hipDeviceSynchronize();
    resolve_stages(c);
    for (int i = 0; i < KR_ST_COUNT; i++) { c->stage_ms[i] = 0; c->stage_n[i] = 0; }
    return KR_OK;
}

double kr_stage_ms(kr_ctx* c, int stage) {
    if (!c || stage < 0 || stage >= KR_ST_COUNT) return -1.0;
    (void)hipDeviceSynchronize();
    resolve_stages(c);
    return c->stage_ms[stage];
}

int64_t kr_stage_launches(kr_ctx* c, int stage) {
    if (!c || stage < 0 || stage >= KR_ST_COUNT) return -1;
    return c->stage_n[stage];
}

int64_t kr_debug_fetch(kr_ctx* c, int id, int what, void* out, size_t cap_bytes) {
    if (!c) return KR_ERR_PARAM;
    auto it = c->genomes.find(id);
    if (it == c->genomes.end()) return fail(c, KR_ERR_PARAM, "unknown genome %d", id);
    Genome& G = it->second;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    Lane& ln = c->lanes[(c->next_lane + c->nlanes - 1) % c->nlanes];   // the lane of the latest sort
    const u32 nb = 1u << c->g.b;
    const void* src = nullptr;
    size_t esz = 8, n = 0;
    if (G.sl.empty()) return fail(c, KR_ERR_STATE, "genome %d has no slices", id);
    Slice& S0 = G.sl[c->nslices - 1];      // the slice sorted last (its scratch is still in the lane)
    u32 total = 0;
    HIPCHK(c, hipMemcpy(&total, (u32*)S0.off.p + nb, 4, hipMemcpyDeviceToHost));
    switch (what) {
    case 0: src = ln.codes.p; esz = 8; n = G.nwords + 2; break;
    case 1: src = ln.bad.p; esz = 4; n = G.nwords + 2; break;
    case 2: src = ln.base1.p; esz = 4; n = 257; break;
    case 3: src = S0.off.p; esz = 4; n = nb + 1; break;
    case 4: src = c->g.b > 8 ? ln.tmpkeys.p : S0.keys.p; esz = 8; n = total; break;
    case 5: src = S0.keys.p; esz = 8; n = total; break;
    default: return fail(c, KR_ERR_PARAM, "kr_debug_fetch: unknown selector %d", what);
    }
    if (n * esz > cap_bytes) return fail(c, KR_ERR_CAPACITY, "debug buffer too small");
    if (n) HIPCHK(c, hipMemcpy(out, src, n * esz, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

// ----------------------------------------------------------------------------
// host side: FASTA / sequence text -> upload buffer, one pass, reference reader semantics
// (kstream.py:458-583; see include/krisp_hip.h).  Pure CPU code; ctypes releases the GIL, so
// the Python layer ingests several files concurrently.
// ----------------------------------------------------------------------------
static inline bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

int64_t kr_fasta_to_bases(const uint8_t* text, size_t n, int universal_newlines, int one_shot, uint8_t* out,
                          size_t cap, int64_t* stats) {
    if (!text && n) return KR_ERR_PARAM;
    if (cap < n + 1) return KR_ERR_CAPACITY;
    size_t pos = 0, o = 0;
    int64_t nrec = 0;
    bool first_line = true, fasta = false, in_record = false, any_record = false;
    auto emit_separator = [&]() {
        if (any_record) out[o++] = '\n';
        any_record = true;
        nrec++;
    };
    while (pos < n) {
        // one line: [pos, eol)
        size_t eol = pos;
        if (universal_newlines) { while (eol < n && text[eol] != '\n' && text[eol] != '\r') eol++; }
        else { while (eol < n && text[eol] != '\n') eol++; }
        size_t next = eol;
        if (next < n) {
            if (universal_newlines && text[next] == '\r' && next + 1 < n && text[next + 1] == '\n') next += 2;
            else next += 1;
        }
        const uint8_t* ln = text + pos;
        size_t len = eol - pos;
        pos = next;
        if (first_line) {
            first_line = false;
            fasta = memchr(ln, '>', len) != nullptr;         // decided on the first line only
            if (one_shot) continue;                          // ... which the detection consumed
        }
        while (len && is_space(ln[0])) { ln++; len--; }
        while (len && is_space(ln[len - 1])) len--;
        if (!fasta) {                                        // every stripped line is a record
            emit_separator();
            memcpy(out + o, ln, len);
            o += len;
            continue;
        }
        if (len && ln[0] == '>') { in_record = false; continue; }
        if (!len) continue;
        if (!in_record) { emit_separator(); in_record = true; }
        memcpy(out + o, ln, len);
        o += len;
    }
    // RNA iff the first record holding T/t/U/u holds U/u and no T/t (kstream.py:481-508)
    int rna = -1;
    for (size_t i = 0, rs = 0; i <= o && rna < 0; i++) {
        if (i == o || out[i] == '\n') {
            bool t = false, u = false;
            for (size_t j = rs; j < i; j++) {
                uint8_t c = out[j];
                t |= (c == 'T' || c == 't');
                u |= (c == 'U' || c == 'u');
            }
            if (t) rna = 0; else if (u) rna = 1;
            rs = i + 1;
        }
    }
    int64_t special = 0;
    for (size_t i = 0; i < o; i++) {
        uint8_t c = out[i];
        if (rna == 1) {
            if (c == 'U') c = out[i] = 'T';
            else if (c == 'u') c = out[i] = 't';
        }
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'N': case 'a': case 'c': case 'g': case 't': case 'n':
        case '\n': break;
        default: special++;
        }
    }
    if (stats) { stats[0] = nrec; stats[1] = special; stats[2] = rna; stats[3] = fasta ? 1 : 0; }
    return (int64_t)o;
}

int64_t kr_debug_inversions(kr_ctx* c, int id) {
    if (!c) return KR_ERR_PARAM;
    auto it = c->genomes.find(id);
    if (it == c->genomes.end() || !it->second.sorted) return fail(c, KR_ERR_STATE, "genome %d not sorted", id);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = finalize(c, {&it->second});
    if (rc) return rc;
    if ((rc = ensure(c, c->nrec, 16))) return rc;
    HIPCHK(c, hipMemset(c->nrec.p, 0, 16));
    for (Slice& S : it->second.sl)
        if (S.count > 1)
            hipLaunchKernelGGL(k_count_inversions, dim3(4096), dim3(256), 0, c->stream, (const u64*)S.keys.p,
                               (u64)S.count, (u64*)c->nrec.p);
    u64 bad = 0;
    HIPCHK(c, hipMemcpy(&bad, c->nrec.p, 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipGetLastError());
    return (int64_t)bad;
}

int kr_debug_info(kr_ctx* c, int64_t* o) {
    if (!c || !o) return KR_ERR_PARAM;
    o[0] = c->g.b;
    o[1] = 1ll << c->g.b;
    o[2] = LS_T;
    o[3] = LS_CAP;
    o[4] = NWG;
    o[5] = c->overflow_segments;
    o[6] = c->fallback_launches;
    o[7] = (int64_t)c->nslices;
    return KR_OK;
}

}  // extern "C"
