/* CPU ORACLE on packed keys -- test infrastructure, NOT product code.
 *
 * A plain-C restatement of the reference's krisp_fasta hot path
 * (grunwaldlab/krisp @ 2024_10_08) on 2-bit packed k-mers, used (a) by the
 * `-m gpu` parity tests as the checker at sizes the text-level oracle
 * (oracle/krisp_oracle.py) cannot reach, (b) by bench.py's cpu_baseline leg
 * (kind "port"; one genome per thread, up to 4).  krisp_amd/ never links or loads it.
 *
 * Parity status: PINNED through tests/test_kmer_oracle.py, which converts these
 * integer results back to the reference's text lines and compares them with
 * the golden vectors captured from the reference (tests/golden/) -- the five
 * sorted test_data .28mers files byte for byte (sha256), merged / filtered sets.
 *
 * Reference map (file:line into /root/reference/src/krisp):
 *   kro_sorted_keys : kstream/kstream.py:617-642 (_kmers), 734-766 (soft mask),
 *                     644-677 (both strands), 715-732 (disallow Nn), 805-832
 *                     (split L,-R), 83-119 (sort -t, -k1,1 -k3,3 + whole-line
 *                     tie-break => order (left, right, diag))
 *   kro_intersect   : krisp_fasta/shared.py:321-347, intersectAmplicons.py:232-310
 *                     (result = (left,right) pairs present in EVERY genome),
 *                     Amplicon.py:495-521 + filterAlignments.py:23-28 (filter)
 *   kro_collect     : shared.py:210-240 + Amplicon.py:170-187 (per-genome
 *                     multiplicity of each distinct sequence)
 *
 * Key format (shared with include/krisp_hip.h): base j of the string
 * left|right|diag sits in bits 63-2j..62-2j (A=0 C=1 G=2 T=3), low bits zero.
 * Input: ASCII bases, records separated by one '\n' byte.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t prefix, in_mask, out_mask; } kro_cand;
typedef struct { uint64_t key; uint32_t genome, count; } kro_record;

#define KRO_ERR_IUPAC   (-2)
#define KRO_ERR_ILLEGAL (-3)
#define KRO_ERR_CAP     (-4)
#define KRO_ERR_PARAM   (-5)

enum { C_A = 0, C_C = 1, C_G = 2, C_T = 3, C_N = 4, C_IUPAC = 5, C_ILLEGAL = 6, C_SEP = 7 };

static int classify(uint8_t c) {       /* case-insensitive class of a byte */
    switch (c) {
    case 'A': case 'a': return C_A;
    case 'C': case 'c': return C_C;
    case 'G': case 'g': return C_G;
    case 'T': case 't': return C_T;
    case 'N': case 'n': return C_N;
    case 'R': case 'r': case 'Y': case 'y': case 'M': case 'm': case 'K': case 'k':
    case 'S': case 's': case 'W': case 'w': case 'B': case 'b': case 'V': case 'v':
    case 'D': case 'd': case 'H': case 'h': return C_IUPAC;
    case '\n': return C_SEP;
    default: return C_ILLEGAL;
    }
}

static void radix_sort_u64(uint64_t* a, uint64_t* tmp, size_t n, int lowbit) {
    /* LSD, 8-bit digits, skipping the all-zero low part of the key */
    for (int sh = lowbit & ~7; sh < 64; sh += 8) {
        size_t cnt[257] = {0};
        for (size_t i = 0; i < n; i++) cnt[((a[i] >> sh) & 255) + 1]++;
        if (cnt[((a[0] >> sh) & 255) + 1] == n) continue;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) tmp[cnt[(a[i] >> sh) & 255]++] = a[i];
        memcpy(a, tmp, n * sizeof(uint64_t));
    }
}

/* Both-strand keys of every surviving window, sorted.  mode 0 = map soft mask
 * to upper case (krisp_fasta default), 1 = omit windows that are not
 * str.isupper().  Returns the count, or a negative KRO_ERR_*. */
static int64_t sorted_keys_impl(const uint8_t* bases, size_t n, int L, int D, int R, int mode, int topbits, uint64_t topval,
                                uint64_t* out, size_t cap);
int64_t kro_sorted_keys(const uint8_t* bases, size_t n, int L, int D, int R, int mode,
                        uint64_t* out, size_t cap) {
    return sorted_keys_impl(bases, n, L, D, R, mode, 0, 0, out, cap);
}
/* ... restricted to ONE key-space slice: the keys whose top `topbits` bits equal topval (the first topbits / 2 bases of
 * `left`).  What a full-size check of a human-scale genome can afford on the host: a 3 Gbp genome has 6e9 keys, one of
 * its 256 slices 2.3e7 (tests/test_gpu_fullsize.py compares that slice with the device's, bit for bit). */
int64_t kro_sorted_keys_slice(const uint8_t* bases, size_t n, int L, int D, int R, int mode, int topbits, uint64_t topval,
                              uint64_t* out, size_t cap) {
    if (topbits < 0 || topbits > 32 || (topbits & 1)) return KRO_ERR_PARAM;
    return sorted_keys_impl(bases, n, L, D, R, mode, topbits, topval, out, cap);
}
static int64_t sorted_keys_impl(const uint8_t* bases, size_t n, int L, int D, int R, int mode, int topbits, uint64_t topval,
                                uint64_t* out, size_t cap) {
    const int k = L + D + R;
    if (k < 1 || k > 32 || L < 0 || D < 0 || R < 0) return KRO_ERR_PARAM;
    const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint64_t dmask = D ? ((1ull << (2 * D)) - 1) : 0, rmask = R ? ((1ull << (2 * R)) - 1) : 0;
    int64_t last_sep = -1, last_ill = -1, last_low = -1, last_up = -1, last_n = -1, last_iu = -1;
    uint64_t fw = 0, rc = 0;
    size_t m = 0;
    for (size_t j = 0; j < n; j++) {
        uint8_t c = bases[j];
        int cls = classify(c);
        int lower = (c >= 'a' && c <= 'z'), upper = (c >= 'A' && c <= 'Z');
        if (cls == C_SEP) last_sep = (int64_t)j;
        if (cls == C_ILLEGAL) last_ill = (int64_t)j;
        if (cls == C_N) last_n = (int64_t)j;
        if (cls == C_IUPAC) last_iu = (int64_t)j;
        if (lower) last_low = (int64_t)j;
        if (upper) last_up = (int64_t)j;
        uint64_t code = (cls <= C_T) ? (uint64_t)cls : 0;
        fw = ((fw << 2) | code) & kmask;
        rc = (rc >> 2) | ((3 - code) << (2 * (k - 1)));
        int64_t start = (int64_t)j - k + 1;
        if (start < 0 || last_sep >= start) continue;           /* no such window */
        if (mode == 1 && (last_low >= start || last_up < start)) continue;  /* not isupper() */
        if (last_ill >= start) return KRO_ERR_ILLEGAL;           /* KeyError in _get_complement */
        if (last_n >= start) continue;                           /* disallow "Nn" */
        if (last_iu >= start) return KRO_ERR_IUPAC;              /* kept by the reference; not packable */
        if (m + 2 > cap) return KRO_ERR_CAP;
        for (int s = 0; s < 2; s++) {
            uint64_t w = s ? rc : fw;
            uint64_t left = (D + R >= 32) ? 0 : (w >> (2 * (D + R)));
            uint64_t diag = (w >> (2 * R)) & dmask, right = w & rmask;
            uint64_t key = ((D + R >= 32) ? 0 : (left << (2 * (D + R)))) | (right << (2 * D)) | diag;
            key = (k == 32) ? key : (key << (64 - 2 * k));
            if (topbits && (key >> (64 - topbits)) != topval) continue;
            out[m++] = key;
        }
    }
    if (m > 1) {
        uint64_t* tmp = (uint64_t*)malloc(m * sizeof(uint64_t));
        if (!tmp) return KRO_ERR_CAP;
        radix_sort_u64(out, tmp, m, 64 - 2 * k);
        free(tmp);
    }
    return (int64_t)m;
}

static inline uint64_t prefix_of(uint64_t key, int L, int R) {
    int pb = 2 * (L + R);
    return pb == 0 ? 0 : (pb >= 64 ? key : (key >> (64 - pb)) << (64 - pb));
}

static inline uint64_t diag_mask_of(uint64_t key, int L, int D, int R) {
    /* bit 4c+b set for base b at diagnostic column c (D <= 16) */
    uint64_t m = 0;
    for (int c = 0; c < D; c++) {
        int b = (int)((key >> (62 - 2 * (L + R + c))) & 3);
        m |= 1ull << (4 * c + b);
    }
    return m;
}

static int passes_filter(uint64_t in_mask, uint64_t out_mask, int D) {
    for (int c = 0; c < D; c++)
        if ((((in_mask & out_mask) >> (4 * c)) & 15) == 0) return 1;
    return 0;
}

/* n-way intersection of sorted key arrays on the (left,right) prefix. */
int64_t kro_intersect(const uint64_t* const* keys, const int64_t* counts, int n,
                      const uint8_t* is_ingroup, int L, int D, int R, int apply_filter,
                      kro_cand* out, size_t cap) {
    if (n < 1 || D > 16) return KRO_ERR_PARAM;
    int64_t* p = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    size_t m = 0;
    for (;;) {
        int done = 0;
        uint64_t top = 0;
        for (int g = 0; g < n; g++) {
            if (p[g] >= counts[g]) { done = 1; break; }
            uint64_t q = prefix_of(keys[g][p[g]], L, R);
            if (q > top) top = q;
        }
        if (done) break;
        int all = 1;
        for (int g = 0; g < n; g++) {
            while (p[g] < counts[g] && prefix_of(keys[g][p[g]], L, R) < top) p[g]++;
            if (p[g] >= counts[g]) { done = 1; break; }
            if (prefix_of(keys[g][p[g]], L, R) != top) all = 0;
        }
        if (done) break;
        if (!all) continue;
        uint64_t im = 0, om = 0;
        for (int g = 0; g < n; g++) {
            while (p[g] < counts[g] && prefix_of(keys[g][p[g]], L, R) == top) {
                uint64_t dm = diag_mask_of(keys[g][p[g]], L, D, R);
                if (is_ingroup[g]) im |= dm; else om |= dm;
                p[g]++;
            }
        }
        if (apply_filter && D > 0 && !passes_filter(im, om, D)) continue;
        if (m >= cap) { free(p); return KRO_ERR_CAP; }
        out[m].prefix = top; out[m].in_mask = im; out[m].out_mask = om; m++;
    }
    free(p);
    return (int64_t)m;
}

/* (key, genome, multiplicity) of every distinct key under each candidate prefix. */
int64_t kro_collect(const uint64_t* const* keys, const int64_t* counts, int n,
                    const kro_cand* cands, int64_t ncand, int L, int D, int R,
                    kro_record* out, size_t cap) {
    (void)D;
    size_t m = 0;
    for (int g = 0; g < n; g++) {
        int64_t i = 0;
        for (int64_t c = 0; c < ncand; c++) {
            uint64_t q = cands[c].prefix;
            int64_t lo = i, hi = counts[g];          /* lower_bound from the last hit */
            while (lo < hi) {
                int64_t mid = lo + (hi - lo) / 2;
                if (prefix_of(keys[g][mid], L, R) < q) lo = mid + 1; else hi = mid;
            }
            i = lo;
            while (i < counts[g] && prefix_of(keys[g][i], L, R) == q) {
                uint64_t key = keys[g][i];
                uint32_t cnt = 0;
                while (i < counts[g] && keys[g][i] == key) { cnt++; i++; }
                if (m >= cap) return KRO_ERR_CAP;
                out[m].key = key; out[m].genome = (uint32_t)g; out[m].count = cnt; m++;
            }
        }
    }
    return (int64_t)m;
}
