// Host-only build of the library's text code (krisp_amd/csrc/h_text.inc: FASTA parser, IUPAC
// side-channel scan; h_pgzip.inc: one gzip member on several threads; h_inflate.inc: gzip / BGZF / bz2 decoders, the bz2
// 48-bit mark scanner and its re-wrapping of blocks, the BGZF member splitter) under AddressSanitizer + UBSan, driven with
// random and adversarial inputs.
// Built and run by tests/test_host_glue.py::test_text_code_under_address_sanitizer (g++; the GPU
// cannot run sanitizers on this pool).  Every output buffer is allocated at exactly the size
// the C ABI documents, so an overrun of one byte is an ASan report.
#include <dlfcn.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <map>
#include <mutex>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "krisp_hip.h"

static thread_local std::string g_last_error;
typedef unsigned int u32;               // (k_keys.inc's, for the host code that shares the library's spelling)
typedef unsigned long long u64;
// stand-ins for the two pinned-memory calls of h_ingest.inc (HIP there): plain heap blocks, so that ASan sees every byte
static void* host_alloc(size_t bytes) { return malloc(bytes); }
extern "C" {
void kr_host_free(void* p) { free(p); }
#include "../../krisp_amd/csrc/h_text.inc"
#include "../../krisp_amd/csrc/h_pgzip.inc"
#include "../../krisp_amd/csrc/h_inflate.inc"      // gzip / BGZF / bz2 decoders and splitters, read_text (round 5)
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}

// ---- h_inflate.inc: files the library did not write.  Sound gzip / BGZF / bz2 files decode to their text on every route
// (one thread and several; the bz2 block route against the sequential decoder); truncated, bit-flipped, overwritten and
// MARK-SPOOFING inputs -- a payload that holds the 48-bit bz2 block or end mark by chance, on any bit -- and BGZF headers
// that lie about their member's length: whatever the verdict, nothing is read or written out of bounds (every input lives
// in a heap block of exactly its length), a file that decodes "fine" on the block route is what the sequential decoder
// makes of it, and trailing bytes behind a complete bz2 stream are ignored as Python's bz2 module ignores them.
typedef int (*bz_compress_t)(char*, unsigned*, char*, unsigned, int, int, int);
static std::vector<uint8_t> bz_make(bz_compress_t comp, const std::vector<uint8_t>& text, int level) {
    std::vector<uint8_t> out(text.size() + text.size() / 50 + 1000);
    unsigned olen = (unsigned)out.size();
    static char none = 0;
    if (comp((char*)out.data(), &olen, text.empty() ? &none : (char*)text.data(), (unsigned)text.size(), level, 0, 0) != 0) return {};
    out.resize(olen);
    return out;
}
static std::vector<uint8_t> gz_make(const std::vector<uint8_t>& text, int level) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return {};
    std::vector<uint8_t> gz(deflateBound(&zs, (uLong)text.size()) + 64);
    zs.next_in = (Bytef*)text.data(); zs.avail_in = (uInt)text.size();
    zs.next_out = gz.data(); zs.avail_out = (uInt)gz.size();
    deflate(&zs, Z_FINISH);
    gz.resize(gz.size() - zs.avail_out);
    deflateEnd(&zs);
    return gz;
}
// BGZF: members of <= 60000 bytes of text, each with the 'B' 'C' extra field naming its length, then the empty end member
static std::vector<uint8_t> bgzf_make(const std::vector<uint8_t>& text) {
    std::vector<uint8_t> out;
    static const uint8_t none = 0;
    auto member = [&](const uint8_t* p, size_t len) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> raw(deflateBound(&zs, (uLong)len) + 64);
        zs.next_in = (Bytef*)(len ? p : &none); zs.avail_in = (uInt)len;
        zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
        deflate(&zs, Z_FINISH);
        raw.resize(raw.size() - zs.avail_out);
        deflateEnd(&zs);
        const uint32_t bsize = (uint32_t)(raw.size() + 25), crc = (uint32_t)crc32(0, len ? p : &none, (uInt)len), isz = (uint32_t)len;
        const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
        out.insert(out.end(), head, head + 18);
        out.insert(out.end(), raw.begin(), raw.end());
        for (int k = 0; k < 4; k++) out.push_back((uint8_t)(crc >> (8 * k)));
        for (int k = 0; k < 4; k++) out.push_back((uint8_t)(isz >> (8 * k)));
    };
    for (size_t a = 0; a < text.size(); a += 60000) member(text.data() + a, std::min<size_t>(60000, text.size() - a));
    member(nullptr, 0);                         // the empty end mark
    return out;
}
static int inflate_fuzz(long* sound, long* damaged, long* agreed) {
    void* h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
    bz_compress_t comp = h ? (bz_compress_t)dlsym(h, "BZ2_bzBuffToBuffCompress") : nullptr;
    const bool have_bz = comp && libbz2().ok;
    auto decode = [&](int kind, const std::vector<uint8_t>& blobv, std::vector<uint8_t>& text, int* rc_seq, std::vector<uint8_t>* seq) {
        // the input in a block of EXACTLY its size
        std::unique_ptr<uint8_t[]> blob(new uint8_t[blobv.size() ? blobv.size() : 1]);
        if (!blobv.empty()) memcpy(blob.get(), blobv.data(), blobv.size());
        RawBuf out;
        int64_t members = 0;
        int used = 0, rc;
        if (kind == 0) rc = inflate_gzip(blob.get(), blobv.size(), out, &members, &used);
        else if (kind == 1) {
            rc = inflate_bgzf(blob.get(), blobv.size(), out, &members, &used);
            if (rc == 0) rc = inflate_gzip(blob.get(), blobv.size(), out, &members, &used);
            else if (rc == 1) rc = KR_OK;
        } else {
            rc = inflate_bz2(blob.get(), blobv.size(), out, &members);
            if (seq) {              // the sequential decoder's verdict on the same bytes
                RawBuf o2;
                int64_t s2 = 0;
                *rc_seq = o2.reserve(blobv.size() * 5 + (1 << 16)) ? bunzip_range(blob.get(), blobv.size(), o2, &s2) : KR_ERR_CAPACITY;
                seq->assign(o2.p, o2.p + (*rc_seq == KR_OK ? o2.len : 0));
            }
        }
        if (rc == KR_OK && out.len > out.cap) return -100;
        text.assign(out.p, out.p + (rc == KR_OK ? out.len : 0));
        return rc;
    };
    for (int it = 0; it < 36; it++) {
        const size_t n = it < 3 ? (size_t)it : (it % 3 == 0 ? 150000 + rnd() % 400000 : 100 + rnd() % 90000);
        std::vector<uint8_t> text(n);
        for (size_t i = 0; i < n; i++) text[i] = (uint8_t)(i % 61 == 60 ? '\n' : (it % 4 == 3 ? rnd() : "ACGTacgtN"[rnd() % 9]));
        for (int kind = 0; kind < 3; kind++) {
            if (kind == 2 && !have_bz) continue;
            std::vector<uint8_t> blob = kind == 0 ? gz_make(text, 1 + it % 9) : kind == 1 ? bgzf_make(text) : bz_make(comp, text, 1);
            if (kind == 2 && it % 5 == 4) {         // several streams, as pbzip2 writes them
                std::vector<uint8_t> t2(text.begin() + n / 2, text.end()), t1(text.begin(), text.begin() + n / 2);
                blob = bz_make(comp, t1, 1);
                const std::vector<uint8_t> b2 = bz_make(comp, t2, 2);
                blob.insert(blob.end(), b2.begin(), b2.end());
            }
            if (blob.empty()) return 30;
            for (const char* threads : {"4", "1"}) {
                setenv("KRISP_INGEST_THREADS", threads, 1);
                std::vector<uint8_t> got, seq;
                int rs = 0;
                const int rc = decode(kind, blob, got, &rs, kind == 2 ? &seq : nullptr);
                if (rc != KR_OK || got != text) { printf("sound file of kind %d (%zu bytes of text, threads %s): rc %d\n", kind, n, threads, rc); return 31; }
                if (kind == 2 && (rs != KR_OK || seq != text)) return 32;
                (*sound)++;
            }
            setenv("KRISP_INGEST_THREADS", "4", 1);
            for (int dmg = 0; dmg < 14; dmg++) {
                std::vector<uint8_t> b(blob);
                const size_t m = b.size();
                bool tail_only = false;
                switch (dmg % 7) {
                case 0: b.resize(rnd() % (m + 1)); break;                                        // truncated anywhere
                case 1: if (m) b[rnd() % m] ^= (uint8_t)(1u << (rnd() & 7)); break;              // one bit
                case 2: for (int q = 0; q < 8 && m; q++) b[rnd() % m] = (uint8_t)rnd(); break;   // bytes overwritten
                case 3: {                                                                         // a spoofed mark inside the payload, on any bit
                    if (m < 40) break;
                    const uint64_t mark = (rnd() & 1) ? 0x314159265359ull : 0x177245385090ull;
                    const size_t bit = (20 + rnd() % (m - 32)) * 8 + (rnd() & 7);
                    for (int q = 0; q < 48; q++) {
                        const size_t at = bit + (size_t)q;
                        const uint8_t msk = (uint8_t)(0x80u >> (at & 7));
                        if ((mark >> (47 - q)) & 1) b[at >> 3] |= msk; else b[at >> 3] &= (uint8_t)~msk;
                    }
                    break;
                }
                case 4: {                                                                         // trailing bytes behind the file
                    const char* tails[] = {"XYZ\n", "\0\0\0", "BZh9garbage-not-a-stream", "BZx"};
                    const char* t = tails[rnd() % 4];
                    const size_t tl = t[0] ? strlen(t) : 3;
                    b.insert(b.end(), (const uint8_t*)t, (const uint8_t*)t + tl);
                    tail_only = true;
                    break;
                }
                case 5: if (m > 30) { const size_t a = 10 + rnd() % (m - 20); b.erase(b.begin() + a, b.begin() + a + 1 + rnd() % 9); } break;   // bytes missing
                case 6: if (m > 18) { b[16] = (uint8_t)rnd(); b[17] = (uint8_t)rnd(); } break;   // a BGZF header that lies about its length
                }
                std::vector<uint8_t> got, seq;
                int rs = 0;
                const int rc = decode(kind, b, got, &rs, kind == 2 ? &seq : nullptr);
                if (rc == -100) return 33;
                if (kind == 2) {
                    // the block route may decline, never disagree: same verdict class and, when both decode, the same text
                    if ((rc == KR_OK) != (rs == KR_OK) || (rc == KR_OK && got != seq)) {
                        printf("bz2 damage %d: block route rc %d (%zu bytes), sequential rc %d (%zu bytes)\n", dmg % 7, rc, got.size(), rs, seq.size());
                        return 34;
                    }
                    if (tail_only && (rc != KR_OK || got != text)) { printf("bz2: trailing bytes behind a stream are not ignored (rc %d)\n", rc); return 35; }
                    (*agreed)++;
                }
                (*damaged)++;
            }
        }
    }
    return 0;
}

int main() {
    const char* alphabets[] = {"ACGT", "ACGTacgtNn\n", "ACGTACGT>\n\r \t", "ACGTURYKMSWBDHVXx-.\n>", "\n\n>\r\n"};
    long checks = 0;
    for (int it = 0; it < 20000; it++) {
        const char* al = alphabets[rnd() % 5];
        const size_t na = strlen(al);
        const size_t n = rnd() % (it % 50 == 0 ? 5000 : 200);
        std::vector<uint8_t> text(n);
        for (auto& c : text) c = (uint8_t)al[rnd() % na];
        if (it % 7 == 0 && n > 3) text[rnd() % n] = (uint8_t)(rnd() & 0xFF);     // any byte at all
        for (int universal = 0; universal < 2; universal++)
            for (int one_shot = 0; one_shot < 2; one_shot++) {
                std::vector<uint8_t> out(n + 1);                      // "out needs n + 1 bytes"
                int64_t stats[4];
                const int64_t m = kr_fasta_to_bases(n ? text.data() : nullptr, n, universal, one_shot, out.data(), out.size(), stats);
                if (m < 0 || (size_t)m > n) { printf("bad length %lld for n %zu\n", (long long)m, n); return 1; }
                // too small a buffer is refused, not overrun
                if (n && kr_fasta_to_bases(text.data(), n, universal, one_shot, out.data(), n, stats) != KR_ERR_CAPACITY) return 2;
                // the side-channel scan over what the parser produced, every k, both soft-mask rules
                for (int k = 1; k <= 40; k += 1 + rnd() % 6)
                    for (int omit = 0; omit < 2; omit++) {
                        int bad = 0;
                        const int64_t c0 = kr_scan_special(out.data(), (size_t)m, k, omit, nullptr, 0, &bad);
                        if (c0 >= 0) {
                            std::vector<uint64_t> st((size_t)c0);            // exactly as many as announced
                            const int64_t c1 = kr_scan_special(out.data(), (size_t)m, k, omit, st.data(), st.size(), &bad);
                            if (c1 != c0) { printf("count changed %lld -> %lld\n", (long long)c0, (long long)c1); return 3; }
                            for (uint64_t s : st)
                                if (s + (uint64_t)k > (uint64_t)m) { printf("window beyond the buffer\n"); return 4; }
                            if (c0 > 1 && !std::is_sorted(st.begin(), st.end())) return 5;
                        } else if (c0 != KR_ERR_KEY && c0 != KR_ERR_HOST) {
                            return 6;
                        }
                        checks++;
                    }
            }
    }
    // the record renderer: random groups in (key, label) order, every geometry, with and without an ingroup, both
    // alignment forms; records out of order or with ids outside the tables are refused, never read past
    long renders = 0;
    for (int it = 0; it < 3000; it++) {
        const int L = rnd() % 9, D = rnd() % 4, R = rnd() % 6;
        if (L + D + R == 0) continue;
        const int k = L + D + R;
        const size_t ngen = 1 + rnd() % 5, nlab = 1 + rnd() % ngen;
        std::vector<uint32_t> label_of(ngen);
        for (auto& v : label_of) v = rnd() % nlab;
        std::vector<std::string> names(nlab);
        std::vector<const char*> text(nlab);
        for (size_t i = 0; i < nlab; i++) { names[i] = "lab" + std::to_string(i); text[i] = names[i].c_str(); }
        std::vector<uint8_t> is_in(nlab);
        for (auto& v : is_in) v = rnd() & 1;
        std::vector<kr_record> recs;
        // ascending distinct keys of k bases (the groups are whatever shares the (left,right) prefix)
        const uint64_t space = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
        uint64_t v = rnd() % 3;
        const int namps = rnd() % 12;
        for (int a = 0; a < namps && v <= space; a++) {
            const uint64_t key = k >= 32 ? v : (v << (64 - 2 * k));
            std::vector<std::pair<uint32_t, uint32_t>> gs;          // genomes of this Amplicon in label order
            for (uint32_t q = 0; q < ngen; q++)
                if (rnd() & 1) gs.push_back({label_of[q], q});
            if (gs.empty()) gs.push_back({label_of[0], 0});
            std::sort(gs.begin(), gs.end());
            for (auto& pr : gs) recs.push_back(kr_record{key, pr.second, 1 + rnd() % 3});
            const uint64_t step = 1 + rnd() % 5;
            if (space - v < step) break;
            v += step;
        }
        for (int with_in = 0; with_in < 2; with_in++)
            for (int dot = 0; dot < 2; dot++) {
                char *csv = nullptr, *al = nullptr;
                size_t nc = 0, na = 0;
                const int64_t r = kr_render_records(recs.data(), recs.size(), L, D, R, label_of.data(), ngen, text.data(), nlab,
                                                    with_in ? is_in.data() : nullptr, dot, &csv, &nc, &al, &na);
                if (r < 0 && r != KR_ERR_HOST) { printf("render failed %lld\n", (long long)r); return 7; }
                if (r >= 0 && (nc < 28 || csv[nc - 1] != '\n')) return 8;
                kr_text_free(csv);
                kr_text_free(al);
                renders++;
            }
        if (recs.size() > 1) {
            std::swap(recs[0], recs[recs.size() - 1]);
            char *csv = nullptr, *al = nullptr;
            size_t nc = 0, na = 0;
            const int64_t r = kr_render_records(recs.data(), recs.size(), L, D, R, label_of.data(), ngen, text.data(), nlab, nullptr, 0,
                                                &csv, &nc, &al, &na);
            if (r >= 0 && recs[0].key != recs[recs.size() - 1].key) { kr_text_free(csv); kr_text_free(al); }
            else if (r >= 0) { kr_text_free(csv); kr_text_free(al); }
            recs[0].genome = 1000;
            if (kr_render_records(recs.data(), recs.size(), L, D, R, label_of.data(), ngen, text.data(), nlab, nullptr, 0, &csv, &nc,
                                  &al, &na) != KR_ERR_PARAM) return 9;
        }
    }
    // kr_render_windows: window rows of random geometry (flanks beyond 32 letters too), group numbers in any order, the rows of
    // a group interleaved with its neighbours', one (left,right) under two numbers (declined), a genome out of range
    long wrenders = 0;
    for (int it = 0; it < 300; it++) {
        const int L = 1 + rnd() % 40, D = rnd() % 20, R = 1 + rnd() % 40, k = L + D + R;
        const uint32_t ngen = 1 + rnd() % 5, nlab = 1 + rnd() % ngen;
        std::vector<uint32_t> label_of(ngen);
        for (auto& l : label_of) l = rnd() % nlab;
        std::vector<std::string> names(nlab);
        std::vector<const char*> text(nlab);
        for (size_t i = 0; i < nlab; i++) { names[i] = "w" + std::to_string(i); text[i] = names[i].c_str(); }
        std::vector<uint8_t> is_in(nlab);
        for (auto& v : is_in) v = rnd() & 1;
        const int ngroups = rnd() % 6;
        std::vector<uint8_t> rows;
        std::vector<uint32_t> cand, gen;
        for (int g = 0; g < ngroups; g++) {
            std::string fl(k, 'A');
            for (int q = 0; q < k; q++) fl[q] = "ACGT"[rnd() & 3];
            fl[0] = "ACGT"[g & 3];                          // (distinct flanks per group, mostly)
            if (L > 1) fl[1] = "ACGT"[(g >> 2) & 3];
            const int nm = 1 + rnd() % 6;
            for (int m = 0; m < nm; m++) {
                std::string w = fl;
                for (int q = L; q < L + D; q++) w[q] = "ACGT"[rnd() & 3];
                rows.insert(rows.end(), w.begin(), w.end());
                cand.push_back((uint32_t)g | ((rnd() & 7) == 0 ? 0u : 0u));
                gen.push_back(rnd() % ngen);
            }
        }
        // interleave: swap random rows (the renderer orders by group number itself)
        const size_t n = cand.size();
        for (size_t q = 0; q + 1 < n; q++) {
            const size_t o = q + rnd() % (n - q);
            std::swap(cand[q], cand[o]);
            std::swap(gen[q], gen[o]);
            for (int b = 0; b < k; b++) std::swap(rows[q * k + b], rows[o * k + b]);
        }
        for (int with_in = 0; with_in < 2; with_in++)
            for (int dot = 0; dot < 2; dot++) {
                char *csv = nullptr, *al = nullptr;
                size_t nc = 0, na = 0;
                const int64_t r = kr_render_windows(rows.data(), n, L, D, R, cand.data(), gen.data(), label_of.data(), ngen, text.data(),
                                                    nlab, with_in ? is_in.data() : nullptr, dot, it & 1, &csv, &nc, &al, &na);
                if (r < 0 && r != KR_ERR_HOST) { printf("window render failed %lld\n", (long long)r); return 10; }
                if (r >= 0 && (nc < 28 || csv[nc - 1] != '\n')) return 11;
                kr_text_free(csv);
                kr_text_free(al);
                wrenders++;
            }
        if (n) {
            gen[0] = 1000;
            char *csv = nullptr, *al = nullptr;
            size_t nc = 0, na = 0;
            if (kr_render_windows(rows.data(), n, L, D, R, cand.data(), gen.data(), label_of.data(), ngen, text.data(), nlab, nullptr, 0, 0,
                                  &csv, &nc, &al, &na) != KR_ERR_PARAM) return 12;
        }
    }
    // ---- one gzip member on several threads (h_pgzip.inc): texts of several kinds, every level and strategy, chunks as
    // small as they go; then the same members damaged -- whatever comes back, nothing may be read or written out of bounds,
    // and a member that is "done" must be the text
    long members = 0, refused = 0;
    for (int it = 0; it < 60; it++) {
        const size_t n = 200000 + rnd() % 1500000;
        std::vector<uint8_t> text(n);
        const int kind = it % 5;
        for (size_t i = 0; i < n; i++) {
            if (kind == 0) text[i] = (uint8_t)"ACGT"[rnd() & 3];
            else if (kind == 1) text[i] = (uint8_t)(i % 71 == 70 ? '\n' : "ACGTN"[rnd() % 5]);
            else if (kind == 2) text[i] = (uint8_t)rnd();
            else if (kind == 3) text[i] = (uint8_t)(i > 40000 && (rnd() & 7) ? text[i - 1 - rnd() % 40000] : "ACGT"[rnd() & 3]);
            else text[i] = (uint8_t)((i / 5000) & 1 ? 'A' : "ACGT"[rnd() & 3]);
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        const int level = it % 10, strategy = (it / 10) % 4 == 3 ? Z_FIXED : ((it / 10) % 4 == 2 ? Z_RLE : Z_DEFAULT_STRATEGY);
        if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 1 + rnd() % 9, strategy) != Z_OK) return 20;
        std::vector<uint8_t> gz(deflateBound(&zs, (uLong)n) + 64);
        zs.next_in = text.data(); zs.avail_in = (uInt)n;
        zs.next_out = gz.data(); zs.avail_out = (uInt)gz.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return 21;
        gz.resize(gz.size() - zs.avail_out);
        deflateEnd(&zs);
        for (int damage = 0; damage < 4; damage++) {
            std::vector<uint8_t> blob(gz);                  // (exactly as long as the member: a read behind it is an ASan report)
            if (damage == 1) blob[blob.size() / 3 + rnd() % (blob.size() / 3)] ^= (uint8_t)(1 + rnd() % 255);
            if (damage == 2) blob.resize(blob.size() / 2 + rnd() % (blob.size() / 2));
            if (damage == 3) for (int q = 0; q < 20; q++) blob[20 + rnd() % (blob.size() - 20)] = (uint8_t)rnd();
            PgzMember m;
            const int r = pgz_decode_member(blob.data(), blob.size(), 3, 65536, m);
            if (r == 1) {
                std::vector<uint8_t> out(m.total);
                const int e = pgz_emit(m, out.data(), 3);
                if (e == 1 && (m.total != n || memcmp(out.data(), text.data(), n) != 0)) { printf("another text accepted\n"); return 22; }
                if (e == 1) members++; else refused++;
                if (damage == 0 && e != 1) { printf("sound member refused by its checksum\n"); return 23; }
            } else {
                refused++;
            }
        }
    }
    long sound = 0, damaged = 0, agreed = 0;
    const int fr = inflate_fuzz(&sound, &damaged, &agreed);
    if (fr) { printf("inflate fuzz failed: %d\n", fr); return fr; }
    printf("ASAN_HOST_OK %ld scans %ld renders %ld window renders %ld members on threads %ld refused; inflate: %ld sound files, "
           "%ld damaged, %ld bz2 verdicts agreed\n", checks, renders, wrenders, members, refused, sound, damaged, agreed);
    return 0;
}
