// Host-only build of the library's text code (krisp_amd/csrc/h_text.inc: FASTA parser, IUPAC
// side-channel scan) under AddressSanitizer + UBSan, driven with random and adversarial inputs.
// Built and run by tests/test_host_glue.py::test_text_code_under_address_sanitizer (g++; the GPU
// cannot run sanitizers on this pool).  Every output buffer is allocated at exactly the size
// the C ABI documents, so an overrun of one byte is an ASan report.
#include <zlib.h>

#include <algorithm>
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
extern "C" {
#include "../../krisp_amd/csrc/h_text.inc"
#include "../../krisp_amd/csrc/h_pgzip.inc"
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
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
    printf("ASAN_HOST_OK %ld scans %ld renders %ld window renders %ld members on threads %ld refused\n", checks, renders, wrenders, members, refused);
    return 0;
}
