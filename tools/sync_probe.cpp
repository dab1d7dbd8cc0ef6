// How fast does a DEFLATE decoder that starts at an arbitrary bit of a BGZF block fall into step with the true symbol sequence?
// (design input for the speculative wave-per-block token pass: chunk size vs. the share of chunks whose speculative exit is wrong)
//   g++ -O2 -std=c++17 -o build/sync_probe tools/sync_probe.cpp && build/sync_probe file.bam [max_blocks]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Code {
    int count[16], first[16], off[16];
    uint16_t sym[320];
    int maxlen;
    bool build(const uint8_t* lens, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) ++count[lens[i]];
        count[0] = 0;
        int code = 0, o = 0;
        maxlen = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + count[l - 1]) << 1;
            first[l] = code; off[l] = o; o += count[l];
            if (count[l]) maxlen = l;
        }
        int cur[16];
        std::memcpy(cur, off, sizeof cur);
        for (int i = 0; i < n; ++i) if (lens[i]) sym[cur[lens[i]]++] = (uint16_t)i;
        return true;
    }
};
struct Bits {
    const uint8_t* p; size_t nbits; size_t pos;
    uint32_t peek(int k) const {  // k <= 24
        uint64_t v = 0;
        const size_t b = pos >> 3;
        for (int i = 0; i < 5; ++i) if ((b + i) * 8 < nbits + 64) v |= (uint64_t)p[b + i] << (8 * i);
        return (uint32_t)((v >> (pos & 7)) & ((1u << k) - 1));
    }
    uint32_t take(int k) { const uint32_t v = peek(k); pos += k; return v; }
};
static int decode(Bits& b, const Code& c) {
    int code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code << 1) | (int)b.take(1);
        const int d = code - c.first[l];
        if (d >= 0 && d < c.count[l]) return c.sym[c.off[l] + d];
    }
    return -1;
}
static const int LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const int LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const int DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
// one lit/len symbol (with its distance half) at b.pos; returns 0 literal, 1 match, 2 end of block, -1 invalid
static int step(Bits& b, const Code& ll, const Code& dd) {
    const int s = decode(b, ll);
    if (s < 0) return -1;
    if (s < 256) return 0;
    if (s == 256) return 2;
    if (s > 285) return -1;
    b.pos += LEXT[s - 257];
    const int d = decode(b, dd);
    if (d < 0 || d > 29) return -1;
    b.pos += DEXT[d];
    return 1;
}
int main(int argc, char** argv) {
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    const long maxb = argc > 2 ? std::atol(argv[2]) : 200;
    const int CHS[5] = {128, 256, 512, 1024, 2048};
    long nchunks[5] = {0}, nfail[5] = {0};
    long long tot_sym = 0, tot_lit = 0, tot_match = 0, tot_bits = 0, tot_dblocks = 0, tot_long_ll = 0, tot_sym_long = 0;
    std::vector<long> syncdist;  // symbols until in step, sampled at 512-bit boundaries
    std::vector<uint8_t> buf;
    for (long bi = 0; bi < maxb; ++bi) {
        uint8_t h[18];
        if (std::fread(h, 1, 18, f) != 18) break;
        const int bsize = (h[16] | (h[17] << 8)) + 1;
        buf.assign((size_t)bsize - 18 + 16, 0);
        if (std::fread(buf.data(), 1, (size_t)bsize - 18, f) != (size_t)bsize - 18) break;
        const size_t clen = (size_t)bsize - 18 - 8;
        Bits b{buf.data(), clen * 8, 0};
        for (;;) {
            const int last = (int)b.take(1), type = (int)b.take(2);
            if (type == 0) { b.pos = (b.pos + 7) & ~(size_t)7; const int len = (int)b.take(16); b.take(16); b.pos += (size_t)len * 8; if (last) break; continue; }
            uint8_t lens[320] = {0};
            int nlen = 288, ndist = 30;
            if (type == 1) { for (int i = 0; i < 144; ++i) lens[i] = 8; for (int i = 144; i < 256; ++i) lens[i] = 9; for (int i = 256; i < 280; ++i) lens[i] = 7; for (int i = 280; i < 288; ++i) lens[i] = 8; for (int i = 0; i < 30; ++i) lens[288 + i] = 5; }
            else {
                nlen = (int)b.take(5) + 257; ndist = (int)b.take(5) + 1;
                const int ncode = (int)b.take(4) + 4;
                static const int ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[ord[i]] = (uint8_t)b.take(3);
                Code cc; cc.build(cl, 19);
                int i = 0, prev = 0;
                while (i < nlen + ndist) {
                    const int s = decode(b, cc);
                    if (s < 16) { lens[i++] = (uint8_t)s; prev = s; }
                    else if (s == 16) { int r = 3 + (int)b.take(2); while (r--) lens[i++] = (uint8_t)prev; }
                    else if (s == 17) { int r = 3 + (int)b.take(3); while (r--) lens[i++] = 0; prev = 0; }
                    else { int r = 11 + (int)b.take(7); while (r--) lens[i++] = 0; prev = 0; }
                }
            }
            Code ll, dd;
            ll.build(lens, nlen); dd.build(lens + nlen, ndist);
            for (int i = 0; i < nlen; ++i) if (lens[i] > 11) ++tot_long_ll;
            ++tot_dblocks;
            // the true symbol starts of this deflate block
            const size_t start = b.pos;
            std::vector<uint8_t> is_start;
            std::vector<size_t> starts;
            for (;;) {
                starts.push_back(b.pos);
                const size_t at = b.pos;
                const int r = step(b, ll, dd);
                if (r < 0) { std::fprintf(stderr, "bad stream in block %ld\n", bi); return 2; }
                ++tot_sym;
                if (r == 0) ++tot_lit; else if (r == 1) ++tot_match;
                { Bits t{buf.data(), clen * 8, at}; const int s = decode(t, ll); if (s >= 0 && lens[s] > 11) ++tot_sym_long; }
                if (r == 2) break;
            }
            const size_t end = b.pos;
            tot_bits += (long long)(end - start);
            is_start.assign(end - start + 1, 0);
            for (size_t s : starts) is_start[s - start] = 1;
            // speculative decoders at every chunk boundary: in step before the end of the chunk?
            for (int ci = 0; ci < 5; ++ci) {
                const int CH = CHS[ci];
                for (size_t c0 = start + CH; c0 + CH <= end; c0 += CH) {
                    Bits t{buf.data(), clen * 8, c0};
                    bool ok = false;
                    long nsym = 0;
                    while (t.pos < c0 + CH) {
                        if (is_start[t.pos - start]) { ok = true; break; }
                        const int r = step(t, ll, dd);
                        ++nsym;
                        if (r < 0 || r == 2 || t.pos >= end) break;
                    }
                    if (!ok && t.pos >= c0 + CH && t.pos < end && is_start[t.pos - start]) ok = true;  // in step exactly at the exit
                    ++nchunks[ci];
                    if (!ok) ++nfail[ci];
                    if (CH == 2048 && ok) syncdist.push_back((long)(t.pos - c0));
                }
            }
            if (last) break;
        }
    }
    std::printf("deflate blocks %lld, symbols %lld (literals %lld, matches %lld), %.2f bits/symbol, symbols/deflate block %.0f\n", tot_dblocks, tot_sym, tot_lit, tot_match, (double)tot_bits / tot_sym, (double)tot_sym / tot_dblocks);
    std::printf("lit/len codes longer than 11 bits per block: %.1f; symbols decoded through them: %.3f %%\n", (double)tot_long_ll / tot_dblocks, 100.0 * tot_sym_long / tot_sym);
    for (int ci = 0; ci < 5; ++ci) std::printf("chunk %4d bits: %ld chunks, %ld not in step at their end = %.3f %%\n", CHS[ci], nchunks[ci], nfail[ci], 100.0 * nfail[ci] / std::max(1L, nchunks[ci]));
    std::sort(syncdist.begin(), syncdist.end());
    if (!syncdist.empty()) std::printf("bits until in step (of those within 2048): median %ld, p90 %ld, p99 %ld, max %ld\n", syncdist[syncdist.size() / 2], syncdist[syncdist.size() * 9 / 10], syncdist[syncdist.size() * 99 / 100], syncdist.back());
    return 0;
}
