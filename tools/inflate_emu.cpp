// The speculative token pass (squid_amd/csrc/sq_inflate_spec.inc) run on the CPU -- the kernel source itself, its 64 lanes as coroutines
// (sq_wave.h, SQ_WAVE_EMU) -- and compared with zlib: on the BGZF blocks of a BAM file, and on fuzzed streams of every block type.
//   g++ -O1 -g -std=c++17 -DSQ_WAVE_EMU -o build/inflate_emu tools/inflate_emu.cpp -lz
//   build/inflate_emu file.bam [blocks] [cfg]   |   build/inflate_emu --fuzz [cases] [seed]   |   build/inflate_emu --resolve-fuzz [cases] [seed]
// Test infrastructure only.
#define __host__
#define __device__
#include "../squid_amd/csrc/sq_inflate_spec.inc"
#include "../squid_amd/csrc/sq_resolve.inc"

#include <zlib.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

struct Job { std::vector<uint32_t> lds; const uint8_t* payload; uint32_t clen, tcap; std::vector<uint32_t> tok; uint32_t nt[64]; bool err[64]; int cfg; };
template <int CH, int PB>
static void lane_main(void* p) {
    Job& j = *(Job*)p;
    bool err = false;
    const uint32_t nt = isp::inflate_block_spec<CH, PB>(j.lds.data(), j.payload, j.clen, j.tok.data(), j.tcap, err);
    j.nt[wv::lane()] = nt; j.err[wv::lane()] = err;
}
// the staged resolve (sq_resolve.inc, k_lz_resolve5's body) as a wave of the emulator
struct RJob { const uint32_t* tok; int n; uint8_t* out; uint32_t isize; std::vector<uint8_t> st; bool ok[64]; };
template <int STAGE>
static void resolve_lane_main(void* p) {
    RJob& j = *(RJob*)p;
    j.ok[wv::lane()] = rsv::resolve_block<STAGE>(j.tok, j.n, j.out, j.isize, j.st.data());
}
template <int STAGE>
static int emu_resolve(const std::vector<uint32_t>& tok, uint32_t nt, const std::vector<uint8_t>& want) {
    std::vector<uint8_t> out(want.size() + 64, 0xee);
    RJob j{tok.data(), (int)nt, out.data(), (uint32_t)want.size(), std::vector<uint8_t>((size_t)STAGE + 16, 0xdd), {}};
    wv::run_wave(resolve_lane_main<STAGE>, &j);
    for (int l = 1; l < 64; ++l) if (j.ok[l] != j.ok[0]) { std::fprintf(stderr, "staged resolve <%d>: lanes disagree on the result\n", STAGE); return 2; }
    if (!j.ok[0]) { std::fprintf(stderr, "staged resolve <%d>: refuses tokens that the plain resolve takes\n", STAGE); return 2; }
    if (std::memcmp(out.data(), want.data(), want.size()) != 0) { size_t q = 0; while (out[q] == want[q]) ++q; std::fprintf(stderr, "staged resolve <%d>: bytes differ at %zu of %zu\n", STAGE, q, want.size()); return 2; }
    for (size_t k = want.size(); k < out.size(); ++k) if (out[k] != 0xee) { std::fprintf(stderr, "staged resolve <%d>: byte written behind the block\n", STAGE); return 2; }
    return 0;
}
// tokens -> bytes, as k_lz_resolve3 reads them
static bool resolve(const std::vector<uint32_t>& tok, uint32_t nt, std::vector<uint8_t>& out, uint32_t isize) {
    out.clear();
    for (uint32_t i = 0; i < nt; ++i) {
        const uint32_t t = tok[i];
        if (t >> 31) {
            const uint32_t len = (t >> 16) & 0x1ffu, dist = (t & 0x7fffu) + 1u;
            if (dist > out.size() || out.size() + len > isize) return false;
            for (uint32_t k = 0; k < len; ++k) out.push_back(out[out.size() - dist]);
        } else {
            const uint32_t nl = (t >> 24) & 3u, n = nl ? nl : 1u;
            if (out.size() + n > isize) return false;
            for (uint32_t k = 0; k < n; ++k) out.push_back((uint8_t)(t >> (8 * k)));
        }
    }
    return out.size() == isize;
}
// returns 0 same bytes as zlib, 1 error flag raised, 2 WRONG
// SQ_EMU_RSTAT: how many dependent rounds of copies a resolve needs per block when it takes W tokens at a time -- under the rule of
// k_lz_resolve3 (a match may go when its source ends below the output of the first pending match) and under the exact rule (its source
// touches no pending match's output)
struct RStat { unsigned long long round_gt[3] = {0, 0, 0}, rounds64 = 0; unsigned long long dist_le[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mbytes = 0; unsigned long long blocks = 0, tokens = 0, matches = 0, trips_hwm[4] = {0, 0, 0, 0}, trips_exact[4] = {0, 0, 0, 0}, windows[4] = {0, 0, 0, 0}, far[4] = {0, 0, 0, 0}; };
static RStat& rstat() { static RStat r; return r; }
template <class TOK>
static void resolve_stats(const TOK& tok, uint32_t nt) {
    RStat& R = rstat();
    ++R.blocks; R.tokens += nt;
    std::vector<uint32_t> o(nt), len(nt), src(nt); std::vector<char> ism(nt);
    uint32_t at = 0;
    for (uint32_t i = 0; i < nt; ++i) {
        const uint32_t tk = tok[i];
        ism[i] = (char)(tk >> 31);
        const uint32_t nl = (tk >> 24) & 3u;
        len[i] = ism[i] ? (tk >> 16) & 0x1ffu : (nl ? nl : 1u);
        o[i] = at; src[i] = ism[i] ? at - ((tk & 0x7fffu) + 1) : 0; at += len[i];
        if (ism[i]) { ++R.matches; R.mbytes += len[i]; const uint32_t d = (tk & 0x7fffu) + 1; for (int q = 0; q < 8; ++q) if (d <= (128u << q)) ++R.dist_le[q]; }
    }
    for (uint32_t w0 = 0; w0 < nt; w0 += 64) { const uint32_t w1 = std::min(nt, w0 + 64), tot = o[w1 - 1] + len[w1 - 1] - o[w0]; ++R.rounds64; R.round_gt[0] += tot > 496; R.round_gt[1] += tot > 752; R.round_gt[2] += tot > 1008; }
    const int Ws[4] = {64, 128, 256, 512};
    for (int wi = 0; wi < 4; ++wi) {
        const uint32_t W = (uint32_t)Ws[wi];
        for (uint32_t w0 = 0; w0 < nt; w0 += W) {
            const uint32_t w1 = std::min(nt, w0 + W);
            ++R.windows[wi];
            for (int rule = 0; rule < 2; ++rule) {
                std::vector<char> pend(w1 - w0);
                bool any = false;
                for (uint32_t i = w0; i < w1; ++i) { pend[i - w0] = ism[i]; any |= ism[i]; if (rule == 0 && ism[i] && std::min(src[i] + len[i], o[i]) <= o[w0]) ++R.far[wi]; }
                while (any) {
                    ++(rule == 0 ? R.trips_hwm : R.trips_exact)[wi];
                    uint32_t first = w1;
                    for (uint32_t i = w0; i < w1; ++i) if (pend[i - w0]) { first = i; break; }
                    std::vector<uint32_t> go;
                    for (uint32_t i = w0; i < w1; ++i) {
                        if (!pend[i - w0]) continue;
                        const uint32_t need_end = std::min(src[i] + len[i], o[i]);
                        bool ok;
                        if (rule == 0) ok = need_end <= o[first];
                        else { ok = true; for (uint32_t j = w0; j < i && ok; ++j) if (pend[j - w0] && o[j] < need_end && o[j] + len[j] > src[i]) ok = false; }
                        if (ok) go.push_back(i);
                    }
                    for (uint32_t i : go) pend[i - w0] = 0;
                    any = false;
                    for (char c : pend) any |= (bool)c;
                }
            }
        }
    }
}
static int run_block(const uint8_t* payload, uint32_t clen, const std::vector<uint8_t>& want, int cfg, uint32_t* ntok_out = nullptr) {
    Job j;
    j.lds.assign(65536, 0xdeadbeefu);
    j.payload = payload; j.clen = clen; j.cfg = cfg;
    j.tcap = isp::tok_cap_spec((uint32_t)want.size(), clen);
    j.tok.assign(j.tcap + 64, 0x55555555u);
    switch (cfg) {
        case 0: wv::run_wave(lane_main<512, 11>, &j); break;
        case 1: wv::run_wave(lane_main<256, 10>, &j); break;
        case 2: wv::run_wave(lane_main<128, 10>, &j); break;
        case 3: wv::run_wave(lane_main<384, 10>, &j); break;
        case 4: wv::run_wave(lane_main<256, 9>, &j); break;
        case 5: wv::run_wave(lane_main<128, 9>, &j); break;
        default: wv::run_wave(lane_main<1024, 11>, &j); break;
    }
    for (int l = 1; l < 64; ++l) if (j.nt[l] != j.nt[0] || j.err[l] != j.err[0]) { std::fprintf(stderr, "lanes disagree on the result\n"); return 2; }
    for (uint32_t k = j.tcap; k < j.tcap + 64; ++k) if (j.tok[k] != 0x55555555u) { std::fprintf(stderr, "token written behind the block's slots\n"); return 2; }
    if (ntok_out) *ntok_out = j.nt[0];
    if (j.err[0]) return 1;
    if (std::getenv("SQ_EMU_RSTAT")) resolve_stats(j.tok, j.nt[0]);
    std::vector<uint8_t> got;
    if (!resolve(j.tok, j.nt[0], got, (uint32_t)want.size())) { std::fprintf(stderr, "tokens do not resolve to %zu bytes (got %zu)\n", want.size(), got.size()); return 2; }
    if (got != want) { size_t q = 0; while (got[q] == want[q]) ++q; std::fprintf(stderr, "bytes differ at %zu\n", q); return 2; }
    // ... and through the resolve the device runs, with staging areas that nearly every round / hardly any round fits
    if (want.size()) { if (int rc = emu_resolve<496>(j.tok, j.nt[0], want)) return rc; if (int rc = emu_resolve<64>(j.tok, j.nt[0], want)) return rc; if (int rc = emu_resolve<16>(j.tok, j.nt[0], want)) return rc; }
    return 0;
}
static bool zinflate(const uint8_t* p, uint32_t clen, std::vector<uint8_t>& out, uint32_t cap) {
    out.assign(cap, 0);
    z_stream zs; std::memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, -15);
    zs.next_in = (Bytef*)p; zs.avail_in = clen; zs.next_out = out.data(); zs.avail_out = cap;
    const int rc = inflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    inflateEnd(&zs);
    return rc == Z_STREAM_END;
}
static std::vector<uint8_t> zdeflate(const std::vector<uint8_t>& in, int level, int strategy, int memlevel, std::mt19937& rng, bool flushes) {
    z_stream zs; std::memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, memlevel, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, in.size()) + 4096 + in.size() / 8);
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    size_t at = 0;
    while (flushes && at < in.size()) {  // pieces closed by flushes: several deflate blocks, empty stored blocks in between, other parameters
        const size_t n = std::min(in.size() - at, (size_t)(1 + rng() % 20000));
        zs.next_in = (Bytef*)in.data() + at; zs.avail_in = (uInt)n;
        const int fl = rng() % 3 == 0 ? Z_FULL_FLUSH : (rng() % 2 ? Z_SYNC_FLUSH : Z_BLOCK);
        deflate(&zs, fl);
        at += n;
        if (rng() % 3 == 0) deflateParams(&zs, (int)(rng() % 10), (int)(rng() % 5));
    }
    zs.next_in = (Bytef*)in.data() + at; zs.avail_in = (uInt)(in.size() - at);
    deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}
static std::vector<uint8_t> make_data(std::mt19937& rng, size_t n) {
    std::vector<uint8_t> d(n);
    const int kind = (int)(rng() % 8);
    switch (kind) {
        case 0: for (auto& b : d) b = (uint8_t)rng(); break;                                   // incompressible
        case 1: for (auto& b : d) b = (uint8_t)("ACGT"[rng() % 4]); break;                     // four symbols
        case 2: { uint8_t v = 0; for (size_t i = 0; i < n; ++i) { if (rng() % 50 == 0) v = (uint8_t)rng(); d[i] = v; } break; }  // long runs (distance 1, length 258)
        case 3: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i < 300 ? rng() : d[i - 1 - rng() % 300] ^ (rng() % 20 == 0)); break;  // near repeats
        case 4: { std::geometric_distribution<int> g(0.05); for (auto& b : d) b = (uint8_t)std::min(255, g(rng)); break; }  // skewed: long and short codes
        case 5: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i % 7 == 0 ? rng() : 'a' + rng() % 3); break;
        case 6: { std::geometric_distribution<int> g(0.3); for (auto& b : d) b = (uint8_t)(g(rng) * 37); break; }
        default: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i & 0xff); break;             // every byte value, far matches
    }
    return d;
}
int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: inflate_emu file.bam [blocks] | --fuzz [cases] [seed]\n"); return 1; }
    long n_ok = 0, n_flag = 0, n_wrong = 0;
    if (std::string(argv[1]) == "--resolve-fuzz") {
        // random token lists straight into the staged resolve (sq_resolve.inc): matches of every length 3..258 and distance 1..32768 (as far as the bytes in front
        // allow), runs of literals, lists that end inside a round, and damaged lists (a distance beyond the block's start, a length that overruns the block), which
        // must be refused without a byte behind the block
        const long cases = argc > 2 ? std::atol(argv[2]) : 200;
        std::mt19937 rng(argc > 3 ? (unsigned)std::atol(argv[3]) : 4242u);
        long ok = 0, refused = 0, wrong = 0;
        for (long c = 0; c < cases; ++c) {
            const uint32_t target = 1 + rng() % 65536;
            const int flavour = (int)(rng() % 4);  // 0 mixed, 1 long matches, 2 short distances (overlaps), 3 literals mostly
            std::vector<uint32_t> tok;
            std::vector<uint8_t> want;
            while (want.size() < target) {
                const bool match = want.size() >= 1 && (flavour == 3 ? rng() % 8 == 0 : rng() % 2 == 0);
                if (match) {
                    uint32_t len = flavour == 1 ? 200 + rng() % 59 : 3 + rng() % (rng() % 4 == 0 ? 256 : 12);
                    if (len > 258) len = 258;
                    uint32_t dist = flavour == 2 ? 1 + rng() % 9 : 1 + rng() % 32768;
                    if (dist > want.size()) dist = 1 + rng() % (uint32_t)want.size();
                    if (want.size() + len > target) { len = (uint32_t)(target - want.size()); if (len < 3) { for (uint32_t k = 0; k < len; ++k) { tok.push_back((1u << 24) | (rng() & 0xffu)); want.push_back((uint8_t)tok.back()); } continue; } }
                    tok.push_back(0x80000000u | (len << 16) | (dist - 1));
                    for (uint32_t k = 0; k < len; ++k) want.push_back(want[want.size() - dist]);
                } else {
                    uint32_t n = 1 + rng() % 3;
                    if (want.size() + n > target) n = (uint32_t)(target - want.size());
                    uint32_t t = n << 24;
                    for (uint32_t k = 0; k < n; ++k) { const uint8_t b = (uint8_t)rng(); t |= (uint32_t)b << (8 * k); want.push_back(b); }
                    tok.push_back(t);
                }
            }
            const bool damage = c % 5 == 4;
            if (damage && !tok.empty()) {
                const size_t at = rng() % tok.size();
                if (rng() % 2) tok[at] = 0x80000000u | ((3 + rng() % 256) << 16) | 0x7fffu;  // a distance of 32768 (beyond the start unless the block is nearly full there)
                else tok.push_back(0x80000000u | (258u << 16));                                 // one match too many: overruns the block
            }
            tok.resize(tok.size() + 64, 0);
            const uint32_t nt = (uint32_t)tok.size() - 64;
            std::vector<uint8_t> host;
            const bool host_ok = resolve(tok, nt, host, (uint32_t)want.size()) && host == want;
            int rcs[3] = {0, 0, 0};
            if (!damage || host_ok) { rcs[0] = emu_resolve<496>(tok, nt, want); rcs[1] = emu_resolve<64>(tok, nt, want); rcs[2] = emu_resolve<16>(tok, nt, want); if (rcs[0] || rcs[1] || rcs[2]) { ++wrong; std::fprintf(stderr, "case %ld WRONG (flavour %d, %u tokens, %zu bytes)\n", c, flavour, nt, want.size()); } else ++ok; }
            else {
                // a damaged list: every staging size must refuse it and leave the bytes behind the block alone
                bool all_refuse = true;
                const int stages[3] = {496, 64, 16};
                for (int q = 0; q < 3; ++q) {
                    std::vector<uint8_t> out(want.size() + 512, 0xee);
                    RJob j{tok.data(), (int)nt, out.data(), (uint32_t)want.size(), std::vector<uint8_t>((size_t)stages[q] + 16, 0xdd), {}};
                    if (q == 0) wv::run_wave(resolve_lane_main<496>, &j); else if (q == 1) wv::run_wave(resolve_lane_main<64>, &j); else wv::run_wave(resolve_lane_main<16>, &j);
                    if (j.ok[0]) all_refuse = false;
                    for (size_t k = want.size() + 264; k < out.size(); ++k) if (out[k] != 0xee) all_refuse = false;  // (a refused round may have stored its literals and one match's bytes: not further)
                }
                if (all_refuse) ++refused; else { ++wrong; std::fprintf(stderr, "case %ld: a damaged token list was taken (or bytes were written far behind the block)\n", c); }
            }
        }
        std::printf("resolve fuzz: %ld identical to the plain resolve, %ld damaged lists refused, %ld WRONG\n", ok, refused, wrong);
        return wrong ? 2 : 0;
    }
    if (std::string(argv[1]) == "--fuzz") {
        const long cases = argc > 2 ? std::atol(argv[2]) : 200;
        std::mt19937 rng(argc > 3 ? (unsigned)std::atol(argv[3]) : 12345u);
        for (long c = 0; c < cases; ++c) {
            const size_t sizes[6] = {0, 1, 2, 300, 5000, 65280};
            const size_t n = rng() % 3 == 0 ? sizes[rng() % 6] : rng() % 65281;
            const std::vector<uint8_t> data = make_data(rng, n);
            const int level = (int)(rng() % 10), strat = (int)(rng() % 5), mem = 1 + (int)(rng() % 9);
            std::vector<uint8_t> comp = zdeflate(data, level, strat, mem, rng, rng() % 3 == 0);
            if (comp.size() > 65535 + 4096) continue;
            const uint32_t off = rng() % 16;  // the payload anywhere relative to a 16-byte boundary
            std::vector<uint8_t> buf(comp.size() + 64 + 16 + 16);
            uint8_t* base = buf.data() + ((16 - ((uintptr_t)buf.data() & 15)) & 15);
            for (size_t i = 0; i < buf.size() - (size_t)(base - buf.data()); ++i) base[i] = (uint8_t)rng();  // (noise around the payload)
            std::memcpy(base + off, comp.data(), comp.size());
            const int cfg = (int)(rng() % 7);
            const int rc = run_block(base + off, (uint32_t)comp.size(), data, cfg);
            if (rc == 0) ++n_ok; else if (rc == 1) { ++n_flag; std::fprintf(stderr, "case %ld: error flag on a valid stream (n %zu level %d strategy %d cfg %d)\n", c, n, level, strat, cfg); } else { ++n_wrong; std::fprintf(stderr, "case %ld WRONG (n %zu level %d strategy %d mem %d cfg %d off %u)\n", c, n, level, strat, mem, cfg, off); }
        }
        // damaged streams: whatever comes out, no token may land outside the block's slots and the lanes must agree (run_block checks both)
        long n_dam = 0, n_dam_flag = 0;
        for (long c = 0; c < cases / 4; ++c) {
            const std::vector<uint8_t> data = make_data(rng, 1 + rng() % 20000);
            std::vector<uint8_t> comp = zdeflate(data, 6, 0, 8, rng, false);
            for (int k = 0; k < 3; ++k) comp[rng() % comp.size()] ^= (uint8_t)(1u << (rng() % 8));
            std::vector<uint8_t> buf(comp.size() + 96);
            uint8_t* base = buf.data() + ((16 - ((uintptr_t)buf.data() & 15)) & 15);
            std::memcpy(base, comp.data(), comp.size());
            std::vector<uint8_t> want;
            const bool zok = zinflate(base, (uint32_t)comp.size(), want, (uint32_t)data.size());
            std::vector<uint8_t> ref = zok && want.size() == data.size() ? want : data;
            const int rc = run_block(base, (uint32_t)comp.size(), ref, (int)(rng() % 4));
            ++n_dam;
            if (rc == 1) ++n_dam_flag;
            if (rc == 0 && !(zok && want.size() == data.size())) { std::fprintf(stderr, "damaged case %ld: accepted where zlib refuses\n", c); }
        }
        std::printf("fuzz: %ld identical to zlib, %ld flagged, %ld WRONG; damaged streams: %ld run, %ld flagged\n", n_ok, n_flag, n_wrong, n_dam, n_dam_flag);
        return n_wrong || n_flag ? 2 : 0;
    }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    const long maxb = argc > 2 ? std::atol(argv[2]) : 50;
    const int cfg = argc > 3 ? std::atoi(argv[3]) : 0;
    std::vector<uint8_t> raw;
    unsigned long long tot_tok = 0, tot_bytes = 0;
    for (long bi = 0; bi < maxb; ++bi) {
        uint8_t h[18];
        if (std::fread(h, 1, 18, f) != 18) break;
        const int bsize = (h[16] | (h[17] << 8)) + 1;
        raw.assign((size_t)bsize - 18 + 80, 0);
        uint8_t* base = raw.data() + ((16 - ((uintptr_t)raw.data() & 15)) & 15) + (bi % 16);
        if (std::fread(base, 1, (size_t)bsize - 18, f) != (size_t)bsize - 18) break;
        const uint32_t clen = (uint32_t)bsize - 18 - 8;
        uint32_t isize; std::memcpy(&isize, base + clen + 4, 4);
        std::vector<uint8_t> want;
        if (!zinflate(base, clen, want, isize) || want.size() != isize) { std::fprintf(stderr, "block %ld: zlib refuses it\n", bi); return 2; }
        uint32_t nt = 0;
        const int rc = run_block(base, clen, want, cfg, &nt);
        tot_tok += nt; tot_bytes += isize;
        if (rc == 0) ++n_ok; else if (rc == 1) { ++n_flag; std::fprintf(stderr, "block %ld: error flag\n", bi); } else { ++n_wrong; std::fprintf(stderr, "block %ld WRONG\n", bi); }
    }
    std::printf("%ld blocks identical to zlib, %ld flagged, %ld WRONG; %.3f tokens per inflated byte\n", n_ok, n_flag, n_wrong, tot_bytes ? (double)tot_tok / (double)tot_bytes : 0.0);
    if (std::getenv("SQ_EMU_RSTAT")) {
        const RStat& R = rstat();
        std::printf("resolve: %.0f tokens per block, %.0f of them matches\n", (double)R.tokens / R.blocks, (double)R.matches / R.blocks);
        std::printf("  rounds of 64 tokens that write more than 496 / 752 / 1008 bytes: %.2f / %.2f / %.2f %%\n", 100.0 * R.round_gt[0] / R.rounds64, 100.0 * R.round_gt[1] / R.rounds64, 100.0 * R.round_gt[2] / R.rounds64);
        std::printf("  match distances: " ); for (int q = 0; q < 8; ++q) std::printf("<= %u: %.1f %%  ", 128u << q, 100.0 * R.dist_le[q] / std::max<unsigned long long>(1, R.matches)); std::printf("| %.1f bytes per match\n", (double)R.mbytes / std::max<unsigned long long>(1, R.matches));
        const int Ws[4] = {64, 128, 256, 512};
        for (int wi = 0; wi < 4; ++wi) std::printf("  %3d tokens at a time: %.0f windows per block, rounds of copies per block %.0f (rule of k_lz_resolve3) / %.0f (exact), %.1f / %.1f per window; matches whose source lies in front of the window %.0f %%\n", Ws[wi],
            (double)R.windows[wi] / R.blocks, (double)R.trips_hwm[wi] / R.blocks, (double)R.trips_exact[wi] / R.blocks, (double)R.trips_hwm[wi] / R.windows[wi], (double)R.trips_exact[wi] / R.windows[wi], 100.0 * R.far[wi] / std::max<unsigned long long>(1, R.matches));
    }
    { const isp::EmuStat& e = isp::emu_stat();
      if (e.windows) std::printf("wave cost model (SQ_SPEC_PRE %d): %.2f rounds of scans per window; per window %.1f steps in those rounds (sum over the lanes %.0f) + %.1f steps writing tokens\n", SQ_SPEC_PRE,
          (double)e.rounds / e.windows, (double)e.wave_steps_fix / e.windows, (double)e.lane_steps_fix / e.windows, (double)e.wave_steps_emit / e.windows); }
    const isp::EmuStat& st = isp::emu_stat();
    std::printf("writing pass: %llu steps for %llu literals (%llu as pairs: %.1f %% of the literals) and %llu matches: %.2f symbols per step\n", st.iters, st.lits, st.pairs, st.lits ? 200.0 * st.pairs / st.lits : 0.0, st.matches, st.iters ? (double)(st.lits + st.matches) / st.iters : 0.0);
    return n_wrong || n_flag ? 2 : 0;
}
