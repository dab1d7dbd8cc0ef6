// Host-side BGZF/BAM decoder of the product: file -> SoA alignment batches (sq_aln_batch).
// Replaces the BamTools calls of the reference and folds the per-record part of
// ReadRec_t::ReadRec_t (src/ReadRec.cpp:10-88: TotalLen, low-Phred run, CIGAR -> aligned blocks, poly-A/T
// filter, strand-mirrored read offsets) into the decode, so that the sequence/quality bytes never have to
// leave the host.  BGZF blocks are inflated by a small thread pool; records are then walked in file order.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <dlfcn.h>
#include <unistd.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <memory>
#include <mutex>
#include <future>
#include <condition_variable>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "sq_internal.h"

namespace sq {

void HostBatch::clear() {
    refid.clear(); pos.clear(); mrefid.clear(); mpos.clear(); endpos.clear(); b_refpos.clear(); b_matchref.clear();
    flag.clear(); totlen.clear(); b_readpos.clear(); b_matchread.clear(); mapq.clear(); aux.clear();
    blk_off.assign(1, 0); name_off.assign(1, 0); names.clear();
}
void HostBatch::append(const HostBatch& o) {
    auto cat = [](auto& d, const auto& s) { d.insert(d.end(), s.begin(), s.end()); };
    const uint32_t b0 = (uint32_t)b_refpos.size(), n0 = (uint32_t)names.size();
    cat(refid, o.refid); cat(pos, o.pos); cat(mrefid, o.mrefid); cat(mpos, o.mpos); cat(endpos, o.endpos);
    cat(flag, o.flag); cat(totlen, o.totlen); cat(mapq, o.mapq); cat(aux, o.aux);
    cat(b_refpos, o.b_refpos); cat(b_matchref, o.b_matchref); cat(b_readpos, o.b_readpos); cat(b_matchread, o.b_matchread);
    cat(names, o.names);
    for (size_t i = 1; i < o.blk_off.size(); ++i) blk_off.push_back(o.blk_off[i] + b0);
    for (size_t i = 1; i < o.name_off.size(); ++i) name_off.push_back(o.name_off[i] + n0);
}
// the first `count` parts behind one another, every part copied by a thread of its own (a dense sample's chimeric file: millions of
// records per chunk; the one-by-one form above spent more time here than the inflate and the decode together)
void HostBatch::append_parts(const std::vector<HostBatch>& parts, int count) {
    if (count <= 1) { if (count == 1) append(parts[0]); return; }
    struct At { size_t rec, blk, name, boff, noff; };  // (boff / noff: entries of blk_off / name_off; the latter stay at 1 without names)
    std::vector<At> at((size_t)count + 1);
    at[0] = At{refid.size(), b_refpos.size(), names.size(), blk_off.size(), name_off.size()};
    for (int t = 0; t < count; ++t) {
        const HostBatch& o = parts[(size_t)t];
        const At& a = at[(size_t)t];
        at[(size_t)t + 1] = At{a.rec + o.refid.size(), a.blk + o.b_refpos.size(), a.name + o.names.size(), a.boff + o.blk_off.size() - 1, a.noff + o.name_off.size() - 1};
    }
    const At end = at[(size_t)count];
    auto grow = [](auto& v, size_t n) { if (v.capacity() < n) v.reserve(std::max(n, v.capacity() * 2)); v.resize(n); };
    grow(refid, end.rec); grow(pos, end.rec); grow(mrefid, end.rec); grow(mpos, end.rec); grow(endpos, end.rec);
    grow(flag, end.rec); grow(totlen, end.rec); grow(mapq, end.rec); grow(aux, end.rec);
    grow(b_refpos, end.blk); grow(b_matchref, end.blk); grow(b_readpos, end.blk); grow(b_matchread, end.blk);
    grow(names, end.name);
    grow(blk_off, end.boff); grow(name_off, end.noff);
    auto work = [&](int t) {
        const HostBatch& o = parts[(size_t)t];
        const At a = at[(size_t)t];
        auto put = [](auto& d, size_t off, const auto& s) { if (!s.empty()) std::memcpy(d.data() + off, s.data(), s.size() * sizeof(s[0])); };
        put(refid, a.rec, o.refid); put(pos, a.rec, o.pos); put(mrefid, a.rec, o.mrefid); put(mpos, a.rec, o.mpos); put(endpos, a.rec, o.endpos);
        put(flag, a.rec, o.flag); put(totlen, a.rec, o.totlen); put(mapq, a.rec, o.mapq); put(aux, a.rec, o.aux);
        put(b_refpos, a.blk, o.b_refpos); put(b_matchref, a.blk, o.b_matchref); put(b_readpos, a.blk, o.b_readpos); put(b_matchread, a.blk, o.b_matchread);
        put(names, a.name, o.names);
        for (size_t i = 1; i < o.blk_off.size(); ++i) blk_off[a.boff + i - 1] = o.blk_off[i] + (uint32_t)a.blk;
        for (size_t i = 1; i < o.name_off.size(); ++i) name_off[a.noff + i - 1] = o.name_off[i] + (uint32_t)a.name;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < count; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}
void HostBatch::view(sq_aln_batch* b, bool with_names) const {
    std::memset(b, 0, sizeof *b);
    b->n_rec = (int64_t)refid.size();
    b->n_blk = (int64_t)b_refpos.size();
    b->refid = refid.data(); b->pos = pos.data(); b->mate_refid = mrefid.data(); b->mate_pos = mpos.data(); b->end_pos = endpos.data();
    b->flag = flag.data(); b->mapq = mapq.data(); b->aux = aux.data(); b->totlen = totlen.data(); b->blk_off = blk_off.data();
    b->b_refpos = b_refpos.data(); b->b_matchref = b_matchref.data(); b->b_readpos = b_readpos.data(); b->b_matchread = b_matchread.data();
    if (with_names) { b->name_off = name_off.data(); b->name_blob = names.data(); }
}

namespace {

struct ByteView {  // the bytes of a file: a vector's or a mapping's
    const uint8_t* p; size_t n;
    ByteView(const std::vector<uint8_t>& v) : p(v.data()), n(v.size()) {}
    ByteView(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    size_t size() const { return n; }
    const uint8_t& operator[](size_t i) const { return p[i]; }
};

struct FileBytes {
    std::vector<uint8_t> data;
    const uint8_t* map = nullptr;  // a large file read whole: mapped from the page cache instead of copied (a dense sample's chimeric BAM is over a gigabyte)
    size_t map_n = 0;
    ByteView view() const { return map ? ByteView(map, map_n) : ByteView(data); }
    bool load(const char* path, long limit = -1) {
        if (limit < 0) {
            const int fd = ::open(path, O_RDONLY);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= ((size_t)64 << 20)) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
                if (m != MAP_FAILED) { ::close(fd); map = (const uint8_t*)m; map_n = (size_t)st.st_size; return true; }
            }
            ::close(fd);
        }
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (limit >= 0 && n > limit) n = limit;
        data.resize((size_t)n);
        size_t got = n ? std::fread(data.data(), 1, (size_t)n, f) : 0;
        std::fclose(f);
        return got == (size_t)n;
    }
    ~FileBytes() { if (map) munmap((void*)map, map_n); }
};

// growable byte buffer WITHOUT value-initialisation (std::vector::resize would zero-fill hundreds of MB per chunk)
struct RawBuf {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    ~RawBuf() { std::free(p); }
    size_t size() const { return n; }
    uint8_t* data() { return p; }
    uint8_t& operator[](size_t i) { return p[i]; }
    void resize(size_t m) {
        if (m > cap) { size_t c = std::max(m, cap + cap / 2); p = (uint8_t*)std::realloc(p, c); cap = c; }
        n = m;
    }
    void drop_front(size_t k) { if (k) { std::memmove(p, p + k, n - k); n -= k; } }
};

struct BgzfBlock { size_t coff; uint32_t clen, isize; size_t uoff; };
// walk the BGZF container: one entry per block (payload offset/length, inflated size)
bool index_bgzf(const ByteView& d, std::vector<BgzfBlock>& blocks, size_t& total) {
    size_t p = 0;
    total = 0;
    while (p + 18 <= d.size()) {
        if (d[p] != 0x1f || d[p + 1] != 0x8b || !(d[p + 3] & 4)) return false;
        uint32_t xlen = d[p + 10] | (d[p + 11] << 8);
        int bsize = -1;
        for (size_t o = p + 12; o + 4 <= p + 12 + xlen;) {
            uint32_t slen = d[o + 2] | (d[o + 3] << 8);
            if (d[o] == 'B' && d[o + 1] == 'C' && slen == 2) bsize = (d[o + 4] | (d[o + 5] << 8)) + 1;
            o += 4 + slen;
        }
        if (bsize < 0 || p + bsize > d.size()) return false;
        BgzfBlock b;
        b.coff = p + 12 + xlen;
        b.clen = (uint32_t)(bsize - 12 - xlen - 8);
        std::memcpy(&b.isize, &d[p + bsize - 4], 4);
        b.uoff = total;
        total += b.isize;
        blocks.push_back(b);
        p += bsize;
    }
    return p == d.size();
}

// like index_bgzf, but stops quietly at the first incomplete block (for file prefixes)
void index_bgzf_prefix(const ByteView& d, std::vector<BgzfBlock>& blocks, size_t& total) {
    size_t p = 0;
    total = 0;
    while (p + 18 <= d.size()) {
        if (d[p] != 0x1f || d[p + 1] != 0x8b || !(d[p + 3] & 4)) return;
        uint32_t xlen = d[p + 10] | (d[p + 11] << 8);
        int bsize = -1;
        for (size_t o = p + 12; o + 4 <= p + 12 + xlen && o + 6 <= d.size();) {
            uint32_t slen = d[o + 2] | (d[o + 3] << 8);
            if (d[o] == 'B' && d[o + 1] == 'C' && slen == 2) bsize = (d[o + 4] | (d[o + 5] << 8)) + 1;
            o += 4 + slen;
        }
        if (bsize < 0 || p + bsize > d.size()) return;
        BgzfBlock b;
        b.coff = p + 12 + xlen;
        b.clen = (uint32_t)(bsize - 12 - xlen - 8);
        std::memcpy(&b.isize, &d[p + bsize - 4], 4);
        b.uoff = total;
        total += b.isize;
        blocks.push_back(b);
        p += bsize;
    }
}

bool inflate_one(const uint8_t* d, const BgzfBlock& b, uint8_t* out);  // (libdeflate when present, zlib otherwise; below)
bool inflate_range(const ByteView& d, const std::vector<BgzfBlock>& blocks, size_t b0, size_t b1, uint8_t* out, size_t out_base, int n_threads) {
    std::vector<char> ok((size_t)std::max(1, n_threads), 1);
    auto work = [&](int t) {
        for (size_t i = b0 + t; i < b1; i += (size_t)n_threads) {
            const BgzfBlock& b = blocks[i];
            if (!inflate_one(d.p, b, out + (b.uoff - out_base))) { ok[t] = 0; return; }
        }
    };
    if (n_threads <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (char c : ok) if (!c) return false;
    return true;
}

inline int32_t rd32(const uint8_t* p) { int32_t v; std::memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

// aux scan: XA present, IH value (integer typed) -- SegmentGraph.cpp:297-301
bool scan_tags(const uint8_t* p, const uint8_t* e, bool& has_xa, int& ih) {
    has_xa = false;
    bool has_ih = false;
    ih = 0;
    while (p + 3 <= e) {
        uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        const uint8_t* v = p + 3;
        size_t sz;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t* q = v; while (q < e && *q) ++q; if (q >= e) return false; sz = (size_t)(q - v) + 1; break; }
            case 'B': {
                if (v + 5 > e) return false;
                size_t es = (v[0] == 'c' || v[0] == 'C') ? 1 : ((v[0] == 's' || v[0] == 'S') ? 2 : 4);
                uint32_t n; std::memcpy(&n, v + 1, 4);
                sz = 5 + es * n;
                break;
            }
            default: return false;
        }
        if (v + sz > e) return false;
        if (t0 == 'X' && t1 == 'A') has_xa = true;
        if (t0 == 'I' && t1 == 'H' && !has_ih) {
            has_ih = true;
            uint32_t x = 0;
            if (ty == 'c' || ty == 'C' || ty == 'A') x = v[0];
            else if (ty == 's' || ty == 'S') x = v[0] | (v[1] << 8);
            else if (ty == 'i') x = v[0] | (v[1] << 8) | (v[2] << 16) | ((uint32_t)v[3] << 24);  // ('I': BamTools' GetTag<int> refuses UINT32, the value stays 0)
            ih = (int)x;
        }
        p = v + sz;
    }
    return true;
}


// decodes one BAM record (p = first byte after block_size) into a batch; thread-safe (no shared mutable state)
struct RecordDecoder {
    ParseOpts o;
    int thr;
    explicit RecordDecoder(const ParseOpts& o) : o(o), thr((signed char)(((o.phred_type ? 33 : 64) + o.min_phred) & 0xff)) {}
    int decode(const uint8_t* p, int32_t bs, HostBatch& hb, std::string& err) const {
        static const char cigops[] = "MIDNSHP=X???????";
        struct Op { char t; int len; };
        Op cigbuf[64];
        std::vector<Op> cigdyn;
        std::string namebuf;
        const uint8_t* pend = p + bs;
        int32_t refid = rd32(p), pos = rd32(p + 4);
        int lname = p[8];
        int mapq = p[9];
        int ncig = rd16(p + 12);
        int flag = rd16(p + 14);
        int32_t lseq = rd32(p + 16), mrefid = rd32(p + 20), mpos = rd32(p + 24);
        const uint8_t* name = p + 32;
        const uint8_t* cg = name + lname;
        const uint8_t* seq = cg + 4 * (size_t)ncig;
        const uint8_t* qual = seq + (lseq + 1) / 2;
        const uint8_t* aux = qual + lseq;
        if (aux > pend) { err = "corrupt record"; return SQ_E_IO; }
        size_t nlen = lname > 0 ? (size_t)lname - 1 : 0;

        Op* cig = cigbuf;
        if (ncig > 64) { cigdyn.resize(ncig); cig = cigdyn.data(); }
        int totlen = 0, endpos = pos;
        for (int i = 0; i < ncig; ++i) {
            uint32_t v = (uint32_t)rd32(cg + 4 * i);
            cig[i].t = cigops[v & 0xf];
            cig[i].len = (int)(v >> 4);
            char t = cig[i].t;
            if (t == 'M' || t == 'S' || t == 'H' || t == 'I' || t == '=' || t == 'X') totlen += cig[i].len;
            if (t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X') endpos += cig[i].len;  // GetEndPosition()
        }
        // longest run of qualities below the threshold (signed-char compare like the reference)
        int lowrun = 0, run = 0;
        for (int i = 0; i < lseq; ++i) {
            int c = (signed char)((qual[i] + 33) & 0xff);
            run = (c < thr) ? run + 1 : 0;
            if (run > lowrun) lowrun = run;
        }
        bool has_xa = false;
        int ih = 0;
        if (!scan_tags(aux, pend, has_xa, ih)) { err = "corrupt aux data"; return SQ_E_IO; }
        uint8_t ax = 0;
        if (has_xa || ih > 1) ax |= SQ_AUX_MULTI;
        if (lowrun > o.max_lowphred_len) ax |= SQ_AUX_LOWPHRED;
        if (o.inchim) {
            namebuf.assign((const char*)name, nlen);
            if (o.inchim->count(namebuf)) ax |= SQ_AUX_INCHIM;
        }
        // CIGAR -> aligned blocks (ReadRec.cpp:45-87)
        bool rev = flag & 0x10;
        int readpos = 0, refpos = pos, hardclip = 0;
        bool violated = false;
        for (int ic = 0; ic < ncig; ++ic) {
            char t = cig[ic].t;
            if (t == 'S' || t == 'H') {
                readpos += cig[ic].len;
                if (t == 'H') hardclip += cig[ic].len;
            } else if (t == 'M' || t == '=') {
                int tr = 0, tf = 0, ic2;
                for (ic2 = ic; ic2 < ncig && cig[ic2].t != 'S' && cig[ic2].t != 'H' && cig[ic2].t != 'N'; ++ic2) {
                    if (cig[ic2].t != 'D') tr += cig[ic2].len;
                    if (cig[ic2].t != 'I') tf += cig[ic2].len;
                }
                int s0 = readpos - hardclip, s1 = readpos + tr - hardclip;
                if (!(readpos >= hardclip && s1 <= lseq)) { violated = true; break; }  // assert of ReadRec.cpp:64
                int na = 0, nt = 0;
                for (int i = s0; i < s1; ++i) {
                    int code = (seq[i >> 1] >> ((~i & 1) << 2)) & 0xf;
                    na += code == 1;
                    nt += code == 8;
                }
                if (4 * na < 3 * tr && 4 * nt < 3 * tr) {  // 1.0*count/tmpRead < 0.75, exact in integers
                    hb.b_refpos.push_back(refpos);
                    hb.b_matchref.push_back(tf);
                    hb.b_readpos.push_back((uint16_t)(rev ? totlen - readpos - tr : readpos));
                    hb.b_matchread.push_back((uint16_t)tr);
                }
                readpos += tr;
                refpos += tf;
                ic = ic2 - 1;
            } else if (t == 'N')
                refpos += cig[ic].len;
        }
        if (violated) {
            // the reference only constructs a ReadRec_t for records that survive its filters; for those the
            // assert is live (no -DNDEBUG in the Makefile) and the run aborts
            bool filtered = (ax & (SQ_AUX_MULTI | SQ_AUX_INCHIM)) || (flag & 0x400) || (flag & 0x4) || mapq < 0;
            if (!filtered && !o.keep_names) { err = "record without stored bases for an aligned block (reference asserts, ReadRec.cpp:64)"; return SQ_E_ASSERT; }
            if (o.keep_names && !(flag & 0x4) && !(flag & 0x400)) { err = "chimeric record without stored bases (reference asserts, ReadRec.cpp:64)"; return SQ_E_ASSERT; }
            hb.b_refpos.resize(hb.blk_off.back()); hb.b_matchref.resize(hb.blk_off.back());
            hb.b_readpos.resize(hb.blk_off.back()); hb.b_matchread.resize(hb.blk_off.back());
        }
        if (totlen > 65535) { err = "a read longer than 65535 bases (the record layout keeps 16-bit read offsets)"; return SQ_E_CAPACITY; }
        hb.refid.push_back(refid); hb.pos.push_back(pos); hb.mrefid.push_back(mrefid); hb.mpos.push_back(mpos); hb.endpos.push_back(endpos);
        hb.flag.push_back((uint16_t)flag); hb.mapq.push_back((uint8_t)mapq); hb.aux.push_back(ax); hb.totlen.push_back((uint16_t)totlen);
        hb.blk_off.push_back((uint32_t)hb.b_refpos.size());
        if (o.keep_names) {
            hb.names.insert(hb.names.end(), (const char*)name, (const char*)name + nlen);
            hb.name_off.push_back((uint32_t)hb.names.size());
        }
        return SQ_OK;
    }
};

}  // namespace

static int read_bam_header_bytes(const std::vector<uint8_t>& d, bool whole_file, std::vector<std::string>& names, std::vector<int32_t>& lens, std::string& err);

int read_bam_header(const char* path, std::vector<std::string>& names, std::vector<int32_t>& lens, std::string& err) {
    // the header sits in the leading BGZF blocks: read a prefix and grow it only if the header text is longer
    for (long limit = 1 << 20;; limit *= 8) {
        FileBytes probe;
        if (!probe.load(path, limit)) { err = std::string("cannot open bamfile ") + path; return SQ_E_IO; }
        int rc = read_bam_header_bytes(probe.data, (long)probe.data.size() < limit, names, lens, err);
        if (rc != 1) return rc;
    }
}

// returns 1 when `d` (a prefix of the file) ends before the header does
static int read_bam_header_bytes(const std::vector<uint8_t>& d, bool whole_file, std::vector<std::string>& names, std::vector<int32_t>& lens, std::string& err) {
    FileBytes fb;
    fb.data = d;
    std::vector<BgzfBlock> blocks;
    size_t total;
    index_bgzf_prefix(fb.data, blocks, total);
    if (blocks.empty()) { if (whole_file) { err = "not a BGZF file"; return SQ_E_IO; } return 1; }
    // the header may span several blocks: inflate until it is complete
    std::vector<uint8_t> u;
    size_t nb = 0;
    auto need = [&](size_t n) {
        while (u.size() < n && nb < blocks.size()) {
            size_t old = u.size();
            u.resize(old + blocks[nb].isize);
            BgzfBlock b = blocks[nb];
            std::vector<BgzfBlock> one(1, b);
            one[0].uoff = 0;
            if (!inflate_range(fb.data, one, 0, 1, u.data() + old, 0, 1)) return false;
            ++nb;
        }
        return u.size() >= n;
    };
    if (!need(12)) { if (whole_file) { err = "not a BAM file"; return SQ_E_IO; } return 1; }
    if (std::memcmp(u.data(), "BAM\1", 4) != 0) { err = "not a BAM file"; return SQ_E_IO; }
    int32_t ltext = rd32(&u[4]);
    if (!need(12 + (size_t)ltext)) { if (whole_file) { err = "truncated header"; return SQ_E_IO; } return 1; }
    // names and lengths come from the header TEXT, as in ReadRec.cpp:274-279 (stoi on LN)
    std::string text((const char*)&u[8], (size_t)ltext);
    names.clear();
    lens.clear();
    size_t i = 0;
    while (i < text.size()) {
        size_t e = text.find('\n', i);
        if (e == std::string::npos) e = text.size();
        if (text.compare(i, 3, "@SQ") == 0) {
            std::string sn, ln;
            size_t f = i;
            while (f < e) {
                size_t t = text.find('\t', f);
                if (t == std::string::npos || t > e) t = e;
                if (text.compare(f, 3, "SN:") == 0) sn = text.substr(f + 3, t - f - 3);
                if (text.compare(f, 3, "LN:") == 0) ln = text.substr(f + 3, t - f - 3);
                f = t + 1;
            }
            names.push_back(sn);
            lens.push_back(std::atoi(ln.c_str()));
        }
        i = e + 1;
    }
    return SQ_OK;
}

typedef std::function<int(const uint8_t*, size_t, const unsigned long long*, int64_t)> RawSink;
// Scratch of a whole-file read (the chimeric BAM as one batch), kept by the process between reads: the inflated rounds and the decoded
// batch.  A dense sample's chimeric BAM is 2.9 GB inflated and 0.6 GB decoded; as fresh allocations every read pays for them in page faults
// (measured on the box: inflate 270 ms on 64 threads, the final concatenation 170 ms -- both mostly first touches).  One read at a time
// uses the cache (a second concurrent one allocates its own); sq_release_reader_buffers gives the memory back (it holds no file content that a
// later read would trust: every read fills it again).
namespace {
struct WholeFileScratch { std::mutex mu; bool busy = false; RawBuf u; HostBatch hb; };
WholeFileScratch g_whole_file;
}
void drop_whole_file_scratch() {
    std::lock_guard<std::mutex> lk(g_whole_file.mu);
    if (g_whole_file.busy) return;
    std::free(g_whole_file.u.p); g_whole_file.u.p = nullptr; g_whole_file.u.n = g_whole_file.u.cap = 0;
    g_whole_file.hb = HostBatch();
}
static int parse_bam_impl(const char* path, const ParseOpts& o, size_t batch_records, int n_threads, std::string& err,
                          const std::function<int(const HostBatch&)>& sink, const RawSink* raw_sink) {
    using clk = std::chrono::steady_clock;
    double t_read = 0, t_inflate = 0, t_walk = 0, t_decode = 0, t_append = 0, t_sink = 0;
    auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    struct Report { double &a, &b, &c, &d, &e, &f; const char* path; ~Report() { if (std::getenv("SQUID_INGEST_TIMING")) std::fprintf(stderr, "ingest %s: read %.1f inflate %.1f walk %.1f decode %.1f append %.1f sink %.1f ms\n", path, a, b, c, d, e, f); } } report{t_read, t_inflate, t_walk, t_decode, t_append, t_sink, path};
    auto tr0 = clk::now();
    FileBytes fb;
    if (!fb.load(path)) { err = std::string("cannot open bamfile ") + path; return SQ_E_IO; }
    std::vector<BgzfBlock> blocks;
    size_t total;
    const ByteView file = fb.view();
    if (!index_bgzf(file, blocks, total)) { err = "not a BGZF file"; return SQ_E_IO; }
    t_read = since(tr0);
    n_threads = std::max(1, n_threads);

    // <= 128 MiB inflated per round; the whole-file mode (the chimeric BAM: one batch, millions of records on the dense config) takes 512 MiB
    // rounds, inflates them on as many threads as the machine gives (the decode stays at n_threads: more of those were slower, DESIGN.md
    // section 5) and strings the decoded parts together ONCE at the end -- round 4 grew sixteen vectors round by round, 0.18 s of a 0.72 s read
    const bool whole_file = batch_records >= ((size_t)1 << 32) && !raw_sink;
    const size_t kChunkBlocks = whole_file ? 8192 : 2048;
    const int inflate_threads = whole_file ? std::max(n_threads, std::min(64, usable_cpus() / (4 * local_process_count(1)))) : n_threads;  // (its share of the host: the processes side by side divide the CPUs)
    std::vector<HostBatch> kept;  // whole-file mode: the decoded parts of all rounds, in stream order
    bool cached = false;
    if (whole_file) { std::lock_guard<std::mutex> lk(g_whole_file.mu); if (!g_whole_file.busy) { g_whole_file.busy = true; cached = true; } }
    struct Unbusy { bool on; ~Unbusy() { if (on) { std::lock_guard<std::mutex> lk(g_whole_file.mu); g_whole_file.busy = false; } } } unbusy{cached};
    RawBuf u_local;
    HostBatch hb_local;
    RawBuf& u = cached ? g_whole_file.u : u_local;  // inflated bytes not yet consumed
    u.n = 0;
    size_t nb = 0;
    size_t consumed = 0;
    auto refill = [&]() -> bool {
        if (nb >= blocks.size()) return false;
        if (consumed) { u.drop_front(consumed); consumed = 0; }
        size_t b1 = std::min(blocks.size(), nb + kChunkBlocks);
        size_t base = blocks[nb].uoff, bytes = (b1 == blocks.size() ? total : blocks[b1].uoff) - base;
        size_t old = u.size();
        u.resize(old + bytes);
        auto ti0 = clk::now();
        bool ok = inflate_range(file, blocks, nb, b1, u.data() + old, base, inflate_threads);
        t_inflate += since(ti0);
        nb = b1;
        return ok;
    };
    auto need = [&](size_t n) -> bool {
        while (u.size() - consumed < n) if (!refill()) return u.size() - consumed >= n;
        return true;
    };
    // ---- header
    if (!need(12) || std::memcmp(&u[consumed], "BAM\1", 4) != 0) { err = "not a BAM file"; return SQ_E_IO; }
    int32_t ltext = rd32(&u[consumed + 4]);
    if (!need(12 + (size_t)ltext)) { err = "truncated header"; return SQ_E_IO; }
    int32_t nref = rd32(&u[consumed + 8 + ltext]);
    consumed += 12 + (size_t)ltext;
    for (int i = 0; i < nref; ++i) {
        if (!need(4)) { err = "truncated header"; return SQ_E_IO; }
        int32_t ln = rd32(&u[consumed]);
        if (!need(8 + (size_t)ln)) { err = "truncated header"; return SQ_E_IO; }
        consumed += 8 + (size_t)ln;
    }
    // ---- records.  The chain of block_size fields is inherently serial, and walking it through memory that other
    // cores have just written costs a cache miss per record.  Each decoder thread therefore takes a byte slice of the
    // inflated chunk, finds the first record boundary inside it by validating a chain of plausible record headers,
    // and walks + decodes from there; afterwards the slices are stitched: slice t must end exactly where slice t+1
    // started, otherwise (a false synchronisation -- never seen in practice) the tail of the chunk is redone serially.
    HostBatch& hb = cached ? g_whole_file.hb : hb_local;
    hb.clear();
    RecordDecoder dec(o);
    std::vector<HostBatch> parts((size_t)n_threads);
    std::vector<int> prc((size_t)n_threads);
    std::vector<std::string> perr((size_t)n_threads);
    std::vector<size_t> s_begin((size_t)n_threads), s_end((size_t)n_threads);
    std::vector<std::vector<unsigned long long>> roffs((size_t)n_threads);  // raw mode: record offsets per slice
    auto plausible = [&](size_t p, size_t limit) -> long {  // record header at p?  returns its block_size or -1
        if (limit - p < 36) return -1;
        const uint8_t* q = &u[p];
        int32_t bs = rd32(q);
        if (bs < 34 || bs > (1 << 26)) return -1;
        int32_t refid = rd32(q + 4), pos = rd32(q + 8), mrefid = rd32(q + 24), mpos = rd32(q + 28), lseq = rd32(q + 20);
        int lname = q[12], ncig = rd16(q + 16);
        if (refid < -1 || refid >= nref || mrefid < -1 || mrefid >= nref || pos < -1 || mpos < -1 || lseq < 0 || lname < 1) return -1;
        size_t need_bytes = 32 + (size_t)lname + 4 * (size_t)ncig + ((size_t)lseq + 1) / 2 + (size_t)lseq;
        if (need_bytes > (size_t)bs) return -1;
        if (p + 4 + 32 + (size_t)lname <= limit && q[4 + 32 + lname - 1] != 0) return -1;  // QNAME is NUL terminated
        return bs;
    };
    auto sync_from = [&](size_t from, size_t upto, size_t limit) -> size_t {  // first offset in [from,upto) that starts a chain of 4 plausible records
        for (size_t p = from; p < upto; ++p) {
            size_t q = p;
            int ok = 0;
            bool accept = false;
            for (;;) {
                long bs = plausible(q, limit);
                if (bs < 0) break;
                if (q + 4 + (size_t)bs > limit) { accept = ok >= 2; break; }  // runs off the buffer: trust it only after two whole records
                q += 4 + (size_t)bs;
                if (++ok == 4 || q == limit) { accept = true; break; }
            }
            if (accept) ok = 4;
            if (ok == 4) return p;
        }
        return upto;
    };
    for (;;) {
        const size_t limit = u.size();
        size_t avail = limit - consumed;
        if (avail >= 4) {
            auto td0 = clk::now();
            const int T = (int)std::min<size_t>((size_t)n_threads, std::max<size_t>(1, avail / ((size_t)1 << 20)));
            auto work = [&](int t) {
                HostBatch& part = parts[t];
                part.clear();
                roffs[t].clear();
                prc[t] = SQ_OK;
                size_t lo = consumed + avail * t / T, hi = consumed + avail * (t + 1) / T;
                size_t p = t == 0 ? consumed : sync_from(lo, hi, limit);
                s_begin[t] = p;
                while (p < hi) {
                    if (limit - p < 4) break;
                    int32_t bs = rd32(&u[p]);
                    if (bs < 32) { prc[t] = SQ_E_IO; perr[t] = "truncated record"; break; }
                    if (limit - p < 4 + (size_t)bs) break;  // incomplete: belongs to the next chunk
                    if (raw_sink) roffs[t].push_back((unsigned long long)(p - consumed));
                    else {
                        int rc = dec.decode(&u[p + 4], bs, part, perr[t]);
                        if (rc) { prc[t] = rc; break; }
                    }
                    p += 4 + (size_t)bs;
                }
                s_end[t] = p;
            };
            if (T == 1) work(0);
            else {
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t) th.emplace_back(work, t);
                for (auto& x : th) x.join();
            }
            t_decode += since(td0);
            // stitch
            size_t good_end = s_end[0];
            int good = 1;
            if (prc[0]) { err = perr[0]; return prc[0]; }
            for (int t = 1; t < T; ++t) {
                size_t hi = consumed + avail * (t + 1) / T;
                if (s_begin[t] >= hi && parts[t].size() == 0 && roffs[t].empty() && good_end >= hi) { ++good; continue; }  // slice swallowed by a long record
                if (s_begin[t] != good_end) { if (std::getenv("SQUID_INGEST_DEBUG")) std::fprintf(stderr, "stitch mismatch at slice %d/%d: begin=%zu expected=%zu lo=%zu hi=%zu\n", t, T, s_begin[t], good_end, consumed + avail * t / T, hi); break; }
                if (prc[t]) { err = perr[t]; return prc[t]; }
                good_end = s_end[t];
                ++good;
            }
            if (raw_sink) {
                auto ta0 = clk::now();
                std::vector<unsigned long long> offs;
                for (int t = 0; t < good; ++t) offs.insert(offs.end(), roffs[t].begin(), roffs[t].end());
                size_t p = good_end;
                if (good < T)  // false synchronisation: finish the chunk with the plain serial walk
                    for (;;) {
                        if (limit - p < 4) break;
                        int32_t bs = rd32(&u[p]);
                        if (bs < 32) { err = "truncated record"; return SQ_E_IO; }
                        if (limit - p < 4 + (size_t)bs) break;
                        offs.push_back((unsigned long long)(p - consumed));
                        p += 4 + (size_t)bs;
                    }
                t_append += since(ta0);
                auto ts0 = clk::now();
                int rc = (*raw_sink)(&u[consumed], p - consumed, offs.data(), (int64_t)offs.size());
                t_sink += since(ts0);
                if (rc) return rc;
                consumed = p;
                if (!refill()) break;
                continue;
            }
            if (whole_file) {  // one batch for the whole file (the chimeric BAM): the parts are kept and strung together at the end
                for (int t = 0; t < good; ++t) { kept.push_back(std::move(parts[(size_t)t])); parts[(size_t)t] = HostBatch(); parts[(size_t)t].clear(); }
            } else
            for (int t = 0; t < good; ++t) {
                auto ta0 = clk::now();
                hb.append(parts[t]);
                t_append += since(ta0);
                if (hb.size() >= batch_records) {
                    auto ts0 = clk::now();
                    int rc = sink(hb);
                    t_sink += since(ts0);
                    if (rc) return rc;
                    hb.clear();
                }
            }
            if (good < T) {  // false synchronisation somewhere: finish this chunk with the plain serial walk
                auto tw0 = clk::now();
                size_t p = good_end;
                HostBatch& part = parts[0];
                part.clear();
                for (;;) {
                    if (limit - p < 4) break;
                    int32_t bs = rd32(&u[p]);
                    if (bs < 32) { err = "truncated record"; return SQ_E_IO; }
                    if (limit - p < 4 + (size_t)bs) break;
                    int rc = dec.decode(&u[p + 4], bs, part, err);
                    if (rc) return rc;
                    p += 4 + (size_t)bs;
                }
                if (whole_file) { kept.push_back(std::move(part)); parts[0] = HostBatch(); parts[0].clear(); }
                else hb.append(part);
                good_end = p;
                t_walk += since(tw0);
            }
            consumed = good_end;
        }
        if (!refill()) break;
    }
    if (u.size() - consumed >= 4) { err = "truncated record"; return SQ_E_IO; }
    if (whole_file && !kept.empty()) {
        auto ta0 = clk::now();
        hb.append_parts(kept, (int)kept.size());
        std::vector<HostBatch>().swap(kept);
        t_append += since(ta0);
    }
    if (hb.size()) {
        auto ts0 = clk::now();
        int rc = sink(hb);
        t_sink += since(ts0);
        if (rc) return rc;
    }
    return SQ_OK;
}

int parse_bam_file(const char* path, const ParseOpts& o, size_t batch_records, int n_threads, std::string& err, const std::function<int(const HostBatch&)>& sink) {
    return parse_bam_impl(path, o, batch_records, n_threads, err, sink, nullptr);
}
// ---------------------------------------------------------------------------------------------- raw (K0) reader
// File -> inflated record stream for the GPU parser, as a pipeline: the file is memory-mapped (no read copy), a pool of
// threads inflates the next 64 MiB of BGZF blocks and finds the record boundaries in it while the previous chunk is
// being copied to the GPU and parsed there (the sink runs on its own thread, one chunk in flight, two buffers).
namespace {
struct FileMap {
    const uint8_t* p = nullptr;
    size_t n = 0;
    bool open(const char* path, bool populate = true) {
        int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        n = (size_t)st.st_size;
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);  // (populate: the block index walks every page anyway, one fault at a time)
            if (m == MAP_FAILED) { ::close(fd); return false; }
            madvise(m, n, MADV_SEQUENTIAL);
            p = (const uint8_t*)m;
        }
        ::close(fd);
        return true;
    }
    ~FileMap() { if (p) munmap((void*)p, n); }
};
// The mapping of the last file read stays alive until another file is read: unmapping gigabytes takes ~150 ms during which
// every page fault and allocation of the process queues behind the address-space lock (measured: the graph build that
// follows an ingest took 114 ms instead of 14), and a second read of the same file (parameter sweeps, benchmark steps)
// finds its page tables filled.  Identity = path + size + mtime + inode.
struct MapCache {
    std::mutex mu;
    std::string path; off_t size = 0; time_t mtime = 0; long mtime_ns = 0; ino_t ino = 0;
    std::shared_ptr<FileMap> map;
    bool populated = false;
    // the BGZF block index of the mapped file, kept once a reader has walked the whole file: a later read of the same file
    // (benchmark steps, parameter sweeps without the record cache) plans its batches without touching the file again -- the walk
    // reads one header per ~20 KB of a 6 GB mapping, ~4 ms per batch, and stands between two batches
    std::shared_ptr<const std::vector<BgzfBlock>> index;
    size_t index_total = 0;
    std::shared_ptr<const std::vector<BgzfBlock>> get_index(const std::shared_ptr<FileMap>& of, size_t& total) {
        std::lock_guard<std::mutex> lk(mu);
        if (of && of == map && index) { total = index_total; return index; }
        return nullptr;
    }
    void set_index(const std::shared_ptr<FileMap>& of, const std::vector<BgzfBlock>& blocks, size_t total) {
        std::lock_guard<std::mutex> lk(mu);
        if (of && of == map && !index) { index = std::make_shared<const std::vector<BgzfBlock>>(blocks); index_total = total; }
    }
    std::shared_ptr<FileMap> acquire(const char* pth, bool populate, bool& reused) {
        struct stat st;
        reused = false;
        if (::stat(pth, &st) != 0) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        if (map && path == pth && size == st.st_size && mtime == st.st_mtim.tv_sec && mtime_ns == st.st_mtim.tv_nsec && ino == st.st_ino) { reused = true; return map; }
        if (map) {  // another file: the old mapping goes away off the caller's path
            index.reset(); index_total = 0;
            std::shared_ptr<FileMap> old;
            old.swap(map);
            std::thread([old]() mutable { old.reset(); }).detach();
        }
        auto m = std::make_shared<FileMap>();
        if (!m->open(pth, populate)) return nullptr;
        map = m; path = pth; size = st.st_size; mtime = st.st_mtim.tv_sec; mtime_ns = st.st_mtim.tv_nsec; ino = st.st_ino;
        return m;
    }
};
// Two files are remembered: a sample is a pair of BAM files, and since round 5 both go through this reader (sq_ingest_files sends the
// chimeric BAM of a dense sample through the GPU reader too) -- with one slot every read of the one file threw the mapping and the block
// index of the other away.  A third file takes the slot that was used longest ago.
static MapCache g_map_slots[2];
static std::mutex g_map_pick_mu;
static unsigned long g_map_tick = 0, g_map_used[2] = {0, 0};
static MapCache& map_cache_for(const char* pth) {
    std::lock_guard<std::mutex> lk(g_map_pick_mu);
    int k = -1;
    for (int i = 0; i < 2; ++i) { std::lock_guard<std::mutex> l2(g_map_slots[i].mu); if (g_map_slots[i].map && g_map_slots[i].path == pth) k = i; }
    if (k < 0) for (int i = 0; i < 2; ++i) { std::lock_guard<std::mutex> l2(g_map_slots[i].mu); if (!g_map_slots[i].map) { k = i; break; } }
    if (k < 0) k = g_map_used[0] <= g_map_used[1] ? 0 : 1;
    g_map_used[k] = ++g_map_tick;
    return g_map_slots[k];
}
// `count` jobs on the calling thread + helpers; the helpers live as long as the pool (no spawn per chunk)
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    const std::function<void(int)>* job = nullptr;
    int job_count = 0, next = 0, running = 0, epoch = 0;
    bool stop = false;
    explicit Pool(int helpers) {
        for (int i = 0; i < helpers; ++i) th.emplace_back([this]() {
            int seen = 0;
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [&]() { return stop || epoch != seen; });
                if (stop) return;
                seen = epoch;
                while (next < job_count) { int i = next++; ++running; lk.unlock(); (*job)(i); lk.lock(); --running; }
                if (running == 0) done_cv.notify_all();
            }
        });
    }
    void run(int count, const std::function<void(int)>& f) {
        std::unique_lock<std::mutex> lk(mu);
        job = &f; job_count = count; next = 0; ++epoch;
        cv.notify_all();
        while (next < job_count) { int i = next++; ++running; lk.unlock(); f(i); lk.lock(); --running; }
        done_cv.wait(lk, [&]() { return running == 0 && next >= job_count; });
        job = nullptr;
    }
    ~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
};
// walks the BGZF container of a mapped file, a piece at a time
struct BgzfIndexer {
    const uint8_t* d; size_t n, p = 0, total = 0; bool bad = false;
    size_t stop = (size_t)-1;  // the last block to take starts at this file offset (a shard's range)
    bool at_end() const { return bad || p + 18 > n; }
    bool complete() const { return !bad && (p == n || p > stop); }
    bool more(std::vector<BgzfBlock>& blocks, size_t max_new) {  // false: nothing was added
        size_t added = 0;
        while (added < max_new && p + 18 <= n && p <= stop) {
            if (d[p] != 0x1f || d[p + 1] != 0x8b || !(d[p + 3] & 4)) { bad = true; break; }
            uint32_t xlen = d[p + 10] | (d[p + 11] << 8);
            int bsize = -1;
            for (size_t o = p + 12; o + 4 <= p + 12 + xlen && o + 6 <= n;) {
                uint32_t slen = d[o + 2] | (d[o + 3] << 8);
                if (d[o] == 'B' && d[o + 1] == 'C' && slen == 2) bsize = (d[o + 4] | (d[o + 5] << 8)) + 1;
                o += 4 + slen;
            }
            if (bsize < 0 || p + bsize > n) { bad = true; break; }
            BgzfBlock b;
            b.coff = p + 12 + xlen;
            b.clen = (uint32_t)(bsize - 12 - xlen - 8);
            std::memcpy(&b.isize, &d[p + bsize - 4], 4);
            b.uoff = total;
            total += b.isize;
            blocks.push_back(b);
            p += bsize;
            ++added;
        }
        return added > 0;
    }
};
bool index_bgzf_view(const uint8_t* d, size_t n, std::vector<BgzfBlock>& blocks, size_t& total) {
    BgzfIndexer ix{d, n};
    while (ix.more(blocks, (size_t)1 << 20)) {}
    total = ix.total;
    return ix.complete();
}
// ---- BAI (SAM spec 5.2): what a chromosome shard needs from it is, per reference, the virtual file offset of its first record
// (coffset << 16 | offset inside the inflated block) -- from the metadata pseudo-bin 37450 when the writer left one (samtools,
// the generator), else the smallest chunk start over its bins.  `<bam>.bai` or `<bam minus extension>.bai`.
struct BaiIndex {
    std::vector<unsigned long long> ref_beg;  // ~0ull: the reference has no record
    bool load(const std::string& bam_path, int n_ref_expected) {
        std::string cand[2] = {bam_path + ".bai", bam_path.size() > 4 ? bam_path.substr(0, bam_path.size() - 4) + ".bai" : std::string()};
        for (const std::string& pth : cand) {
            if (pth.empty()) continue;
            FILE* f = std::fopen(pth.c_str(), "rb");
            if (!f) continue;
            std::vector<uint8_t> d;
            uint8_t buf[65536];
            for (size_t k; (k = std::fread(buf, 1, sizeof buf, f)) > 0;) d.insert(d.end(), buf, buf + k);
            std::fclose(f);
            if (parse(d, n_ref_expected)) return true;
        }
        return false;
    }
    bool parse(const std::vector<uint8_t>& d, int n_ref_expected) {
        size_t p = 0;
        auto u32 = [&](uint32_t& v) { if (p + 4 > d.size()) return false; std::memcpy(&v, &d[p], 4); p += 4; return true; };
        auto u64 = [&](unsigned long long& v) { if (p + 8 > d.size()) return false; std::memcpy(&v, &d[p], 8); p += 8; return true; };
        if (d.size() < 8 || std::memcmp(d.data(), "BAI\1", 4) != 0) return false;
        p = 4;
        uint32_t nref = 0;
        if (!u32(nref) || (n_ref_expected >= 0 && (int)nref != n_ref_expected)) return false;
        ref_beg.assign(nref, ~0ull);
        for (uint32_t r = 0; r < nref; ++r) {
            uint32_t nbin = 0;
            if (!u32(nbin)) return false;
            unsigned long long best = ~0ull, meta = ~0ull;
            for (uint32_t b = 0; b < nbin; ++b) {
                uint32_t bin = 0, nchunk = 0;
                if (!u32(bin) || !u32(nchunk)) return false;
                for (uint32_t k = 0; k < nchunk; ++k) {
                    unsigned long long beg = 0, end = 0;
                    if (!u64(beg) || !u64(end)) return false;
                    if (bin == 37450) { if (k == 0) meta = beg; }
                    else if (beg < best) best = beg;
                }
            }
            uint32_t nintv = 0;
            if (!u32(nintv)) return false;
            if (p + 8ull * nintv > d.size()) return false;
            p += 8ull * nintv;
            ref_beg[r] = meta != ~0ull ? meta : best;
        }
        return true;
    }
};

// raw-DEFLATE decoder of libdeflate (2-3x faster than zlib's inflate), bound at run time when the shared library is on
// the system (no header needed: three functions of its stable C API); zlib otherwise
struct FastInflate {
    typedef void* (*alloc_fn)();
    typedef int (*run_fn)(void*, const void*, size_t, void*, size_t, size_t*);
    typedef void (*free_fn)(void*);
    alloc_fn alloc = nullptr; run_fn run = nullptr; free_fn release = nullptr;
    FastInflate() {
        if (std::getenv("SQUID_ZLIB_ONLY")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
        run = (run_fn)dlsym(h, "libdeflate_deflate_decompress");
        release = (free_fn)dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !run || !release) { alloc = nullptr; run = nullptr; release = nullptr; }
    }
};
const FastInflate& fast_inflate() { static FastInflate f; return f; }
struct ThreadDecompressor { void* d = nullptr; ~ThreadDecompressor() { if (d) fast_inflate().release(d); } };
bool inflate_one(const uint8_t* d, const BgzfBlock& b, uint8_t* out) {
    if (!b.isize) return true;
    const FastInflate& fi = fast_inflate();
    if (fi.run) {
        static thread_local ThreadDecompressor td;
        if (!td.d) td.d = fi.alloc();
        size_t got = 0;
        if (td.d && fi.run(td.d, d + b.coff, b.clen, out, b.isize, &got) == 0 && got == b.isize) return true;
        // (fall through: let zlib have a look before calling the block corrupt)
    }
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = (Bytef*)(d + b.coff);
    zs.avail_in = b.clen;
    zs.next_out = out;
    zs.avail_out = b.isize;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END;
}
// record boundaries in u[begin, limit): slices find their first boundary by validating a chain of plausible record
// headers and walk from there; the slices are stitched, and a false synchronisation (never seen) is redone serially.
// offs are relative to `begin`; `end` = end of the last whole record.
int find_records(const uint8_t* u, size_t begin, size_t limit, int nref, Pool& pool, int n_threads, std::vector<unsigned long long>& offs, size_t& end, std::string& err,
                 bool begin_unsynced = false, size_t* first_record = nullptr) {
    auto plausible = [&](size_t p) -> long {
        if (limit - p < 36) return -1;
        const uint8_t* q = u + p;
        int32_t bs = rd32(q);
        if (bs < 34 || bs > (1 << 26)) return -1;
        int32_t refid = rd32(q + 4), pos = rd32(q + 8), mrefid = rd32(q + 24), mpos = rd32(q + 28), lseq = rd32(q + 20);
        int lname = q[12], ncig = rd16(q + 16);
        if (refid < -1 || refid >= nref || mrefid < -1 || mrefid >= nref || pos < -1 || mpos < -1 || lseq < 0 || lname < 1) return -1;
        size_t need_bytes = 32 + (size_t)lname + 4 * (size_t)ncig + ((size_t)lseq + 1) / 2 + (size_t)lseq;
        if (need_bytes > (size_t)bs) return -1;
        if (p + 4 + 32 + (size_t)lname <= limit && q[4 + 32 + lname - 1] != 0) return -1;  // QNAME is NUL terminated
        // the optional fields must tile the rest of the record exactly: a header read one or two bytes early can pass all of
        // the above and even run into true records (seen: 1 slice in 8000), but its "fields" are other records
        if (p + 4 + (size_t)bs <= limit) {
            const uint8_t *a = q + 4 + need_bytes, *e = q + 4 + bs;
            while (a < e) {
                if (e - a < 3) return -1;
                if (!std::isalpha(a[0]) || !std::isalnum(a[1])) return -1;
                const uint8_t ty = a[2];
                const uint8_t* v = a + 3;
                size_t sz;
                if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') { const uint8_t* z = v; while (z < e && *z) ++z; if (z >= e) return -1; sz = (size_t)(z - v) + 1; }
                else if (ty == 'B') {
                    if (e - v < 5) return -1;
                    const size_t es = (v[0] == 'c' || v[0] == 'C') ? 1 : ((v[0] == 's' || v[0] == 'S') ? 2 : ((v[0] == 'i' || v[0] == 'I' || v[0] == 'f') ? 4 : 0));
                    if (!es) return -1;
                    sz = 5 + es * (size_t)(uint32_t)rd32(v + 1);
                } else return -1;
                if (sz > (size_t)(e - v)) return -1;
                a = v + sz;
            }
        }
        return bs;
    };
    auto sync_from = [&](size_t from, size_t upto) -> size_t {
        for (size_t p = from; p < upto; ++p) {
            size_t q = p;
            int ok = 0;
            bool accept = false;
            for (;;) {
                long bs = plausible(q);
                if (bs < 0) break;
                if (q + 4 + (size_t)bs > limit) { accept = ok >= 2; break; }  // runs off the buffer: trust it only after two whole records
                q += 4 + (size_t)bs;
                if (++ok == 4 || q == limit) { accept = true; break; }
            }
            if (accept) return p;
        }
        return upto;
    };
    offs.clear();
    end = begin;
    const size_t avail = limit - begin;
    if (avail < 4) return SQ_OK;
    const int T = (int)std::min<size_t>((size_t)std::max(1, n_threads), std::max<size_t>(1, avail / ((size_t)1 << 20)));
    std::vector<std::vector<unsigned long long>> roffs((size_t)T);
    std::vector<size_t> s_begin((size_t)T), s_end((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    pool.run(T, [&](int t) {
        size_t lo = begin + avail * t / T, hi = begin + avail * (t + 1) / T;
        size_t p = (t == 0 && !begin_unsynced) ? begin : sync_from(lo, hi);
        s_begin[t] = p;
        while (p < hi) {
            if (limit - p < 4) break;
            int32_t bs = rd32(u + p);
            if (bs < 32) { bad[t] = 1; break; }
            if (limit - p < 4 + (size_t)bs) break;  // incomplete: belongs to the next chunk
            roffs[t].push_back((unsigned long long)(p - begin));
            p += 4 + (size_t)bs;
        }
        s_end[t] = p;
    });
    if (bad[0]) { err = "truncated record"; return SQ_E_IO; }
    if (first_record) *first_record = s_begin[0];
    size_t good_end = s_end[0];
    int good = 1;
    for (int t = 1; t < T; ++t) {
        size_t hi = begin + avail * (t + 1) / T;
        if (s_begin[t] >= hi && roffs[t].empty() && good_end >= hi) { ++good; continue; }  // slice swallowed by a long record
        if (s_begin[t] != good_end) break;
        if (bad[t]) { err = "truncated record"; return SQ_E_IO; }
        good_end = s_end[t];
        ++good;
    }
    size_t total = 0;
    for (int t = 0; t < good; ++t) total += roffs[t].size();
    offs.reserve(total);
    for (int t = 0; t < good; ++t) offs.insert(offs.end(), roffs[t].begin(), roffs[t].end());
    size_t p = good_end;
    if (good < T)
        for (;;) {
            if (limit - p < 4) break;
            int32_t bs = rd32(u + p);
            if (bs < 32) { err = "truncated record"; return SQ_E_IO; }
            if (limit - p < 4 + (size_t)bs) break;
            offs.push_back((unsigned long long)(p - begin));
            p += 4 + (size_t)bs;
        }
    end = p;
    return SQ_OK;
}
}  // namespace
// forget the mapping and block index of the last file read (sq_drop_file_cache): the next read maps and indexes its file again
void drop_file_cache() {
    for (MapCache& mc : g_map_slots) {
        std::lock_guard<std::mutex> lk(mc.mu);
        mc.index.reset(); mc.index_total = 0; mc.map.reset(); mc.path.clear();
    }
}

// CPUs this process can really use: the affinity mask, cut by the cgroup's CPU quota (a container may see 256 logical
// CPUs and be allowed 16 of them)
int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[32]; long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
            const long long quota = std::atoll(q);
            if (quota > 0) n = std::min<long long>(n, (quota + period - 1) / period);
        }
        std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        long long quota = -1, period = 0;
        if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
        std::fclose(g);
        if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(h, "%lld", &period) != 1) period = 0; std::fclose(h); }
        if (quota > 0 && period > 0) n = std::min<long long>(n, (quota + period - 1) / period);
    }
    return std::max(1, n);
}
int scan_bam_file(const char* path, int n_threads, std::string& err, const RawSink& sink, const std::function<void(size_t)>& on_total, const RefRange* only, const GpuIngest& gpu, bool force_gpu, bool allow_bai, bool gpu_streams) {
    using clk = std::chrono::steady_clock;
    auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    double t_map = 0, t_inflate = 0, t_find = 0, t_wait = 0;
    const auto t_all = clk::now();
    // BGZF inflate on the GPU (k_inflate_lanes + k_lz_resolve): ~11 GB/s of inflated bytes on an
    // MI355X against ~0.5 GB/s per host thread, after a fixed start-up -- taken for files of at least 1 GiB when at
    // most 24 threads can really run (the caller's count, the affinity mask, the cgroup quota).  SQUID_GPU_INFLATE=1 / =0 forces / forbids it.  For a whole file the block index
    // is then built batch by batch while the GPU works (`lazy`: no page-table fill and no index walk up front).
    const char* gpu_env = std::getenv("SQUID_GPU_INFLATE");
    struct stat fst;
    const size_t file_bytes = ::stat(path, &fst) == 0 ? (size_t)fst.st_size : 0;
    // (round 3 left files to the host pipeline when more than 24 host threads could run; with the streamed read the GPU reader is ahead
    // of any number of them)
    const bool gpu_auto = file_bytes >= ((size_t)1 << 30);
    bool try_gpu = gpu && (force_gpu || (gpu_env ? std::atoi(gpu_env) != 0 : gpu_auto));
    // A chromosome shard with a .bai next to the BAM starts at the virtual offset of its first record and walks only the BGZF
    // blocks of its own range (lazily, like a whole file); without one, the whole file is indexed and the range is found by probing.
    BaiIndex bai;
    static const bool no_bai_env = std::getenv("SQUID_NO_BAI") != nullptr;
    bool use_bai = only && try_gpu && allow_bai && !no_bai_env && bai.load(path, -1);
    const bool lazy = try_gpu && (!only || use_bai);
    bool map_reused = false;
    MapCache& g_map_cache = map_cache_for(path);
    std::shared_ptr<FileMap> fm_hold = g_map_cache.acquire(path, !lazy, map_reused);
    if (!fm_hold) { err = std::string("cannot open bamfile ") + path; return SQ_E_IO; }
    struct { const uint8_t* p; size_t n; } fm{fm_hold->p, fm_hold->n};
    // lazy: a helper fills the page tables front to back (24 GB/s, far ahead of the ~6 GB/s the GPU pipeline reads at):
    // the copies to the device and the index walk then run over mapped pages instead of taking a fault every 4 KiB
    struct Prefault {
        std::thread th; std::atomic<bool> stop{false}; std::atomic<size_t> upto{0};  // upto: fill the tables up to this file offset, then wait
        std::atomic<size_t> from{0};  // (a shard: the fill starts at its own range)
        void start(const uint8_t* p, size_t n) {
#ifdef MADV_POPULATE_READ
            th = std::thread([this, p, n]() {
                const size_t step = (size_t)32 << 20;
                size_t o = 0;
                for (size_t f0 = from.load(); o < n && !stop.load(std::memory_order_relaxed);) {
                    if (from.load(std::memory_order_relaxed) != f0) { f0 = from.load(); o = f0 & ~(step - 1); }
                    if (o >= upto.load(std::memory_order_relaxed)) { std::this_thread::sleep_for(std::chrono::microseconds(500)); continue; }
                    if (madvise((void*)(p + o), std::min(step, n - o), MADV_POPULATE_READ) != 0) break;  // (then the readers fault the pages themselves)
                    o += step;
                }
            });
#else
            (void)p; (void)n;
#endif
        }
        void finish() { stop = true; if (th.joinable()) th.join(); }
        ~Prefault() { finish(); }
    } prefault;
    // (it stays about two batches ahead of the index walk: while it runs, everything that maps or unmaps memory -- device
    // allocations, thread stacks, large vectors -- queues behind it)
    const size_t prefault_ahead = (size_t)1400 << 20;
    prefault.upto = prefault_ahead;
    // (a reused mapping has its page tables filled; a GPU reader that streams the file itself never looks at the mapping beyond the header)
    const bool streams = lazy && gpu_streams && std::getenv("SQUID_NO_STREAM") == nullptr;
    if (lazy && !map_reused && !streams) prefault.start(fm.p, fm.n);
    std::vector<BgzfBlock> blocks;
    size_t total = 0;
    BgzfIndexer ix{fm.p, fm.n};
    bool index_cached = false;
    if (lazy && !only) {  // a whole file that has been walked before: its index is still there
        size_t cached_total = 0;
        if (std::shared_ptr<const std::vector<BgzfBlock>> ci = g_map_cache.get_index(fm_hold, cached_total)) {
            blocks = *ci;
            ix.p = fm.n; ix.total = cached_total;  // (complete: more() adds nothing)
            index_cached = true;
        }
    }
    if (lazy && !index_cached) { (void)ix.more(blocks, 64); if (ix.bad || blocks.empty()) { err = "not a BGZF file"; return SQ_E_IO; } }
    else {
        while (ix.more(blocks, (size_t)1 << 20)) {}
        if (!ix.complete()) { err = "not a BGZF file"; return SQ_E_IO; }
        total = ix.total;
    }
    t_map = since(t_all);
    if (!lazy && on_total) on_total(total);  // inflated size of the whole file: lets the sink size its arrays once
    n_threads = std::min(std::max(1, n_threads), 64);  // more helpers than that only cost their start-up
    // (the helpers are only started when the host pipeline runs: creating 15 threads next to the page-table helper costs 150+ ms)
    std::unique_ptr<Pool> pool_holder;
    if (!lazy) pool_holder.reset(new Pool(n_threads - 1));
    size_t only_begin = 0;
    const size_t kChunkBlocks = 1024;  // <= 64 MiB inflated per round (two such buffers; small enough to stay cheap to fault in and to free)
    RawBuf buf[2];
    std::vector<unsigned long long> offs[2];
    std::future<int> inflight;
    size_t nb = 0, carry = 0, nb_end = blocks.size();
    const uint8_t* carry_src = nullptr;
    int cur = 0, nref = -1;
    bool header_done = false, unsynced = false;
    int rc = SQ_OK;
    size_t hp = 0, first_rec_block = 0;
    if (only || try_gpu) {
        // the header first (a few blocks, inflated here): the number of references is part of the plausibility test
        std::vector<uint8_t> hdr;
        size_t hb = 0;
        auto hneed = [&](size_t n) { while (hdr.size() < n && (hb < blocks.size() || (lazy && ix.more(blocks, 64)))) { size_t o = hdr.size(); hdr.resize(o + blocks[hb].isize); if (!inflate_one(fm.p, blocks[hb], hdr.data() + o)) return false; ++hb; } return hdr.size() >= n; };
        if (!hneed(12) || std::memcmp(hdr.data(), "BAM\1", 4) != 0) { err = "not a BAM file"; return SQ_E_IO; }
        const int32_t ltext = rd32(hdr.data() + 4);
        if (!hneed(12 + (size_t)ltext)) { err = "truncated header"; return SQ_E_IO; }
        nref = rd32(hdr.data() + 8 + ltext);
        hp = 12 + (size_t)ltext;
        for (int i = 0; i < nref; ++i) { if (!hneed(hp + 4)) { err = "truncated header"; return SQ_E_IO; } int32_t ln = rd32(hdr.data() + hp); if (!hneed(hp + 8 + (size_t)ln)) { err = "truncated header"; return SQ_E_IO; } hp += 8 + (size_t)ln; }
        first_rec_block = (size_t)(std::upper_bound(blocks.begin(), blocks.end(), hp, [](size_t v, const BgzfBlock& b) { return v < b.uoff; }) - blocks.begin()) - 1;
        header_done = true;
        nb = first_rec_block;
        only_begin = hp - blocks[nb].uoff;
    }
    size_t gpu_file_bytes = fm.n;
    if (only && use_bai && (int)bai.ref_beg.size() != nref) {
        // an index of another file: take the probing path (it wants the whole block index)
        return scan_bam_file(path, n_threads, err, sink, on_total, only, gpu, force_gpu, false, gpu_streams);
    }
    if (only && use_bai) {
        // first record of the first owned reference that has records; the range ends at the first record of the first later
        // reference that has records (the last rank also owns the unplaced records: to the end of the file)
        unsigned long long vb = ~0ull, ve = ~0ull;
        for (int r = only->first_ref; r < only->end_ref && r < nref; ++r) if (bai.ref_beg[(size_t)r] != ~0ull) { vb = bai.ref_beg[(size_t)r]; break; }
        for (int r = only->end_ref; r < nref && !only->with_unplaced; ++r) if (bai.ref_beg[(size_t)r] != ~0ull) { ve = bai.ref_beg[(size_t)r]; break; }
        if (vb == ~0ull && only->with_unplaced) {
            // no owned reference has records, but the unplaced ones (behind every mapped record) are this rank's: start at the
            // last reference that has records -- the device side keeps only what the rank owns
            for (int r = nref - 1; r >= 0; --r) if (bai.ref_beg[(size_t)r] != ~0ull) { vb = bai.ref_beg[(size_t)r]; break; }
            if (vb == ~0ull) return scan_bam_file(path, n_threads, err, sink, on_total, only, gpu, force_gpu, false, gpu_streams);  // (no mapped record at all: probing path)
        }
        if (vb == ~0ull) return SQ_OK;  // nothing of this rank's in the file
        const size_t cb = (size_t)(vb >> 16), ce = ve == ~0ull ? (size_t)-1 : (size_t)(ve >> 16);
        if (cb >= fm.n || (ce != (size_t)-1 && ce < cb)) return scan_bam_file(path, n_threads, err, sink, on_total, only, gpu, force_gpu, false, gpu_streams);  // (a stale index)
        blocks.clear();
        ix.p = cb; ix.total = 0; ix.stop = ce; ix.bad = false;
        prefault.from = cb;
        prefault.upto = cb + prefault_ahead;
        if (!ix.more(blocks, 64) || ix.bad) return scan_bam_file(path, n_threads, err, sink, on_total, only, gpu, force_gpu, false, gpu_streams);
        nb = 0; first_rec_block = 0;
        only_begin = (size_t)(vb & 0xffff);
        unsynced = false;
        if (only_begin >= blocks[0].isize) return scan_bam_file(path, n_threads, err, sink, on_total, only, gpu, force_gpu, false, gpu_streams);
        gpu_file_bytes = (ce == (size_t)-1 ? fm.n : std::min(fm.n, ce + (size_t)(1 << 17))) - cb;  // sizes the record arrays: the shard's share of the file
    }
    if (only && !use_bai) {
        // A chromosome shard only inflates the blocks that can hold its records.  f(b) = RefID of the first record that
        // starts at or behind block b (found by the same plausible-chain search that the slices use) is monotone in a
        // coordinate-sorted file, unplaced records (-1) counting as behind every reference: two binary searches.
        std::vector<uint8_t> tmp;
        std::vector<unsigned long long> toffs;
        auto first_ref_from = [&](size_t b) -> int {  // nref when no record starts at or behind block b
            for (size_t span = 4;; span *= 2) {
                const size_t e = std::min(blocks.size(), b + span);
                const size_t base = blocks[b].uoff, bytes = (e == blocks.size() ? total : blocks[e].uoff) - base;
                tmp.resize(bytes + 64);
                for (size_t i = b; i < e; ++i) if (!inflate_one(fm.p, blocks[i], tmp.data() + (blocks[i].uoff - base))) return -2;
                size_t endp = 0, first = bytes;
                const size_t begin = b == first_rec_block ? hp - base : 0;
                Pool one(0);
                std::string e2;
                if (find_records(tmp.data(), begin, bytes, nref, one, 1, toffs, endp, e2, b != first_rec_block, &first)) return -2;
                if (!toffs.empty()) { int32_t r = rd32(tmp.data() + begin + toffs[0] + 4); return r < 0 ? nref : r; }
                if (e == blocks.size()) return nref;
            }
        };
        auto lower = [&](int ref) -> size_t {  // first block b in [first_rec_block, nblocks] with f(b) >= ref
            size_t lo = first_rec_block, hi = blocks.size();
            while (lo < hi) { size_t mid = (lo + hi) / 2; int f = first_ref_from(mid); if (f == -2) return (size_t)-1; if (f >= ref) hi = mid; else lo = mid + 1; }
            return lo;
        };
        const size_t bl = lower(only->first_ref), bh = only->with_unplaced ? blocks.size() : lower(only->end_ref);
        if (bl == (size_t)-1 || bh == (size_t)-1) { err = "corrupt BGZF block"; return SQ_E_IO; }
        nb = bl > first_rec_block ? bl - 1 : first_rec_block;   // records of the range may start in the block before
        nb_end = std::min(blocks.size(), bh + 2);                // the last owned record may run into the next blocks
        unsynced = nb != first_rec_block;
        if (nb >= nb_end) return SQ_OK;
        only_begin = unsynced ? 0 : hp - blocks[nb].uoff;  // (a range that starts in the header block begins right behind the header)
    }
    if (try_gpu && nb < nb_end) {
        // everything else on the GPU: the compressed blocks are copied as they are, inflated, cut into records and parsed there
        std::vector<BgzfRange> br(blocks.size());
        for (size_t i = 0; i < blocks.size(); ++i) br[i] = BgzfRange{blocks[i].coff, blocks[i].clen, blocks[i].isize, blocks[i].uoff};
        const IndexMore more = [&](std::vector<BgzfRange>& v) -> bool {
            const size_t before = blocks.size();
            prefault.upto = ix.p + prefault_ahead;
            const bool any = ix.more(blocks, blocks.size() < 8192 ? 2048 : 16384);
            for (size_t i = before; i < blocks.size(); ++i) v.push_back(BgzfRange{blocks[i].coff, blocks[i].clen, blocks[i].isize, blocks[i].uoff});
            return any;
        };
        const double t_before_gpu = since(t_all);
        GpuFileSrc src{path, ix.p, ix.total, ix.stop, false, false};
        const int r2 = gpu(fm.p, br, nb, lazy ? (size_t)-1 : nb_end, only_begin, !unsynced, nref, lazy ? more : IndexMore(), gpu_file_bytes, gpu_streams && !std::getenv("SQUID_NO_STREAM") ? &src : nullptr);
        if (r2 == SQ_OK && src.streamed && lazy && !index_cached) {  // the device side has walked the headers: its index is the index
            ix.p = src.walk_p; ix.total = src.walk_total; ix.bad = src.bad;
            blocks.resize(br.size());
            for (size_t i = 0; i < br.size(); ++i) { blocks[i].coff = br[i].coff; blocks[i].clen = br[i].clen; blocks[i].isize = br[i].isize; blocks[i].uoff = br[i].uoff; }
        }
        if (std::getenv("SQUID_INGEST_TIMING")) std::fprintf(stderr, "ingest %s: map+index %.1f, header %.1f, GPU inflate+parse path total %.1f ms (rc %d)\n", path, t_map, t_before_gpu, since(t_all), r2);
        if (r2 != 2) {
            if (r2 == SQ_OK && lazy && !ix.complete()) { err = "not a BGZF file"; return SQ_E_IO; }
            if (r2 == SQ_OK && lazy && !only && !index_cached && ix.p == fm.n) g_map_cache.set_index(fm_hold, blocks, ix.total);
            // unmapping 6 GB takes ~150 ms: off the caller's path, as at the end of the host pipeline -- together with the
            // large vectors (freeing those unmaps too and would wait for the big one)
            prefault.finish();
            auto* junk = new std::pair<std::vector<BgzfBlock>, std::vector<BgzfRange>>(std::move(blocks), std::move(br));
            std::thread([junk]() { delete junk; }).detach();
            if (std::getenv("SQUID_INGEST_TIMING")) std::fprintf(stderr, "ingest %s: returning after %.1f ms\n", path, since(t_all));
            return r2;
        }
        if (lazy && use_bai) { prefault.finish(); return scan_bam_file(path, n_threads, err, sink, on_total, only, gpu, force_gpu, false, gpu_streams); }  // (start over on the probing path)
        if (lazy) {  // the host pipeline wants the whole index (and its helpers)
            prefault.finish();
            pool_holder.reset(new Pool(n_threads - 1));
            while (ix.more(blocks, (size_t)1 << 20)) {}
            if (!ix.complete()) { err = "not a BGZF file"; return SQ_E_IO; }
            total = ix.total;
            nb_end = blocks.size();
            if (on_total) on_total(total);
        }
        // (the device-side checks were not satisfied: nothing was appended, continue with the host pipeline)
    }
    if (!pool_holder) pool_holder.reset(new Pool(n_threads - 1));
    Pool& pool = *pool_holder;
    while (nb < nb_end) {
        const size_t b1 = std::min(nb_end, nb + kChunkBlocks);
        const size_t base = blocks[nb].uoff, bytes = (b1 == blocks.size() ? total : blocks[b1].uoff) - base;
        RawBuf& u = buf[cur];
        u.resize(carry + bytes + 64);
        if (carry) std::memcpy(u.data(), carry_src, carry);  // (the buffer it comes from is only being read by the sink)
        auto ti0 = clk::now();
        std::atomic<bool> ok{true};
        const size_t first = nb;
        uint8_t* out = u.data() + carry;
        pool.run((int)(b1 - nb), [&](int i) { const BgzfBlock& b = blocks[first + (size_t)i]; if (!inflate_one(fm.p, b, out + (b.uoff - base))) ok = false; });
        t_inflate += since(ti0);
        if (!ok) { err = "corrupt BGZF block"; rc = SQ_E_IO; break; }
        nb = b1;
        const size_t limit = carry + bytes;
        size_t begin = 0;
        if (!header_done) {
            // BAM header: magic, text, reference dictionary (everything the caller needs from it came through sq_read_header)
            auto need = [&](size_t n) { return limit >= n; };
            if (!need(12) || std::memcmp(u.data(), "BAM\1", 4) != 0) { err = "not a BAM file"; rc = SQ_E_IO; break; }
            const int32_t ltext = rd32(u.data() + 4);
            if (!need(12 + (size_t)ltext)) { if (nb < nb_end) { carry = limit; carry_src = u.data(); cur ^= 1; continue; } err = "truncated header"; rc = SQ_E_IO; break; }
            nref = rd32(u.data() + 8 + ltext);
            size_t p = 12 + (size_t)ltext;
            bool complete = true;
            for (int i = 0; i < nref; ++i) {
                if (!need(p + 4)) { complete = false; break; }
                const int32_t ln = rd32(u.data() + p);
                if (!need(p + 8 + (size_t)ln)) { complete = false; break; }
                p += 8 + (size_t)ln;
            }
            if (!complete) { if (nb < nb_end) { carry = limit; carry_src = u.data(); cur ^= 1; continue; } err = "truncated header"; rc = SQ_E_IO; break; }
            header_done = true;
            begin = p;
        }
        if (only_begin) { begin = only_begin; only_begin = 0; }
        auto tf0 = clk::now();
        size_t end = begin;
        rc = find_records(u.data(), begin, limit, nref, pool, n_threads, offs[cur], end, err, unsynced);
        unsynced = false;
        t_find += since(tf0);
        if (rc) break;
        auto tw0 = clk::now();
        if (inflight.valid()) { rc = inflight.get(); if (rc) break; }
        t_wait += since(tw0);
        if (!offs[cur].empty()) {
            const uint8_t* data = u.data() + begin;
            const size_t nbytes = end - begin;
            const std::vector<unsigned long long>* o = &offs[cur];
            inflight = std::async(std::launch::async, [&sink, data, nbytes, o]() { return sink(data, nbytes, o->data(), (int64_t)o->size()); });
        }
        carry = limit - end;
        carry_src = u.data() + end;
        cur ^= 1;
    }
    if (inflight.valid()) { auto tw0 = clk::now(); int r2 = inflight.get(); t_wait += since(tw0); if (!rc) rc = r2; }
    if (std::getenv("SQUID_INGEST_TIMING"))
        std::fprintf(stderr, "ingest %s: map+index %.1f inflate %.1f boundaries %.1f waiting for the GPU sink %.1f total %.1f ms (%d threads)\n", path, t_map, t_inflate, t_find, t_wait, since(t_all), n_threads);
    {   // giving 130 MB of touched pages back to the system takes ~20 ms: do it off the caller's path
        uint8_t *p0 = buf[0].p, *p1 = buf[1].p;
        buf[0].p = nullptr; buf[1].p = nullptr;
        std::thread([p0, p1]() { std::free(p0); std::free(p1); }).detach();
    }
    return rc;
}

}  // namespace sq
