// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED: the reference cannot be built in this image (it needs BamTools, GLPK and Boost,
// none of which is installed or vendored), and it ships no tests or golden vectors.  This file
// restates the part of the BamTools 2.4.0 surface that squid v1.5 uses from
// the public SAM/BAM specification: BGZF inflate, BAM record decode, header @SQ parse, tag lookup.
//
// Reference call sites served by this reader:
//   src/ReadRec.cpp:271-279      BamReader::Open/GetHeader, SamHeader.Sequences (Name, Length)
//   src/ReadRec.cpp:340-343      GetNextAlignment over the chimeric BAM
//   src/SegmentGraph.cpp:293-302 GetNextAlignment, HasTag("XA"), HasTag("IH"), GetTag("IH", int&)
//   src/SegmentGraph.cpp:3149    BamAlignment::GetEndPosition()
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace oracle {

struct CigarOp {
    char Type;
    uint32_t Length;
};

// Mirror of the BamTools::BamAlignment members the reference reads.
struct BamAlignment {
    std::string Name;
    int32_t RefID = -1, Position = -1, MateRefID = -1, MatePosition = -1;
    uint16_t MapQuality = 0;
    uint16_t AlignmentFlag = 0;
    std::vector<CigarOp> CigarData;
    std::string Qualities;   // Phred+33 characters, as BamTools hands them out
    std::string QueryBases;  // "=ACMGRSVTWYHKDBN" letters
    std::string TagData;     // raw aux bytes

    bool IsMapped() const { return !(AlignmentFlag & 0x4); }
    bool IsMateMapped() const { return !(AlignmentFlag & 0x8); }
    bool IsReverseStrand() const { return AlignmentFlag & 0x10; }
    bool IsMateReverseStrand() const { return AlignmentFlag & 0x20; }
    bool IsFirstMate() const { return AlignmentFlag & 0x40; }
    bool IsSecondMate() const { return AlignmentFlag & 0x80; }
    bool IsProperPair() const { return AlignmentFlag & 0x2; }
    bool IsDuplicate() const { return AlignmentFlag & 0x400; }

    // walk the aux block; returns pointer to the type byte of `tag` or nullptr
    const uint8_t* FindTag(const char* tag) const {
        const uint8_t* p = (const uint8_t*)TagData.data();
        const uint8_t* e = p + TagData.size();
        while (p + 3 <= e) {
            bool hit = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
            const uint8_t* t = p + 2;
            if (hit) return t;
            p = SkipValue(t, e);
            if (!p) return nullptr;
        }
        return nullptr;
    }
    bool HasTag(const char* tag) const { return FindTag(tag) != nullptr; }
    // GetTag(tag, int&): integer-typed tags copy their little-endian bytes into a zeroed int.
    bool GetTagInt(const char* tag, int& dst) const {
        const uint8_t* t = FindTag(tag);
        if (!t) return false;
        uint32_t v = 0;
        switch (*t) {
            case 'c': case 'C': case 'A': v = t[1]; break;
            case 's': case 'S': v = t[1] | (t[2] << 8); break;
            case 'i': v = t[1] | (t[2] << 8) | (t[3] << 16) | ((uint32_t)t[4] << 24); break;
            default: return false;  // incl. 'I': BamTools' TagTypeHelper<int32_t>::CanConvertFrom refuses UINT32 (dst keeps its value)
        }
        dst = (int)v;
        return true;
    }
    // half-open end: Position + sum of M, D, N, =, X lengths (GetEndPosition(false,false))
    int GetEndPosition() const {
        int e = Position;
        for (const CigarOp& c : CigarData)
            if (c.Type == 'M' || c.Type == 'D' || c.Type == 'N' || c.Type == '=' || c.Type == 'X') e += (int)c.Length;
        return e;
    }

private:
    static const uint8_t* SkipValue(const uint8_t* t, const uint8_t* e) {
        auto sz = [](uint8_t c) -> int {
            switch (c) {
                case 'A': case 'c': case 'C': return 1;
                case 's': case 'S': return 2;
                case 'i': case 'I': case 'f': return 4;
                default: return 0;
            }
        };
        if (t >= e) return nullptr;
        uint8_t ty = *t++;
        if (int s = sz(ty)) return t + s <= e ? t + s : nullptr;
        if (ty == 'Z' || ty == 'H') {
            while (t < e && *t) ++t;
            return t < e ? t + 1 : nullptr;
        }
        if (ty == 'B') {
            if (t + 5 > e) return nullptr;
            int s = sz(t[0]);
            uint32_t n;
            std::memcpy(&n, t + 1, 4);
            t += 5 + (size_t)s * n;
            return t <= e ? t : nullptr;
        }
        return nullptr;
    }
};

struct SamSequence {
    std::string Name, Length;
};

class BamReader {
public:
    ~BamReader() { Close(); }
    bool Open(const std::string& path) {
        Close();
        fp_ = std::fopen(path.c_str(), "rb");
        if (!fp_) return false;
        buf_.clear();
        off_ = 0;
        eof_ = false;
        char magic[4];
        if (!ReadBytes(magic, 4) || std::memcmp(magic, "BAM\1", 4) != 0) { Close(); return false; }
        int32_t ltext;
        if (!ReadBytes(&ltext, 4)) { Close(); return false; }
        header_text_.resize(ltext);
        if (ltext && !ReadBytes(&header_text_[0], ltext)) { Close(); return false; }
        int32_t nref;
        if (!ReadBytes(&nref, 4)) { Close(); return false; }
        for (int i = 0; i < nref; ++i) {
            int32_t ln, len;
            ReadBytes(&ln, 4);
            std::string nm(ln, 0);
            ReadBytes(&nm[0], ln);
            ReadBytes(&len, 4);
        }
        ParseHeaderText();
        return true;
    }
    bool IsOpen() const { return fp_ != nullptr; }
    void Close() {
        if (fp_) std::fclose(fp_);
        fp_ = nullptr;
    }
    // the reference reads names/lengths from the header TEXT (ReadRec.cpp:274-279)
    const std::vector<SamSequence>& Sequences() const { return seqs_; }

    bool GetNextAlignment(BamAlignment& a) {
        int32_t bs;
        if (!ReadBytes(&bs, 4)) return false;
        rec_.resize(bs);
        if (!ReadBytes(rec_.data(), bs)) return false;
        const uint8_t* p = rec_.data();
        auto i32 = [&](int o) { int32_t v; std::memcpy(&v, p + o, 4); return v; };
        auto u16 = [&](int o) { uint16_t v; std::memcpy(&v, p + o, 2); return v; };
        a.RefID = i32(0);
        a.Position = i32(4);
        int lname = p[8];
        a.MapQuality = p[9];
        int ncig = u16(12);
        a.AlignmentFlag = u16(14);
        int lseq = i32(16);
        a.MateRefID = i32(20);
        a.MatePosition = i32(24);
        const uint8_t* q = p + 32;
        a.Name.assign((const char*)q, lname > 0 ? lname - 1 : 0);
        q += lname;
        a.CigarData.resize(ncig);
        static const char ops[] = "MIDNSHP=X???????";
        for (int i = 0; i < ncig; ++i) {
            uint32_t v;
            std::memcpy(&v, q + 4 * i, 4);
            a.CigarData[i].Type = ops[v & 0xf];
            a.CigarData[i].Length = v >> 4;
        }
        q += 4 * ncig;
        static const char bases[] = "=ACMGRSVTWYHKDBN";
        a.QueryBases.resize(lseq);
        for (int i = 0; i < lseq; ++i) a.QueryBases[i] = bases[(q[i >> 1] >> ((~i & 1) << 2)) & 0xf];
        q += (lseq + 1) / 2;
        a.Qualities.resize(lseq);
        for (int i = 0; i < lseq; ++i) a.Qualities[i] = (char)(q[i] + 33);
        q += lseq;
        a.TagData.assign((const char*)q, (const char*)(p + bs));
        return true;
    }

private:
    bool Fill() {
        // one BGZF block
        uint8_t h[18];
        if (std::fread(h, 1, 18, fp_) != 18) { eof_ = true; return false; }
        if (h[0] != 0x1f || h[1] != 0x8b) { eof_ = true; return false; }
        int xlen = h[10] | (h[11] << 8);
        int bsize = -1;
        std::vector<uint8_t> extra(xlen);
        std::memcpy(extra.data(), h + 12, std::min(6, xlen));
        if (xlen > 6 && std::fread(extra.data() + 6, 1, xlen - 6, fp_) != (size_t)(xlen - 6)) { eof_ = true; return false; }
        for (int o = 0; o + 4 <= xlen;) {
            int slen = extra[o + 2] | (extra[o + 3] << 8);
            if (extra[o] == 'B' && extra[o + 1] == 'C') bsize = (extra[o + 4] | (extra[o + 5] << 8)) + 1;
            o += 4 + slen;
        }
        if (bsize < 0) { eof_ = true; return false; }
        int clen = bsize - 12 - xlen - 8;
        cbuf_.resize(clen + 8);
        if (std::fread(cbuf_.data(), 1, clen + 8, fp_) != (size_t)(clen + 8)) { eof_ = true; return false; }
        uint32_t isize;
        std::memcpy(&isize, cbuf_.data() + clen + 4, 4);
        // drop consumed bytes, append the inflated block
        if (off_ > 0) { buf_.erase(buf_.begin(), buf_.begin() + off_); off_ = 0; }
        size_t old = buf_.size();
        buf_.resize(old + isize);
        if (isize) {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = cbuf_.data();
            zs.avail_in = clen;
            zs.next_out = buf_.data() + old;
            zs.avail_out = isize;
            int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) { eof_ = true; return false; }
        }
        return true;
    }
    bool ReadBytes(void* dst, size_t n) {
        while (buf_.size() - off_ < n) {
            if (eof_ || !Fill()) {
                if (buf_.size() - off_ < n) return false;
            }
        }
        std::memcpy(dst, buf_.data() + off_, n);
        off_ += n;
        return true;
    }
    void ParseHeaderText() {
        seqs_.clear();
        size_t i = 0;
        while (i < header_text_.size()) {
            size_t e = header_text_.find('\n', i);
            if (e == std::string::npos) e = header_text_.size();
            std::string line = header_text_.substr(i, e - i);
            i = e + 1;
            if (line.compare(0, 3, "@SQ") != 0) continue;
            SamSequence s;
            size_t f = 0;
            while (f < line.size()) {
                size_t t = line.find('\t', f);
                if (t == std::string::npos) t = line.size();
                std::string fld = line.substr(f, t - f);
                if (fld.compare(0, 3, "SN:") == 0) s.Name = fld.substr(3);
                if (fld.compare(0, 3, "LN:") == 0) s.Length = fld.substr(3);
                f = t + 1;
            }
            seqs_.push_back(s);
        }
    }
    FILE* fp_ = nullptr;
    std::vector<uint8_t> buf_, cbuf_, rec_;
    size_t off_ = 0;
    bool eof_ = false;
    std::string header_text_;
    std::vector<SamSequence> seqs_;
};

}  // namespace oracle
