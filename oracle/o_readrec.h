// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED (see o_bam.h).  CPU restatement of the squid v1.5 read model and chimeric ingest:
//   src/SingleBamRec.h:25-61, src/ReadRec.cpp:10-146,171-232,267-283,329-413, src/Config.cpp:11-37.
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "o_bam.h"

namespace oracle {

// the 20 globals of src/Config.cpp:11-37 (defaults identical), bundled so the oracle is re-entrant
struct Params {
    uint16_t ReadLen = 0;
    bool UsingSTAR = true;
    bool Phred_Type = 1;
    uint16_t Max_LowPhred_Len = 10;
    uint8_t Min_Phred = 4;
    uint16_t Min_MapQual = 1;
    int Concord_Dist_Pos = 50000;
    int Concord_Dist_Idx = 20;
    int Min_Edge_Weight = 5;
    double DiscordantRatio = 8;
    int MaxAllowedDegree = 5;
    std::string Input_BAM, Input_Chim_BAM, Input_FASTA, Output_Prefix;
    bool Print_Graph = false, Print_Components_Ordering = false, Print_Total_Ordering = false,
         Print_Rearranged_Genome = false;
};

// src/SingleBamRec.h:25-61
struct SingleBamRec_t {
    int32_t RefID = 0, RefPos = 0, ReadPos = 0, MatchRef = 0, MatchRead = 0;
    uint8_t MapQual = 0;
    bool IsReverse = false, IsFirstRead = false;
    SingleBamRec_t() {}
    SingleBamRec_t(int32_t RefID, int32_t RefPos, int32_t ReadPos, int32_t MatchRef, int32_t MatchRead, uint8_t MapQual,
                   bool IsReverse, bool IsFirstRead)
        : RefID(RefID), RefPos(RefPos), ReadPos(ReadPos), MatchRef(MatchRef), MatchRead(MatchRead), MapQual(MapQual),
          IsReverse(IsReverse), IsFirstRead(IsFirstRead) {}
    bool operator<(const SingleBamRec_t& rhs) const { return RefID != rhs.RefID ? RefID < rhs.RefID : RefPos < rhs.RefPos; }
    bool operator>(const SingleBamRec_t& rhs) const { return RefID != rhs.RefID ? RefID > rhs.RefID : RefPos > rhs.RefPos; }
    bool Same(const SingleBamRec_t& rhs) const {
        return RefID == rhs.RefID && RefPos == rhs.RefPos && ReadPos == rhs.ReadPos && MatchRead == rhs.MatchRead &&
               MatchRef == rhs.MatchRef && IsReverse == rhs.IsReverse && IsFirstRead == rhs.IsFirstRead;
    }
    static bool CompReadPos(const SingleBamRec_t& lhs, const SingleBamRec_t& rhs) { return lhs.ReadPos < rhs.ReadPos; }
};

// src/ReadRec.h:35-58 + src/ReadRec.cpp
struct ReadRec_t {
    std::string Qname;
    std::vector<SingleBamRec_t> FirstRead, SecondMate;
    int FirstTotalLen = 0, SecondTotalLen = 0;
    // The reference leaves the low-Phred flag of the *other* mate indeterminate (ReadRec.cpp:39-44 sets one
    // of the two).  No live code path reads it before it is assigned from a real record (checked call
    // sites: SegmentGraph.cpp:251-258,668-683,1604), so `false` here cannot change a result.
    bool FirstLowPhred = false, SecondLowPhred = false;
    bool MultiFilter = false;

    ReadRec_t() {}
    // src/ReadRec.cpp:10-88
    ReadRec_t(const BamAlignment& record, const Params& P) {
        Qname = record.Name;
        if (Qname.size() >= 2 && (Qname.substr(Qname.size() - 2) == "/1" || Qname.substr(Qname.size() - 2) == "/2"))
            Qname = Qname.substr(0, Qname.size() - 2);
        MultiFilter = false;
        int32_t ReadPos = 0, RefPos = record.Position, TotalLen = 0, LowPhredLen = 0, tmpLowPhredLen = 0;
        for (const CigarOp& c : record.CigarData)
            if (c.Type == 'M' || c.Type == 'S' || c.Type == 'H' || c.Type == 'I' || c.Type == '=' || c.Type == 'X')
                TotalLen += c.Length;
        const char thr = (char)((P.Phred_Type ? 33 : 64) + P.Min_Phred);  // ReadRec.cpp:19-38
        for (size_t i = 0; i < record.Qualities.size(); i++) {
            if (record.Qualities[i] < thr) tmpLowPhredLen++;
            else tmpLowPhredLen = 0;
            if (LowPhredLen < tmpLowPhredLen) LowPhredLen = tmpLowPhredLen;
        }
        if (record.IsFirstMate()) {
            FirstTotalLen = TotalLen; SecondTotalLen = 0; FirstLowPhred = (LowPhredLen > P.Max_LowPhred_Len);
        } else {
            SecondTotalLen = TotalLen; FirstTotalLen = 0; SecondLowPhred = (LowPhredLen > P.Max_LowPhred_Len);
        }
        int HardClipOffset = 0;
        const std::vector<CigarOp>& C = record.CigarData;
        for (size_t ic = 0; ic < C.size(); ic++) {
            if (C[ic].Type == 'S' || C[ic].Type == 'H') {
                ReadPos += C[ic].Length;
                if (C[ic].Type == 'H') HardClipOffset += C[ic].Length;
            } else if (C[ic].Type == 'M' || C[ic].Type == '=') {
                int tmpRead = 0, tmpRef = 0;
                size_t ic2;
                for (ic2 = ic; ic2 < C.size() && C[ic2].Type != 'S' && C[ic2].Type != 'H' && C[ic2].Type != 'N'; ic2++) {
                    if (C[ic2].Type != 'D') tmpRead += C[ic2].Length;
                    if (C[ic2].Type != 'I') tmpRef += C[ic2].Length;
                }
                int polyAcount = 0, polyTcount = 0;
                assert(ReadPos >= HardClipOffset && ReadPos + tmpRead - HardClipOffset <= (int)record.QueryBases.size());
                for (int i = ReadPos - HardClipOffset; i < ReadPos + tmpRead - HardClipOffset; i++) {
                    char b = record.QueryBases[i];
                    if (b == 'a' || b == 'A') polyAcount++;
                    else if (b == 't' || b == 'T') polyTcount++;
                }
                if (1.0 * polyAcount / tmpRead < 0.75 && 1.0 * polyTcount / tmpRead < 0.75) {
                    SingleBamRec_t tmp(record.RefID, RefPos, ReadPos, tmpRef, tmpRead, (uint8_t)record.MapQuality,
                                       record.IsReverseStrand(), record.IsFirstMate());
                    if (record.IsReverseStrand()) tmp.ReadPos = TotalLen - ReadPos - tmpRead;
                    if (record.IsFirstMate()) FirstRead.push_back(tmp);
                    else SecondMate.push_back(tmp);
                }
                ReadPos += tmpRead;
                RefPos += tmpRef;
                ic = ic2 - 1;
            } else if (C[ic].Type == 'N')
                RefPos += C[ic].Length;
        }
    }
    bool operator<(const ReadRec_t& rhs) const { return Qname < rhs.Qname; }

    // src/ReadRec.cpp:90-117 (not a strict weak order; kept as is)
    static bool FrontSmallerThan(const ReadRec_t& lhs, const ReadRec_t& rhs) {
        auto cmp = [](const SingleBamRec_t& a, const SingleBamRec_t& b) { return a.RefID != b.RefID ? a.RefID < b.RefID : a.RefPos < b.RefPos; };
        if (lhs.FirstRead.size() != 0 && rhs.FirstRead.size() != 0) return cmp(lhs.FirstRead.front(), rhs.FirstRead.front());
        else if (lhs.SecondMate.size() != 0 && rhs.SecondMate.size() != 0) return cmp(lhs.SecondMate.front(), rhs.SecondMate.front());
        else if (lhs.FirstRead.size() != 0 && rhs.SecondMate.size() != 0) return cmp(lhs.FirstRead.front(), rhs.SecondMate.front());
        else if (lhs.SecondMate.size() != 0 && rhs.FirstRead.size() != 0) return cmp(lhs.SecondMate.front(), rhs.FirstRead.front());
        else return false;
    }
    // src/ReadRec.cpp:119-141
    static bool Equal(const ReadRec_t& lhs, const ReadRec_t& rhs) {
        auto ne = [](const SingleBamRec_t& a, const SingleBamRec_t& b) { return a.RefID != b.RefID || a.RefPos != b.RefPos || a.MatchRef != b.MatchRef; };
        bool same1 = false, same2 = false;
        if (lhs.FirstRead.size() == rhs.FirstRead.size() && lhs.SecondMate.size() == rhs.SecondMate.size()) {
            same1 = true;
            for (size_t i = 0; i < lhs.FirstRead.size(); i++) if (ne(lhs.FirstRead[i], rhs.FirstRead[i])) same1 = false;
            for (size_t i = 0; i < lhs.SecondMate.size(); i++) if (ne(lhs.SecondMate[i], rhs.SecondMate[i])) same1 = false;
        }
        if (lhs.FirstRead.size() == rhs.SecondMate.size() && lhs.SecondMate.size() == rhs.FirstRead.size()) {
            same2 = true;
            for (size_t i = 0; i < lhs.FirstRead.size(); i++) if (ne(lhs.FirstRead[i], rhs.SecondMate[i])) same2 = false;
            for (size_t i = 0; i < lhs.SecondMate.size(); i++) if (ne(lhs.SecondMate[i], rhs.FirstRead[i])) same2 = false;
        }
        return same1 || same2;
    }
    // src/ReadRec.cpp:143-146 (std::sort, unstable; tie order follows libstdc++ introsort, ledger B8)
    void SortbyReadPos() {
        std::sort(FirstRead.begin(), FirstRead.end(), SingleBamRec_t::CompReadPos);
        std::sort(SecondMate.begin(), SecondMate.end(), SingleBamRec_t::CompReadPos);
    }
    // src/ReadRec.cpp:171-176
    bool IsSingleAnchored() const { return (FirstRead.size() == 0 || SecondMate.size() == 0) && !MultiFilter; }
    // src/ReadRec.cpp:178-209
    bool IsEndDiscordant(bool _isfirst) const {
        const std::vector<SingleBamRec_t>& R = _isfirst ? FirstRead : SecondMate;
        if (R.size() <= 1) return false;
        for (size_t i = 0; i < R.size() - 1; i++) {
            if (R[i].RefID != R[i + 1].RefID || R[i].IsReverse != R[i + 1].IsReverse) return true;
            else if (!R[i].IsReverse && (R[i].RefPos < R[i + 1].RefPos) != (R[i].ReadPos < R[i + 1].ReadPos)) return true;
            else if (R[i].IsReverse && (R[i].RefPos < R[i + 1].RefPos) == (R[i].ReadPos < R[i + 1].ReadPos)) return true;
        }
        return false;
    }
    // src/ReadRec.cpp:211-228
    bool IsPairDiscordant(bool needcheck = true) const {
        if (FirstRead.size() == 0 || SecondMate.size() == 0) return false;
        if (needcheck) {
            if (IsEndDiscordant(true) || IsEndDiscordant(false)) return true;
        }
        if (FirstRead.front().RefID != SecondMate.back().RefID || FirstRead.front().IsReverse == SecondMate.back().IsReverse)
            return true;
        else if (!FirstRead.front().IsReverse && FirstRead.front().RefPos - FirstRead.front().ReadPos >
                                                     SecondMate.back().RefPos - (SecondTotalLen - SecondMate.back().ReadPos - SecondMate.back().MatchRead))
            return true;
        else if (!SecondMate.front().IsReverse && SecondMate.front().RefPos - SecondMate.front().ReadPos >
                                                      FirstRead.back().RefPos - (FirstTotalLen - FirstRead.back().ReadPos - FirstRead.back().MatchRead))
            return true;
        else
            return false;
    }
};

typedef std::vector<ReadRec_t> SBamrecord_t;

// src/ReadRec.cpp:267-283
inline bool BuildRefName(const std::string& bamfile, std::vector<std::string>& RefName, std::map<std::string, int>& RefTable,
                         std::vector<int>& RefLength) {
    RefName.clear(); RefTable.clear(); RefLength.clear();
    BamReader bamreader;
    bamreader.Open(bamfile);
    if (bamreader.IsOpen()) {
        int count = 0;
        for (const SamSequence& s : bamreader.Sequences()) {
            RefName.push_back(s.Name);
            RefTable[s.Name] = count++;
            RefLength.push_back(std::stoi(s.Length));
        }
        return true;
    }
    std::cout << "Cannot open bamfile " << bamfile << std::endl;
    return false;
}

// src/ReadRec.cpp:329-413.  Sets P.ReadLen (global ReadLen in the reference).
inline void BuildChimericSBamRecord(SBamrecord_t& SBamrecord, const std::string& bamfile, Params& P) {
    std::vector<uint16_t> sample_ReadLen;
    sample_ReadLen.reserve(5);
    SBamrecord.clear();
    SBamrecord_t newSBamrecord;
    BamReader bamreader;
    bamreader.Open(bamfile);
    if (bamreader.IsOpen()) {
        BamAlignment record;
        while (bamreader.GetNextAlignment(record)) {
            if (record.IsMapped() && !record.IsDuplicate()) {
                ReadRec_t tmp(record, P);
                SBamrecord.push_back(tmp);
                if (sample_ReadLen.size() < 5) sample_ReadLen.push_back((uint16_t)std::max(tmp.FirstTotalLen, tmp.SecondTotalLen));
            }
        }
        std::sort(SBamrecord.begin(), SBamrecord.end());  // by Qname, unstable (ledger B8)
        newSBamrecord.reserve(SBamrecord.size());
        for (SBamrecord_t::iterator it = SBamrecord.begin(); it != SBamrecord.end(); it++) {
            if (newSBamrecord.size() == 0 || it->Qname != newSBamrecord.back().Qname)
                newSBamrecord.push_back(*it);
            else {
                ReadRec_t& b = newSBamrecord.back();
                if (b.FirstTotalLen == 0 && it->FirstTotalLen != 0) { b.FirstTotalLen = it->FirstTotalLen; b.FirstLowPhred = it->FirstLowPhred; }
                if (b.SecondTotalLen == 0 && it->SecondTotalLen != 0) { b.SecondTotalLen = it->SecondTotalLen; b.SecondLowPhred = it->SecondLowPhred; }
                for (const SingleBamRec_t& s : it->FirstRead) b.FirstRead.push_back(s);
                for (const SingleBamRec_t& s : it->SecondMate) b.SecondMate.push_back(s);
            }
        }
        for (ReadRec_t& r : newSBamrecord) r.SortbyReadPos();
        std::sort(sample_ReadLen.begin(), sample_ReadLen.end());
        // ledger B7: an empty chimeric BAM reads sample_ReadLen[0] out of bounds in the reference; the
        // oracle refuses instead of inventing a value.
        if (sample_ReadLen.empty()) { std::cerr << "oracle: chimeric BAM has no usable record (reference: UB, ledger B7)\n"; std::exit(3); }
        P.ReadLen = sample_ReadLen[(int)sample_ReadLen.size() / 2];
        bamreader.Close();
    }
    std::sort(newSBamrecord.begin(), newSBamrecord.end(), ReadRec_t::FrontSmallerThan);
    // remove PCR duplicates (ReadRec.cpp:387-409)
    SBamrecord.clear();
    for (SBamrecord_t::iterator it = newSBamrecord.begin(); it != newSBamrecord.end(); it++) {
        if (SBamrecord.size() == 0) SBamrecord.push_back(*it);
        else if (it->FirstRead.size() == 0 || SBamrecord.back().FirstRead.size() == 0) SBamrecord.push_back(*it);
        else if (it->FirstRead.front().RefID != SBamrecord.back().FirstRead.front().RefID ||
                 it->FirstRead.front().RefPos != SBamrecord.back().FirstRead.front().RefPos)
            SBamrecord.push_back(*it);
        else {
            bool isdup = false;
            for (SBamrecord_t::reverse_iterator it2 = SBamrecord.rbegin(); it2 != SBamrecord.rend(); it2++) {
                if (it2->FirstRead.size() == 0 || it->FirstRead.front().RefID != it2->FirstRead.front().RefID ||
                    it->FirstRead.front().RefPos != it2->FirstRead.front().RefPos)
                    break;
                if (ReadRec_t::Equal(*it, *it2)) { isdup = true; break; }
            }
            if (!isdup) SBamrecord.push_back(*it);
        }
    }
}

}  // namespace oracle
