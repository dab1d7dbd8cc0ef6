// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED: utils/JunctionSequence.cpp needs Boost and BamTools like the rest of the reference and cannot be built here; it
// ships no fixture.  CPU restatement of that utility (second half): `_sv.txt` + the chimeric BAM + the
// genome FASTA -> <prefix>_junc_precise.fa, _junc_relax.fa, _junc_alt.fa.  Follows utils/JunctionSequence.cpp line by line:
//   Breakpoint_t / SV_t :16-87, ReadBEDPE :89-110, SVfromAlignment :112-168, NearestSV :170-200, FindReadSupport :202-221,
//   ExactSequence :223-396, ReadGenome :398-420, the three writers :422-517, main :527-557.
// Kept as written: BP2's left-hand extension restarts from the first read breakpoint (:318-321) where BP1's continues from the hit
// (:288-291); the "differs" test of a left BP1 compares with the read's BP2 (:358); names the tables do not know land on index 0
// (std::map::operator[], :105-106, :409,417); a base outside the complement table becomes '\0' (:11).
#pragma once
#include <algorithm>
#include <climits>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "o_readrec.h"

namespace oracle {
namespace junction {

struct Breakpoint_t {
    int Chr = 0, StartPos = 0, EndPos = 0;
    bool IsLeft = false;
    Breakpoint_t() {}
    Breakpoint_t(int c, int s, int e, bool l) : Chr(c), StartPos(s), EndPos(e), IsLeft(l) {}
    bool operator<(const Breakpoint_t& r) const {
        if (Chr != r.Chr) return Chr < r.Chr;
        if (StartPos != r.StartPos) return StartPos < r.StartPos;
        if (EndPos != r.EndPos) return EndPos < r.EndPos;
        return IsLeft < r.IsLeft;
    }
    bool operator==(const Breakpoint_t& r) const { return Chr == r.Chr && StartPos == r.StartPos && EndPos == r.EndPos && IsLeft == r.IsLeft; }
    bool operator!=(const Breakpoint_t& r) const { return !(*this == r); }
};
struct SV_t {
    Breakpoint_t BP1, BP2;
    SV_t() {}
    SV_t(Breakpoint_t a, Breakpoint_t b) { if (a < b) { BP1 = a; BP2 = b; } else { BP1 = b; BP2 = a; } }
    bool operator<(const SV_t& r) const { return BP1 != r.BP1 ? BP1 < r.BP1 : BP2 < r.BP2; }
    bool operator==(const SV_t& r) const { return BP1 == r.BP1 && BP2 == r.BP2; }
};

inline std::vector<std::string> SplitAny(const std::string& s, const char* seps) {  // boost::split(.., is_any_of(seps)), no token compression
    std::vector<std::string> out(1);
    for (char c : s) { if (std::strchr(seps, c)) out.push_back(std::string()); else out.back().push_back(c); }
    return out;
}

inline void ReadBEDPE(const std::string& file, std::map<std::string, int>& RefTable, std::vector<SV_t>& SVs) {
    SVs.clear();
    std::ifstream input(file);
    std::string line;
    while (std::getline(input, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> strs = SplitAny(line, "\t");
        if (strs.size() < 10) continue;
        if (strs[0][0] == 'M' || strs[0][0] == 'G' || strs[0][0] == 'K' || strs[3][0] == 'M' || strs[3][0] == 'G' || strs[3][0] == 'K') continue;
        Breakpoint_t bp1(RefTable[strs[0]], std::stoi(strs[1]), std::stoi(strs[2]), strs[8] == "-");
        Breakpoint_t bp2(RefTable[strs[3]], std::stoi(strs[4]), std::stoi(strs[5]), strs[9] == "-");
        SVs.push_back(SV_t(bp1, bp2));
    }
}

inline int SVfromAlignment(const ReadRec_t& r, std::vector<SV_t>& tmpSVs) {
    int flag = -1;
    auto mate = [&](const std::vector<SingleBamRec_t>& R) {
        if (R.size() == 0) return;
        for (int i = 0; i < (int)R.size() - 1; i++) {
            bool is_discordant = false;
            if (R[i].RefID != R[i + 1].RefID || R[i].IsReverse != R[i + 1].IsReverse) is_discordant = true;
            else if (!R[i].IsReverse && (R[i].RefPos < R[i + 1].RefPos) != (R[i].ReadPos < R[i + 1].ReadPos)) is_discordant = true;
            else if (R[i].IsReverse && (R[i].RefPos < R[i + 1].RefPos) == (R[i].ReadPos < R[i + 1].ReadPos)) is_discordant = true;
            if (is_discordant) {
                Breakpoint_t bp1(R[i].RefID, R[i].RefPos, R[i].RefPos + R[i].MatchRef, R[i].IsReverse);
                Breakpoint_t bp2(R[i + 1].RefID, R[i + 1].RefPos, R[i + 1].RefPos + R[i + 1].MatchRef, !R[i + 1].IsReverse);
                tmpSVs.push_back(SV_t(bp1, bp2));
                flag = 0;
            }
        }
    };
    mate(r.FirstRead);
    mate(r.SecondMate);
    if (flag < 0 && r.FirstRead.size() > 0 && r.SecondMate.size() > 0 && r.IsPairDiscordant(false)) {
        const std::vector<SingleBamRec_t>&F = r.FirstRead, &S = r.SecondMate;
        bool partial = false;
        if (F.front().ReadPos > 12 && !r.FirstLowPhred) partial = true;
        if (r.FirstTotalLen - F.back().ReadPos - F.back().MatchRead > 12 && !r.FirstLowPhred) partial = true;
        if (S.front().ReadPos > 12 && !r.SecondLowPhred) partial = true;
        if (r.SecondTotalLen - S.back().ReadPos - S.back().MatchRead > 12 && !r.SecondLowPhred) partial = true;
        if (partial) {
            Breakpoint_t bp1(F.back().RefID, F.back().RefPos, F.back().RefPos + F.back().MatchRef, F.back().IsReverse);
            Breakpoint_t bp2(S.back().RefID, S.back().RefPos, S.back().RefPos + S.back().MatchRef, S.back().IsReverse);
            tmpSVs.push_back(SV_t(bp1, bp2));
            flag = 0;
        }
    }
    return flag;
}

inline int NearestSV(const SV_t& n, const std::vector<SV_t>& SVs, int thresh1 = 5, int thresh2 = 300) {
    int optarg = -1, optvalue = INT_MAX;
    for (int i = 0; i < (int)SVs.size(); i++) {
        const SV_t& s = SVs[i];
        if (n.BP1.Chr != s.BP1.Chr || n.BP2.Chr != s.BP2.Chr) continue;
        if (n.BP1.IsLeft != s.BP1.IsLeft || n.BP2.IsLeft != s.BP2.IsLeft) continue;
        if (n.BP1.IsLeft && (n.BP1.StartPos < s.BP1.StartPos - thresh1 || n.BP1.StartPos > s.BP1.StartPos + thresh2)) continue;
        else if (!n.BP1.IsLeft && (n.BP1.EndPos < s.BP1.EndPos - thresh2 || n.BP1.EndPos > s.BP1.EndPos + thresh1)) continue;
        if (n.BP2.IsLeft && (n.BP2.StartPos < s.BP2.StartPos - thresh1 || n.BP2.StartPos > s.BP2.StartPos + thresh2)) continue;
        else if (!n.BP2.IsLeft && (n.BP2.EndPos < s.BP2.EndPos - thresh2 || n.BP2.EndPos > s.BP2.EndPos + thresh1)) continue;
        int deviation = 0;
        deviation += n.BP1.IsLeft ? std::abs(n.BP1.StartPos - s.BP1.StartPos) : std::abs(n.BP1.EndPos - s.BP1.EndPos);
        deviation += n.BP2.IsLeft ? std::abs(n.BP2.StartPos - s.BP2.StartPos) : std::abs(n.BP2.EndPos - s.BP2.EndPos);
        if (deviation < optvalue) { optvalue = deviation; optarg = i; }
    }
    return optarg;
}

inline void FindReadSupport(const SBamrecord_t& Chimrecord, std::vector<SV_t>& SVs, std::vector<std::vector<SV_t>>& ReadSVs) {
    ReadSVs.assign(SVs.size(), std::vector<SV_t>());
    for (const ReadRec_t& r : Chimrecord) {
        std::vector<SV_t> tmp;
        if (SVfromAlignment(r, tmp) != -1)
            for (const SV_t& t : tmp) { int ind = NearestSV(t, SVs); if (ind != -1) ReadSVs[ind].push_back(t); }
    }
}

// returns false when the reference would abort on its assert (:383-384)
inline bool ExactSequence(std::vector<SV_t>& SVs, const std::vector<std::vector<SV_t>>& ReadSVs, std::vector<int>& NumSupports, std::vector<std::vector<SV_t>>& AltSVs, std::vector<bool>& flags, int thresh = 5) {
    NumSupports.clear();
    AltSVs.clear();
    flags.assign(SVs.size(), false);
    auto byEnd = [](const Breakpoint_t& a, const Breakpoint_t& b) {
        if (a.Chr != b.Chr) return a.Chr < b.Chr;
        if (a.EndPos != b.EndPos) return a.EndPos < b.EndPos;
        if (a.StartPos != b.StartPos) return a.StartPos < b.StartPos;
        return a.IsLeft < b.IsLeft;
    };
    for (size_t i = 0; i < ReadSVs.size(); i++) {
        NumSupports.push_back(0);
        std::vector<SV_t> alternativeSVs, tmpalternativeSVs;
        const std::vector<SV_t>& readsvs = ReadSVs[i];
        SV_t& sv = SVs[i];
        if (readsvs.size() == 0) { AltSVs.push_back(alternativeSVs); continue; }
        std::vector<Breakpoint_t> BP1s, BP2s;
        for (const SV_t& s : readsvs) { BP1s.push_back(s.BP1); BP2s.push_back(s.BP2); }
        if (sv.BP1.IsLeft) std::sort(BP1s.begin(), BP1s.end()); else std::sort(BP1s.begin(), BP1s.end(), byEnd);
        if (sv.BP2.IsLeft) std::sort(BP2s.begin(), BP2s.end()); else std::sort(BP2s.begin(), BP2s.end(), byEnd);
        int support_bp1 = 0, support_bp2 = 0;
        for (const Breakpoint_t& b : BP1s) if ((sv.BP1.IsLeft && std::abs(sv.BP1.StartPos - b.StartPos) < thresh) || (!sv.BP1.IsLeft && std::abs(sv.BP1.EndPos - b.EndPos) < thresh)) support_bp1++;
        for (const Breakpoint_t& b : BP2s) if ((sv.BP2.IsLeft && std::abs(sv.BP2.StartPos - b.StartPos) < thresh) || (!sv.BP2.IsLeft && std::abs(sv.BP2.EndPos - b.EndPos) < thresh)) support_bp2++;
        if (!support_bp1 || !support_bp2) { AltSVs.push_back(alternativeSVs); continue; }
        bool flag_bp1 = false, flag_bp2 = false;
        if (sv.BP1.IsLeft) {
            size_t it = 0;
            while (std::abs(BP1s[it].StartPos - sv.BP1.StartPos) >= thresh) it++;
            int rightmost = BP1s[it].EndPos;
            for (; it < BP1s.size(); it++) if (BP1s[it].StartPos < rightmost) rightmost = std::max(rightmost, BP1s[it].EndPos);
            if (sv.BP1.StartPos < rightmost) { sv.BP1.EndPos = std::min(rightmost, sv.BP1.EndPos); flag_bp1 = true; }
        } else {
            size_t it = BP1s.size();  // reverse iterator: element it - 1
            while (std::abs(BP1s[it - 1].EndPos - sv.BP1.EndPos) >= thresh) it--;
            int leftmost = BP1s[it - 1].StartPos;
            for (; it > 0; it--) if (BP1s[it - 1].EndPos > leftmost) leftmost = std::min(leftmost, BP1s[it - 1].StartPos);
            if (leftmost < sv.BP1.EndPos) { sv.BP1.StartPos = std::max(leftmost, sv.BP1.StartPos); flag_bp1 = true; }
        }
        if (sv.BP2.IsLeft) {
            size_t it = 0;
            while (std::abs(BP2s[it].StartPos - sv.BP2.StartPos) >= thresh) it++;
            int rightmost = BP2s[it].EndPos;
            for (size_t k = 0; k < BP2s.size(); k++) if (BP2s[k].StartPos < rightmost) rightmost = std::max(rightmost, BP2s[k].EndPos);  // (from the beginning: :318-321)
            if (sv.BP2.StartPos < rightmost) { sv.BP2.EndPos = std::min(rightmost, sv.BP2.EndPos); flag_bp2 = true; }
        } else {
            size_t it = BP2s.size();
            while (std::abs(BP2s[it - 1].EndPos - sv.BP2.EndPos) >= thresh) it--;
            int leftmost = BP2s[it - 1].StartPos;
            for (size_t k = BP2s.size(); k > 0; k--) if (BP2s[k - 1].EndPos > leftmost) leftmost = std::min(leftmost, BP2s[k - 1].StartPos);  // (from the end: :333-336)
            if (leftmost < sv.BP2.EndPos) { sv.BP2.StartPos = std::max(leftmost, sv.BP2.StartPos); flag_bp2 = true; }
        }
        if (flag_bp1 && flag_bp2) { flags[i] = true; NumSupports.back() = std::min(support_bp1, support_bp2); }
        for (const SV_t& it : readsvs) {
            SV_t altsv(sv.BP1, sv.BP2);
            bool hasalt_bp1 = false, hasalt_bp2 = false, diffalt_bp1 = false, diffalt_bp2 = false;
            if (sv.BP1.IsLeft == it.BP1.IsLeft) {
                if (sv.BP1.IsLeft && std::abs(sv.BP1.StartPos - it.BP1.StartPos) < thresh) {
                    altsv.BP1.StartPos = it.BP1.StartPos; hasalt_bp1 = true;
                    if (sv.BP1.StartPos != it.BP2.StartPos) diffalt_bp1 = true;  // (:358 compares with BP2)
                } else if (!sv.BP1.IsLeft && std::abs(sv.BP1.EndPos - it.BP1.EndPos) < thresh) {
                    altsv.BP1.EndPos = it.BP1.EndPos; hasalt_bp1 = true;
                    if (sv.BP1.EndPos != it.BP1.EndPos) diffalt_bp1 = true;
                }
            }
            if (sv.BP2.IsLeft == it.BP2.IsLeft) {
                if (sv.BP2.IsLeft && std::abs(sv.BP2.StartPos - it.BP2.StartPos) < thresh) {
                    altsv.BP2.StartPos = it.BP2.StartPos; hasalt_bp2 = true;
                    if (sv.BP2.StartPos != it.BP2.StartPos) diffalt_bp2 = true;
                } else if (!sv.BP2.IsLeft && std::abs(sv.BP2.EndPos - it.BP2.EndPos) < thresh) {
                    altsv.BP2.EndPos = it.BP2.EndPos; hasalt_bp2 = true;
                    if (sv.BP2.EndPos != it.BP2.EndPos) diffalt_bp2 = true;
                }
            }
            if (hasalt_bp1 && hasalt_bp2 && (diffalt_bp1 || diffalt_bp2)) tmpalternativeSVs.push_back(altsv);
        }
        if (tmpalternativeSVs.size() != 0 && !flags[i]) return false;
        std::sort(tmpalternativeSVs.begin(), tmpalternativeSVs.end());
        for (const SV_t& a : tmpalternativeSVs) if (alternativeSVs.size() == 0 || !(alternativeSVs.back() == a)) alternativeSVs.push_back(a);
        AltSVs.push_back(alternativeSVs);
    }
    return true;
}

inline void ReadGenome(const std::string& FAfile, std::vector<std::string>& Genome, std::map<std::string, int>& RefTable) {
    Genome.assign(RefTable.size(), std::string());
    std::ifstream input(FAfile);
    std::string line, prevname, preseq;
    while (std::getline(input, line)) {
        if (!line.empty() && line[0] == '>') {
            if (prevname != "") Genome[RefTable[prevname]] = preseq;
            prevname = SplitAny(line, " \t")[0].substr(1);
            preseq = "";
        } else preseq += line;
    }
    Genome[RefTable[prevname]] = preseq;
}

inline char Complement(char c) {
    switch (std::toupper((unsigned char)c)) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'W'; case 'W': return 'S';
        case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D'; case 'N': return 'N'; case '.': return '.'; case '-': return '-';
    }
    return '\0';
}
inline void ReverseComplement(std::string& s) { for (char& c : s) c = Complement(c); std::reverse(s.begin(), s.end()); }

// false: an SV reaches beyond its chromosome's sequence (the reference asserts, :431,459,492)
inline bool WriteOne(std::ofstream& out, const std::string& head, const SV_t& sv, const std::vector<std::string>& Genome, const std::vector<std::string>& RefName, const std::string& tail) {
    if ((int)Genome[sv.BP1.Chr].size() < sv.BP1.EndPos || (int)Genome[sv.BP2.Chr].size() < sv.BP2.EndPos) return false;
    std::string seq1 = Genome[sv.BP1.Chr].substr(sv.BP1.StartPos, sv.BP1.EndPos - sv.BP1.StartPos), seq2 = Genome[sv.BP2.Chr].substr(sv.BP2.StartPos, sv.BP2.EndPos - sv.BP2.StartPos);
    if (sv.BP1.IsLeft) ReverseComplement(seq1);
    if (!sv.BP2.IsLeft) ReverseComplement(seq2);
    const std::string seq = seq1 + seq2;
    out << head << " " << RefName[sv.BP1.Chr] << ":" << sv.BP1.StartPos << ":" << sv.BP1.EndPos << ":" << (sv.BP1.IsLeft ? "-" : "+") << " " << RefName[sv.BP2.Chr] << ":" << sv.BP2.StartPos << ":" << sv.BP2.EndPos << ":"
        << (sv.BP2.IsLeft ? "+" : "-") << tail << std::endl;
    for (int count = 0; count < (int)seq.size(); count += 80) out << seq.substr(count, std::min(80, (int)seq.size() - count)) << std::endl;
    return true;
}

// main of utils/JunctionSequence.cpp (:527-557); 0 = done, 3 = the reference would have aborted on an assert
inline int Run(const std::string& BEDPEfile, const std::string& Input_Chim_BAM, const std::string& FAfile, const std::string& OUTPrefix) {
    Params P;  // the utility's own globals (:520-524): Phred33, 10, 4, 1
    std::map<std::string, int> RefTable;
    std::vector<std::string> RefName;
    std::vector<int> RefLength;
    SBamrecord_t Chimrecord;
    BuildRefName(Input_Chim_BAM, RefName, RefTable, RefLength);
    BuildChimericSBamRecord(Chimrecord, Input_Chim_BAM, P);
    std::vector<SV_t> SVs;
    ReadBEDPE(BEDPEfile, RefTable, SVs);
    std::vector<std::vector<SV_t>> ReadSVs, AltSVs;
    FindReadSupport(Chimrecord, SVs, ReadSVs);
    std::vector<int> NumSupports;
    std::vector<bool> flags;
    if (!ExactSequence(SVs, ReadSVs, NumSupports, AltSVs, flags)) return 3;
    std::vector<std::string> Genome;
    ReadGenome(FAfile, Genome, RefTable);
    {
        std::ofstream out(OUTPrefix + "_junc_precise.fa");
        for (size_t i = 0; i < SVs.size(); i++) if (flags[i] && !WriteOne(out, ">squid_" + std::to_string(i), SVs[i], Genome, RefName, " " + std::to_string(NumSupports[i]))) return 3;
    }
    {
        std::ofstream out(OUTPrefix + "_junc_relax.fa");
        for (size_t i = 0; i < SVs.size(); i++) {
            SV_t tmp = SVs[i];
            if ((int)Genome[tmp.BP1.Chr].size() < tmp.BP1.EndPos || (int)Genome[tmp.BP2.Chr].size() < tmp.BP2.EndPos) return 3;
            if (flags[i]) {
                if (tmp.BP1.IsLeft) tmp.BP1.EndPos = std::min(tmp.BP1.EndPos + 1000, (int)Genome[tmp.BP1.Chr].size()); else tmp.BP1.StartPos = std::max(0, tmp.BP1.StartPos - 1000);
                if (tmp.BP2.IsLeft) tmp.BP2.EndPos = std::min(tmp.BP2.EndPos + 1000, (int)Genome[tmp.BP2.Chr].size()); else tmp.BP2.StartPos = std::max(0, tmp.BP2.StartPos - 1000);
            }
            if (!WriteOne(out, ">squid_" + std::to_string(i), tmp, Genome, RefName, "")) return 3;
        }
    }
    {
        std::ofstream out(OUTPrefix + "_junc_alt.fa");
        for (size_t i = 0; i < AltSVs.size(); i++)
            for (size_t j = 0; j < AltSVs[i].size(); j++)
                if (!WriteOne(out, ">squid_" + std::to_string(i) + "_alt_" + std::to_string(j + 1), AltSVs[i][j], Genome, RefName, " " + std::to_string(NumSupports[i]))) return 3;
    }
    return 0;
}

}  // namespace junction
}  // namespace oracle
