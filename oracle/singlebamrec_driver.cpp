// Pin of the aligned-block type.  ONE driver, built twice by oracle/Makefile:
//   -DUSE_REFERENCE : against /root/reference/src/SingleBamRec.h where it lies (std-only includes; never copied)
//                     -> oracle/_ref/ref_singlebamrec   (the REAL reference type)
//   default         : against oracle/o_readrec.h's restatement -> build/oracle_singlebamrec
// Reads blocks "RefID RefPos ReadPos MatchRef MatchRead IsReverse IsFirstRead" (one per line, at most 256) from stdin and
// prints, one line each: the five pairwise relations of src/SingleBamRec.h:39-58 as n*n 0/1 strings and the
// permutations libstdc++'s std::sort produces with operator< (SegmentGraph.cpp:264) and with CompReadPos
// (ReadRec.cpp:144-145).  MapQual, which takes part in no comparison, carries the input index.
// Test infrastructure: tests/test_ref_pin.py compares the two binaries and the product's comparators (sq_debug_blocks).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#ifdef USE_REFERENCE
#include "SingleBamRec.h"
typedef SingleBamRec_t Rec;
#else
#include "o_readrec.h"
typedef oracle::SingleBamRec_t Rec;
#endif

int main() {
    std::vector<Rec> v;
    int a, b, c, d, e, f, g;
    while (std::scanf("%d %d %d %d %d %d %d", &a, &b, &c, &d, &e, &f, &g) == 7 && v.size() < 256)
        v.push_back(Rec(a, b, c, d, e, (uint8_t)v.size(), f != 0, g != 0));
    const size_t n = v.size();
    std::string lt, gt, eq, same, rp;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            lt += v[i] < v[j] ? '1' : '0';
            gt += v[i] > v[j] ? '1' : '0';
#ifdef USE_REFERENCE
            eq += v[i] == v[j] ? '1' : '0';
#else
            eq += (v[i].RefID == v[j].RefID && v[i].RefPos == v[j].RefPos) ? '1' : '0';  // (the restatement has no operator==: nothing on the path calls it)
#endif
            same += v[i].Same(v[j]) ? '1' : '0';
            rp += Rec::CompReadPos(v[i], v[j]) ? '1' : '0';
        }
    std::printf("lt %s\ngt %s\neq %s\nsame %s\nreadpos %s\n", lt.c_str(), gt.c_str(), eq.c_str(), same.c_str(), rp.c_str());
    std::vector<Rec> s = v;
    std::sort(s.begin(), s.end());
    std::printf("sort_pos");
    for (const Rec& r : s) std::printf(" %d", (int)r.MapQual);
    s = v;
    std::sort(s.begin(), s.end(), Rec::CompReadPos);
    std::printf("\nsort_readpos");
    for (const Rec& r : s) std::printf(" %d", (int)r.MapQual);
    std::printf("\n");
    return 0;
}
