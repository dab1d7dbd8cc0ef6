// TEST INFRASTRUCTURE (CPU oracle) -- never linked into, imported or executed by the product.
// Restatement of what squid does with the component orders after Ordering() (src/main.cpp:45-53,67-72): the components are sorted,
// singleton and consecutive runs are merged into them, everything is concatenated per chromosome -- only the optional outputs -TO
// (`_component.txt`) and -RG (`_genome.fa`) see the result; `_sv.txt` is invariant to it.
//   SortComponents           src/SegmentGraph.cpp:4010-4041
//   MergeSingleton           src/SegmentGraph.cpp:4043-4137
//   MergeSingleton_Insert    src/SegmentGraph.cpp:4139-4294 (single nodes), :4296-4423 (consecutive runs)
//   MergeComponents          src/SegmentGraph.cpp:4425-4504
//   BuildRefSeq              src/ReadRec.cpp:285-314
//   OutputNewGenome          src/WriteIO.cpp:172-209, ReverseComplement src/SegmentGraph.cpp:9-13
// Parity unpinned (the reference does not build here, oracle/squid_oracle.cpp header).  Two reads of the reference are undefined
// behaviour and are made deterministic here (and in the product, which follows the same rule):
//   * `flagsmall` / `flaglarge` of MergeSingleton_Insert are read uninitialised when both distances keep their initial value
//     vNodes.size() and 2 * vNodes.size() < 50 (:4167-4190): they start as false;
//   * `Consecutive[idxconsecutive].size()` is evaluated with idxconsecutive == Consecutive.size() (:4064,4077); the value is only used
//     under idxconsecutive < Consecutive.size(): the read is skipped.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace opost {

typedef std::vector<std::vector<int>> Comps;
struct NodeGeo { int Chr, Position, Length; };  // what the post-processing looks at of a node

// :4010-4041
inline Comps SortComponents(const Comps& Components) {
    std::map<int, int> Median_ID;
    std::vector<int> Median(Components.size(), 0);
    for (size_t i = 0; i < Components.size(); i++) {
        std::vector<int> tmp = Components[i];
        for (size_t j = 0; j < tmp.size(); j++) if (tmp[j] < 0) tmp[j] = -tmp[j];
        std::sort(tmp.begin(), tmp.end());
        Median[i] = tmp[(tmp.size() - 1) / 2];
        Median_ID[Median[i]] = (int)i;
    }
    std::sort(Median.begin(), Median.end());
    Comps NewComponents(Components.size());
    for (size_t i = 0; i < Median.size(); i++) {
        NewComponents[i] = Components[Median_ID[Median[i]]];
        std::vector<int>& c = NewComponents[i];
        if (c.size() == 1 && c[0] < 0) c[0] = -c[0];
        int count = 0;
        for (size_t j = 0; j + 1 < c.size(); j++) if (std::abs(c[j]) > std::abs(c[j + 1])) count++;
        if (count > (int)c.size() / 2 || (count == (int)c.size() / 2 && std::abs(c.front()) > std::abs(c.back()))) {
            for (size_t j = 0; j < c.size(); j++) c[j] = -c[j];
            std::reverse(c.begin(), c.end());
        }
    }
    return NewComponents;
}

// :4139-4294
inline void MergeSingleton_Insert(const std::vector<NodeGeo>& vNodes, std::vector<int> SingletonComponent, Comps& NewComponents) {
    const int NV = (int)vNodes.size();
    std::vector<int> Median(NewComponents.size(), 0);
    for (size_t i = 0; i < NewComponents.size(); i++) {
        std::vector<int> tmp;
        for (int v : NewComponents[i]) tmp.push_back(std::abs(v));
        std::sort(tmp.begin(), tmp.end());
        Median[i] = tmp[((int)tmp.size() - 1) / 2];
    }
    typedef std::tuple<int, int, bool> InsertionPlace_t;
    std::vector<std::vector<InsertionPlace_t>> InsertionComponents(NewComponents.size());
    std::vector<int> UnInserted;
    for (size_t i = 0; i < SingletonComponent.size(); i++) {
        const int S = SingletonComponent[i];
        int diffmedian1 = NV, diffmedian2 = NV, diffadja = 50;
        int Idxadja = -1, Idxmedian = -1, eleadja = 0;
        for (size_t j = 0; j < NewComponents.size(); j++) {
            const std::vector<int>& C = NewComponents[j];
            for (int k = 0; k < (int)C.size() - 1; k++) {
                int diffsmall = NV, difflarge = NV;
                bool flagsmall = false, flaglarge = false;  // (uninitialised in the reference, see the header)
                for (int l = std::max(0, k - 1); l <= k; l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[std::abs(S) - 1].Chr && std::abs(C[l]) < std::abs(S) && std::abs(S) - std::abs(C[l]) < diffsmall) {
                        diffsmall = std::abs(S) - std::abs(C[l]); flagsmall = C[l] < 0;
                    }
                for (int l = k + 1; l < std::min((int)C.size(), k + 3); l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[std::abs(S) - 1].Chr && std::abs(C[l]) > std::abs(S) && std::abs(C[l]) - std::abs(S) < difflarge) {
                        difflarge = std::abs(C[l]) - std::abs(S); flaglarge = C[l] < 0;
                    }
                if (diffsmall + difflarge < std::abs(diffadja) && !(flagsmall && flaglarge)) { diffadja = diffsmall + difflarge; Idxadja = (int)j; eleadja = k; }
                diffsmall = NV; difflarge = NV;
                for (int l = std::max(0, k - 1); l <= k; l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[std::abs(S) - 1].Chr && std::abs(C[l]) > std::abs(S) && std::abs(C[l]) - std::abs(S) < difflarge) {
                        difflarge = std::abs(C[l]) - std::abs(S); flaglarge = C[l] > 0;
                    }
                for (int l = k + 1; l < std::min((int)C.size(), k + 3); l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[std::abs(S) - 1].Chr && std::abs(C[l]) < std::abs(S) && std::abs(S) - std::abs(C[l]) < diffsmall) {
                        diffsmall = std::abs(S) - std::abs(C[l]); flagsmall = C[l] > 0;
                    }
                if (diffsmall + difflarge < std::abs(diffadja) && !(flagsmall && flaglarge)) { diffadja = -(diffsmall + difflarge); Idxadja = (int)j; eleadja = k; }
            }
            if (vNodes[Median[j] - 1].Chr == vNodes[S - 1].Chr && std::abs(Median[j] - std::abs(S)) < diffmedian1) {
                for (size_t k = 0; k < C.size(); k++)
                    if (std::abs(std::abs(C[k]) - std::abs(S)) < std::abs(diffmedian2)) {
                        diffmedian2 = std::abs(C[k]) - std::abs(S); diffmedian1 = std::abs(Median[j] - std::abs(S));
                        Idxmedian = (int)j;
                    }
            }
        }
        if ((Idxadja == Idxmedian && Idxadja != -1) || (std::abs(diffadja) < std::abs(diffmedian2) && Idxadja != -1))
            InsertionComponents[Idxadja].push_back(std::make_tuple(std::abs(S), eleadja + 1, diffadja > 0));
        else if (Idxmedian != -1) {
            if (diffmedian2 < 0) InsertionComponents[Idxmedian].push_back(std::make_tuple(std::abs(S), (int)NewComponents[Idxmedian].size(), true));
            else if (diffmedian2 > 0) InsertionComponents[Idxmedian].push_back(std::make_tuple(std::abs(S), 0, true));
        } else UnInserted.push_back(std::abs(S));
    }
    Comps tmpNewComponents;
    for (size_t i = 0; i < InsertionComponents.size(); i++) {
        std::vector<InsertionPlace_t>& I = InsertionComponents[i];
        std::sort(I.begin(), I.end(), [](InsertionPlace_t a, InsertionPlace_t b) { if (std::get<1>(a) != std::get<1>(b)) return std::get<1>(a) < std::get<1>(b); else return std::get<0>(a) < std::get<0>(b); });
        std::vector<int> tmp;
        size_t j = 0;
        auto flush = [&](bool upto, int k) {
            std::vector<int> tmp1;
            size_t count = 0;
            for (; j < I.size() && (!upto || std::get<1>(I[j]) <= k); j++) {
                if (std::get<2>(I[j])) tmp1.push_back(std::get<0>(I[j]));
                else { tmp1.push_back(-std::get<0>(I[j])); count++; }
            }
            if (count > tmp1.size() / 2) std::reverse(tmp1.begin(), tmp1.end());
            tmp.insert(tmp.end(), tmp1.begin(), tmp1.end());
        };
        for (int k = 0; k < (int)NewComponents[i].size(); k++) {
            if (j >= I.size() || k < std::get<1>(I[j])) tmp.push_back(NewComponents[i][k]);
            else { flush(true, k); tmp.push_back(NewComponents[i][k]); }
        }
        if (j < I.size()) flush(false, 0);
        tmpNewComponents.push_back(tmp);
    }
    NewComponents = tmpNewComponents;
    for (int u : UnInserted) NewComponents.push_back(std::vector<int>(1, std::abs(u)));
}

// :4296-4423
inline void MergeSingleton_Insert(const std::vector<NodeGeo>& vNodes, Comps Consecutive, Comps& NewComponents) {
    const int NV = (int)vNodes.size();
    auto median_of = [](const std::vector<int>& c) {
        std::vector<int> tmp;
        for (int v : c) tmp.push_back(std::abs(v));
        std::sort(tmp.begin(), tmp.end());
        return tmp[((int)tmp.size() - 1) / 2];
    };
    std::vector<int> Median(NewComponents.size(), 0), ConsecutiveMedian(Consecutive.size(), 0);
    for (size_t i = 0; i < NewComponents.size(); i++) Median[i] = median_of(NewComponents[i]);
    for (size_t i = 0; i < Consecutive.size(); i++) ConsecutiveMedian[i] = median_of(Consecutive[i]);
    typedef std::tuple<std::vector<int>, int, bool> InsertionPlace_t;
    std::vector<std::vector<InsertionPlace_t>> InsertionComponents(NewComponents.size());
    Comps UnInserted;
    for (size_t i = 0; i < Consecutive.size(); i++) {
        const int CM = ConsecutiveMedian[i], F = std::abs(Consecutive[i][0]), B = std::abs(Consecutive[i].back());
        int diffmedian1 = NV, diffmedian2 = NV, diffadja = 50;
        int Idxadja = -1, Idxmedian = -1, eleadja = 0;
        for (size_t j = 0; j < NewComponents.size(); j++) {
            const std::vector<int>& C = NewComponents[j];
            for (int k = 0; k < (int)C.size() - 1; k++) {
                int diffsmall = NV, difflarge = NV;
                bool flagsmall = false, flaglarge = false;  // (uninitialised in the reference, see the header)
                for (int l = std::max(0, k - 1); l <= k; l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[CM - 1].Chr && std::abs(C[l]) < F && F - std::abs(C[l]) < diffsmall) { diffsmall = F - std::abs(C[l]); flagsmall = C[l] < 0; }
                for (int l = k + 1; l < std::min((int)C.size(), k + 3); l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[CM - 1].Chr && std::abs(C[l]) > B && std::abs(C[l]) - B < difflarge) { difflarge = std::abs(C[l]) - B; flaglarge = C[l] < 0; }
                if (diffsmall + difflarge < std::abs(diffadja) && !(flagsmall && flaglarge)) { diffadja = diffsmall + difflarge; Idxadja = (int)j; eleadja = k; }
                diffsmall = NV; difflarge = NV;
                for (int l = std::max(0, k - 1); l <= k; l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[CM - 1].Chr && std::abs(C[l]) > B && std::abs(C[l]) - B < difflarge) { difflarge = std::abs(C[l]) - B; flaglarge = C[l] > 0; }
                for (int l = k + 1; l < std::min((int)C.size(), k + 3); l++)
                    if (vNodes[std::abs(C[l]) - 1].Chr == vNodes[CM - 1].Chr && std::abs(C[l]) < F && F - std::abs(C[l]) < diffsmall) { diffsmall = F - std::abs(C[l]); flagsmall = C[l] > 0; }
                if (diffsmall + difflarge < std::abs(diffadja) && !(flagsmall && flaglarge)) { diffadja = -(diffsmall + difflarge); Idxadja = (int)j; eleadja = k; }
            }
            if (vNodes[Median[j] - 1].Chr == vNodes[CM - 1].Chr && std::abs(Median[j] - CM) < diffmedian1) {
                for (size_t k = 0; k < C.size(); k++)
                    if (std::abs(std::abs(C[k]) - CM) < std::abs(diffmedian2)) { diffmedian2 = std::abs(C[k]) - CM; diffmedian1 = std::abs(Median[j] - CM); Idxmedian = (int)j; }
            }
        }
        if ((Idxadja == Idxmedian && Idxadja != -1) || (std::abs(diffadja) < std::abs(diffmedian2) && Idxadja != -1))
            InsertionComponents[Idxadja].push_back(std::make_tuple(Consecutive[i], eleadja + 1, diffadja > 0));
        else if (Idxmedian != -1) {
            if (diffmedian2 < 0) InsertionComponents[Idxmedian].push_back(std::make_tuple(Consecutive[i], (int)NewComponents[Idxmedian].size(), true));
            else InsertionComponents[Idxmedian].push_back(std::make_tuple(Consecutive[i], 0, true));
        } else UnInserted.push_back(Consecutive[i]);
    }
    Comps tmpNewComponents;
    for (size_t i = 0; i < InsertionComponents.size(); i++) {
        std::vector<InsertionPlace_t>& I = InsertionComponents[i];
        std::sort(I.begin(), I.end(), [](InsertionPlace_t a, InsertionPlace_t b) { if (std::get<1>(a) != std::get<1>(b)) return std::get<1>(a) < std::get<1>(b); else return std::get<0>(a).front() < std::get<0>(b).front(); });
        std::vector<int> tmp;
        size_t j = 0;
        auto flush = [&](bool upto, int k) {
            std::vector<int> tmp1;
            for (; j < I.size() && (!upto || std::get<1>(I[j]) <= k); j++) {
                const std::vector<int>& cur = std::get<0>(I[j]);
                if (std::get<2>(I[j])) tmp1.insert(tmp1.end(), cur.begin(), cur.end());
                else {
                    std::vector<int> rev(cur.size());
                    for (size_t l = 0; l < cur.size(); l++) rev[l] = -cur[cur.size() - 1 - l];
                    tmp1.insert(tmp1.begin(), rev.begin(), rev.end());
                }
            }
            tmp.insert(tmp.end(), tmp1.begin(), tmp1.end());
        };
        for (int k = 0; k < (int)NewComponents[i].size(); k++) {
            if (j >= I.size() || k < std::get<1>(I[j])) tmp.push_back(NewComponents[i][k]);
            else { flush(true, k); tmp.push_back(NewComponents[i][k]); }
        }
        if (j < I.size()) flush(false, 0);
        tmpNewComponents.push_back(tmp);
    }
    NewComponents = tmpNewComponents;
    for (const std::vector<int>& u : UnInserted) NewComponents.push_back(u);
}

// :4043-4137
inline Comps MergeSingleton(const std::vector<NodeGeo>& vNodes, const Comps& Components, const std::vector<int>& RefLength) {
    Comps NewComponents, Consecutive;
    std::vector<int> SingletonComponent, tmp;
    auto chr = [&](int v) { return vNodes[std::abs(v) - 1].Chr; };
    auto whole_chr = [&](const std::vector<int>& c) {
        const NodeGeo &f = vNodes[std::abs(c.front()) - 1], &b = vNodes[std::abs(c.back()) - 1];
        return f.Position == 0 && b.Position + b.Length == RefLength[f.Chr];
    };
    auto is_consecutive = [&](const std::vector<int>& c) {
        for (size_t j = 0; j + 1 < c.size(); j++) if (c[j + 1] - c[j] != 1 || chr(c[j + 1]) != chr(c[j])) return false;
        return true;
    };
    for (size_t i = 0; i < Components.size(); i++)
        if (Components[i].size() != 1) {
            bool isconsecutive = is_consecutive(Components[i]);
            if (isconsecutive && whole_chr(Components[i])) isconsecutive = false;
            if (!isconsecutive) NewComponents.push_back(Components[i]); else Consecutive.push_back(Components[i]);
        }
    size_t idxconsecutive = 0;
    for (size_t i = 0; i < Components.size(); i++) {
        if (Components[i].size() != 1) continue;
        const int v = Components[i][0];
        const NodeGeo& nv = vNodes[v - 1];
        const bool whole = nv.Position == 0 && nv.Length == RefLength[nv.Chr];
        if (whole) { NewComponents.push_back(Components[i]); continue; }
        if (tmp.size() == 0 || (tmp.back() + 1 == v && vNodes[tmp.back() - 1].Chr == chr(v))) { tmp.push_back(std::abs(v)); continue; }
        const int key_chr = vNodes[tmp[(tmp.size() - 1) / 2] - 1].Chr;  // (tmp.size() == 1: the node itself)
        for (; idxconsecutive < Consecutive.size() && Consecutive[idxconsecutive].back() + 1 <= tmp[0]; idxconsecutive++)
            if (Consecutive[idxconsecutive].back() + 1 >= tmp[0] && vNodes[Consecutive[idxconsecutive][(Consecutive[idxconsecutive].size() - 1) / 2] - 1].Chr == key_chr) break;
        const bool have = Consecutive.size() != 0 && idxconsecutive < Consecutive.size();
        const int medianidx = have ? (int)(Consecutive[idxconsecutive].size() - 1) / 2 : -1;  // (see the header: the reference reads past the end here)
        if (tmp.size() == 1) {
            if (have && tmp[0] == Consecutive[idxconsecutive].front() - 1 && vNodes[tmp[0] - 1].Chr == vNodes[Consecutive[idxconsecutive][medianidx] - 1].Chr)
                Consecutive[idxconsecutive].insert(Consecutive[idxconsecutive].begin(), tmp[0]);
            else if (have && tmp[0] == Consecutive[idxconsecutive].back() + 1 && vNodes[tmp[0] - 1].Chr == vNodes[Consecutive[idxconsecutive][medianidx] - 1].Chr)
                Consecutive[idxconsecutive].push_back(tmp[0]);
            else SingletonComponent.push_back(tmp[0]);
        } else {
            if (have && tmp.back() == Consecutive[idxconsecutive].front() - 1 && key_chr == vNodes[Consecutive[idxconsecutive][medianidx] - 1].Chr)
                Consecutive[idxconsecutive].insert(Consecutive[idxconsecutive].begin(), tmp.begin(), tmp.end());
            else if (have && tmp[0] == Consecutive[idxconsecutive].back() + 1 && key_chr == vNodes[Consecutive[idxconsecutive][medianidx] - 1].Chr)
                Consecutive[idxconsecutive].insert(Consecutive[idxconsecutive].end(), tmp.begin(), tmp.end());
            else Consecutive.push_back(tmp);
        }
        tmp.clear(); tmp.push_back(std::abs(v));
    }
    if (tmp.size() > 1) Consecutive.push_back(tmp);
    else if (tmp.size() == 1) SingletonComponent.push_back(tmp[0]);
    MergeSingleton_Insert(vNodes, SingletonComponent, NewComponents);
    // push back new consecutive nodes after singleton insertion (:4098-4133)
    Comps tmpConsecutive, tmpNewComponents;
    idxconsecutive = 0;
    for (size_t i = 0; i < NewComponents.size(); i++) {
        bool isconsecutive = is_consecutive(NewComponents[i]);
        if (isconsecutive && whole_chr(NewComponents[i])) isconsecutive = false;
        if (!isconsecutive || NewComponents[i].size() == 1) tmpNewComponents.push_back(NewComponents[i]);
        else {
            const size_t lastidx = idxconsecutive;
            for (; idxconsecutive < Consecutive.size() && Consecutive[idxconsecutive].back() < NewComponents[i].front(); idxconsecutive++) {}
            for (size_t j = lastidx; j < idxconsecutive; j++) tmpConsecutive.push_back(Consecutive[j]);
            tmpConsecutive.push_back(NewComponents[i]);
        }
    }
    for (size_t j = idxconsecutive; j < Consecutive.size(); j++) tmpConsecutive.push_back(Consecutive[j]);
    Consecutive = tmpConsecutive; tmpConsecutive.clear();
    NewComponents = tmpNewComponents;
    for (size_t i = 0; i < Consecutive.size(); i++) {
        if (tmpConsecutive.size() == 0 || tmpConsecutive.back().back() + 1 != Consecutive[i].front() || chr(tmpConsecutive.back().back()) != chr(Consecutive[i].back()))
            tmpConsecutive.push_back(Consecutive[i]);
        else tmpConsecutive.back().insert(tmpConsecutive.back().end(), Consecutive[i].begin(), Consecutive[i].end());
    }
    Consecutive = tmpConsecutive;
    MergeSingleton_Insert(vNodes, Consecutive, NewComponents);
    return NewComponents;
}

// :4425-4504 (main calls it without the second argument: LenCutOff = 5, src/SegmentGraph.h:118)
inline Comps MergeComponents(const std::vector<NodeGeo>& vNodes, const Comps& Components, int LenCutOff = 5) {
    std::vector<int> ChromoMargin;
    for (size_t i = 0; i + 1 < vNodes.size(); i++) if (vNodes[i].Chr != vNodes[i + 1].Chr) ChromoMargin.push_back((int)i + 1);
    Comps NewComponents;
    auto median_of = [](const std::vector<int>& c) {
        std::vector<int> tmp = c;
        for (int& v : tmp) if (v < 0) v = -v;
        std::sort(tmp.begin(), tmp.end());
        return tmp[(tmp.size() - 1) / 2];
    };
    for (size_t i = 0; i < Components.size(); i++) {
        if (NewComponents.size() == 0) { NewComponents.push_back(Components[i]); continue; }
        int curLen = 0;
        for (int v : Components[i]) curLen += vNodes[std::abs(v) - 1].Length;
        const int curMedian = median_of(Components[i]);
        std::vector<int> reversecomponent;
        for (size_t k = 0; k < Components[i].size(); k++) reversecomponent.push_back(-Components[i][Components[i].size() - 1 - k]);
        std::vector<int> Median(NewComponents.size(), 0);
        for (size_t j = 0; j < NewComponents.size(); j++) Median[j] = median_of(NewComponents[j]);
        size_t plusidx = NewComponents.size(), minusidx = NewComponents.size();
        std::ptrdiff_t pluspos = 0, minuspos = 0;  // positions inside NewComponents[plusidx] / [minusidx] (the reference keeps iterators)
        int ind = 0, diff = std::abs(curMedian - Median[0]) + 1;
        for (size_t j = 0; j < Median.size(); j++)
            if (std::abs(Median[j] - curMedian) < diff) {
                for (size_t q = 0; q < NewComponents[j].size(); q++) {
                    const int e = NewComponents[j][q];
                    if (std::abs(e) == std::abs(Components[i].front()) - 1) { minuspos = (std::ptrdiff_t)q; minusidx = j; }
                    else if (std::abs(e) == std::abs(Components[j].back()) + 1) { pluspos = (std::ptrdiff_t)q; plusidx = j; }  // Components[j] (not [i]): ledger B20
                }
                diff = std::abs(Median[j] - curMedian);
                ind = (int)j;
            }
        size_t j;
        for (j = 0; j < ChromoMargin.size(); j++)
            if ((Median[ind] <= ChromoMargin[j] && curMedian > ChromoMargin[j]) || (Median[ind] > ChromoMargin[j] && curMedian <= ChromoMargin[j])) break;
        const bool both = plusidx != NewComponents.size() && minusidx != NewComponents.size() && plusidx == minusidx;
        if (j != ChromoMargin.size()) NewComponents.push_back(Components[i]);
        else if (curLen < LenCutOff && both && minuspos - pluspos == 1 && !(NewComponents[plusidx][pluspos] > 0 && NewComponents[minusidx][minuspos] > 0))
            NewComponents[minusidx].insert(NewComponents[minusidx].begin() + minuspos, reversecomponent.begin(), reversecomponent.end());
        else if (curLen < LenCutOff && both && minuspos - pluspos == -1 && !(NewComponents[plusidx][pluspos] < 0 && NewComponents[minusidx][minuspos] < 0))
            NewComponents[plusidx].insert(NewComponents[plusidx].begin() + pluspos, Components[i].begin(), Components[i].end());
        else NewComponents[ind].insert(NewComponents[ind].end(), Components[i].begin(), Components[i].end());
    }
    return NewComponents;
}

// src/ReadRec.cpp:285-314 (boost::split on tab / blank; the first token without its '>' names the sequence)
inline bool BuildRefSeq(const std::string& fafile, const std::map<std::string, int>& RefTable, const std::vector<int>& RefLength, std::vector<std::string>& RefSequence) {
    std::ifstream input(fafile);
    RefSequence.assign(RefTable.size(), std::string());
    std::string line;
    int ind = -1;
    while (std::getline(input, line)) {
        if (!line.empty() && line[0] == '>') {
            const size_t e = line.find_first_of("\t ");
            std::map<std::string, int>::const_iterator cit = RefTable.find(line.substr(1, e == std::string::npos ? std::string::npos : e - 1));
            ind = cit != RefTable.end() ? cit->second : -1;
        } else if (ind != -1) RefSequence[ind] += line;
    }
    for (size_t i = 0; i < RefSequence.size(); i++)
        if ((int)RefSequence[i].size() != RefLength[i]) { RefSequence.clear(); return false; }
    return true;
}

// src/SegmentGraph.cpp:9-13 with the table of src/SegmentGraph.h:51 (a character outside it becomes '\0': std::map::operator[])
inline void ReverseComplement(std::string& s) {
    static const std::map<char, char> N = {{'A', 'T'}, {'C', 'G'}, {'G', 'C'}, {'T', 'A'}, {'R', 'Y'}, {'Y', 'R'}, {'S', 'W'}, {'W', 'S'}, {'K', 'M'}, {'M', 'K'}, {'B', 'V'}, {'V', 'B'},
                                           {'D', 'H'}, {'H', 'D'}, {'N', 'N'}, {'.', '.'}, {'-', '-'}};
    for (char& ch : s) { std::map<char, char>::const_iterator it = N.find((char)std::toupper((unsigned char)ch)); ch = it == N.end() ? '\0' : it->second; }
    std::reverse(s.begin(), s.end());
}

// src/WriteIO.cpp:172-209
inline void OutputNewGenome(const std::vector<NodeGeo>& vNodes, const Comps& Components, const std::vector<std::string>& RefSequence, const std::vector<std::string>& RefName, const std::string& outputfile) {
    std::ofstream output(outputfile);
    for (size_t i = 0; i < Components.size(); i++) {
        const std::vector<int>& C = Components[i];
        std::string info = "PA:", seq, tmpseq;
        for (size_t j = 0; j < C.size(); j++) {
            size_t k;
            for (k = j + 1; k < C.size() && C[k] - C[k - 1] == 1 && vNodes[std::abs(C[j]) - 1].Chr == vNodes[std::abs(C[k]) - 1].Chr; k++) {}
            const NodeGeo &a = vNodes[std::abs(C[j]) - 1], &b = vNodes[std::abs(C[k - 1]) - 1];
            int curChr, curStart, curLen;
            if (C[j] > 0) { curChr = a.Chr; curStart = a.Position; curLen = b.Position + b.Length - a.Position; }
            else { curChr = b.Chr; curStart = b.Position; curLen = a.Position + a.Length - b.Position; }
            tmpseq = RefSequence[curChr].substr(curStart, curLen);
            info += "{" + RefName[curChr] + "," + std::to_string(curStart) + "," + std::to_string(curLen) + "}";
            if (C[j] < 0) ReverseComplement(tmpseq);
            seq += tmpseq;
            info += C[j] < 0 ? "R-" : "F-";
            j = k - 1;
        }
        info = info.substr(0, info.size() - 1);
        output << ">chr" << (i + 1) << '\t' << "LN:" << seq.size() << '\t' << info << std::endl;
        for (size_t idx = 0; idx < seq.size(); idx += 80) output << seq.substr(idx, std::min<size_t>(80, seq.size() - idx)) << std::endl;
    }
}

}  // namespace opost
