// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED: squid v1.5 cannot be built in this image (BamTools / GLPK / Boost are absent and may not be
// stubbed), and the reference ships no tests, fixtures or golden vectors.  This program is a single-threaded
// CPU restatement of the reference's `squid` executable for the STAR path (-b/-c/-o), used
//   * by tests/ as the checker for the HIP path (stage dumps + `_sv.txt`), and
//   * by bench.py's `cpu_baseline` leg (kind "port", 1 core).
// It follows src/main.cpp:17-76, src/Config.cpp:80-230, src/WriteIO.cpp:33-124 and the files named in
// o_readrec.h / o_graph.h / o_order.h / o_post.h.  Out of scope in the oracle: --bwa.
// SortComponents/MergeSingleton/MergeComponents (src/main.cpp:45-48, restated in o_post.h) run only with -TO / -RG: `_sv.txt`
// depends on a component only through the relative rank and sign of an edge's two end nodes (src/WriteIO.cpp:57-63), both
// ends of an edge always share a connected component, and those three steps only move, reverse-and-negate or
// interleave whole components (src/SegmentGraph.cpp:4034-4038,4236-4253,4400-4403,4457-4459,4494-4498).
#include <chrono>
#include <fstream>
#include <sstream>

#include "o_order.h"
#include "o_bwa.h"
#include "o_junction.h"
#include "o_post.h"

using namespace oracle;

namespace oracle {
struct StageSink {
    std::string dir;
    void nodes(const std::string& name, const std::vector<Node_t>& N, const std::vector<int>* label = nullptr) const {
        if (dir.empty()) return;
        std::ofstream o(dir + "/" + name);
        o << "# chr\tpos\tlen\tsupport\tavgdepth_hexfloat" << (label ? "\tlabel" : "") << "\n";
        char buf[64];
        for (size_t i = 0; i < N.size(); i++) {
            std::snprintf(buf, sizeof buf, "%a", N[i].AvgDepth);
            o << N[i].Chr << '\t' << N[i].Position << '\t' << N[i].Length << '\t' << N[i].Support << '\t' << buf;
            if (label) o << '\t' << (*label)[i];
            o << '\n';
        }
    }
    void edges(const std::string& name, const std::vector<Edge_t>& E, const std::vector<bool>* keep = nullptr) const {
        if (dir.empty()) return;
        std::ofstream o(dir + "/" + name);
        o << "# ind1\thead1\tind2\thead2\tweight\tgroupweight" << (keep ? "\tkeep" : "") << "\n";
        for (size_t i = 0; i < E.size(); i++) {
            o << E[i].Ind1 << '\t' << (int)E[i].Head1 << '\t' << E[i].Ind2 << '\t' << (int)E[i].Head2 << '\t' << E[i].Weight << '\t' << E[i].GroupWeight;
            if (keep) o << '\t' << (int)(*keep)[i];
            o << '\n';
        }
    }
};

// src/SegmentGraph.cpp:104-124
void SegmentGraph_t::Construct(const std::vector<int>& RefLength, SBamrecord_t& Chimrecord, const std::string& bamfile, StageSink* sink, uint16_t* ReadLen) {
    if (P.UsingSTAR) BuildNode_STAR(RefLength, Chimrecord, bamfile);
    else BuildNode_BWA(RefLength, bamfile, *ReadLen);
    if (sink) { sink->nodes("nodes_seed.txt", seedNodes); sink->nodes("nodes_build.txt", vNodes); }
    BuildEdges(Chimrecord, bamfile);
    if (sink) sink->edges("edges_build.txt", vEdges);
    FilterbyWeight();
    if (sink) sink->edges("edges_weight.txt", vEdges);
    std::vector<bool> KeepEdge;
    FilterbyInterleaving(KeepEdge);
    if (sink) sink->edges("edges_interleave.txt", vEdges, &KeepEdge);
    FilterEdges(KeepEdge);
    if (sink) sink->edges("edges_filter.txt", vEdges);
    CompressNode();
    if (sink) { sink->nodes("nodes_compress.txt", vNodes); sink->edges("edges_compress.txt", vEdges); }
    FurtherCompressNode();
    ConnectedComponent();
    MultiplyDisEdges();
    if (sink) { sink->nodes("nodes_final.txt", vNodes, &Label); sink->edges("edges_final.txt", vEdges); }
    std::cout << vNodes.size() << '\t' << vEdges.size() << std::endl;
}
}  // namespace oracle

// src/Config.cpp:80-230 (flag-for-flag, including ledger B1-B3)
static bool parse_arguments(int argc, char* argv[], Params& P, std::string& dumpdir) {
    bool success = true, specify_mq = false;
    auto bool01 = [&](const char* v, bool& dst) { if (std::string(v) == "0") dst = 0; else if (std::string(v) == "1") dst = 1; else success = false; };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--help") { std::printf("squid_oracle: CPU restatement of squid v1.5 (test infrastructure)\n"); std::exit(0); }
        if (a == "--version") { std::printf("v1.5\n"); std::exit(0); }
        bool hasnext = i < argc - 1;
        if (a == "-b" && hasnext) P.Input_BAM = argv[i + 1];
        if (a == "-o" && hasnext) P.Output_Prefix = argv[i + 1];
        if (a == "--bwa") P.UsingSTAR = false;
        if (a == "-c" && hasnext) P.Input_Chim_BAM = argv[i + 1];
        if (a == "-f" && hasnext) P.Input_FASTA = argv[i + 1];
        if (a == "-pt" && hasnext) bool01(argv[i + 1], P.Phred_Type);
        if (a == "-pl" && hasnext) P.Max_LowPhred_Len = (uint16_t)std::atoi(argv[i + 1]);
        if (a == "-pm" && hasnext) P.Min_Phred = (uint8_t)std::atoi(argv[i + 1]);
        if (a == "-mq" && hasnext) { P.Min_MapQual = (uint16_t)std::atoi(argv[i + 1]); specify_mq = true; }
        if (a == "-dp" && hasnext) P.Concord_Dist_Pos = std::atoi(argv[i + 1]);
        if (a == "-di" && hasnext) P.Concord_Dist_Idx = std::atoi(argv[i + 1]);
        if (a == "-w" && hasnext) P.Min_Edge_Weight = std::atoi(argv[i + 1]);
        if (a == "-r" && hasnext) P.DiscordantRatio = std::atof(argv[i + 1]);
        if (a == "-a" && hasnext) P.MaxAllowedDegree = std::atoi(argv[i + 1]);
        if (a == "-G" && hasnext) bool01(argv[i + 1], P.Print_Graph);
        if (a == "-CO" && hasnext) bool01(argv[i + 1], P.Print_Components_Ordering);
        if (a == "-TO" && hasnext) bool01(argv[i + 1], P.Print_Total_Ordering);
        if (a == "-RG" && hasnext) bool01(argv[i + 1], P.Print_Rearranged_Genome);
        if (a == "--dump" && hasnext) dumpdir = argv[i + 1];  // oracle-only
    }
    if (P.Input_BAM == "" || P.Output_Prefix == "") success = false;
    if (P.Input_FASTA == "" && P.Print_Rearranged_Genome) { std::printf("reference FASTA needed to output rearranged genome sequence.\n"); success = false; }
    if (!specify_mq && P.UsingSTAR) P.Min_MapQual = 255;
    if (P.UsingSTAR && P.Input_Chim_BAM == "") { std::printf("separate chimeric alignment BAM file is required if using STAR aligner.\n"); success = false; }
    if (!success) std::printf("Check your argument.\n");
    return success;
}

// src/WriteIO.cpp:33-43
static void WriteComponents(const std::string& outputfile, const std::vector<std::vector<int>>& Components) {
    std::ofstream output(outputfile, std::ios::out);
    output << "# component_id\tnodes\n";
    for (size_t i = 0; i < Components.size(); i++) {
        output << i << '\t';
        for (size_t j = 0; j + 1 < Components[i].size(); j++) output << Components[i][j] << ",";
        output << Components[i][Components[i].size() - 1] << std::endl;
    }
}

// src/SegmentGraph.cpp:3223-3234
static void OutputGraph(const std::string& outputfile, const SegmentGraph_t& G) {
    std::ofstream output(outputfile, std::ios::out);
    output << "# type=node\tid\tChr\tPosition\tEnd\tSupport\tAvgDepth\tLabel\n";
    output << "# type=edge\tid\tInd1\tHead1\tInd2\tHead2\tWeight\n";
    for (size_t i = 0; i < G.vNodes.size(); i++)
        output << "node\t" << i << '\t' << G.vNodes[i].Chr << '\t' << G.vNodes[i].Position << '\t' << (G.vNodes[i].Position + G.vNodes[i].Length) << '\t' << G.vNodes[i].Support << '\t'
               << G.vNodes[i].AvgDepth << '\t' << G.Label[i] << '\n';
    for (size_t i = 0; i < G.vEdges.size(); i++)
        output << "edge\t" << i << '\t' << G.vEdges[i].Ind1 << '\t' << (G.vEdges[i].Head1 ? "H\t" : "T\t") << G.vEdges[i].Ind2 << '\t' << (G.vEdges[i].Head2 ? "H\t" : "T\t") << G.vEdges[i].Weight
               << std::endl;
}

// src/WriteIO.cpp:45-124
static void WriteBEDPE(const std::string& outputfile, SegmentGraph_t& G, const std::vector<std::vector<int>>& Components, const std::vector<pii>& Node_NewChr,
                       const std::vector<std::string>& RefName, EdgeBPMap& ExactBP, EdgeBPMap& Support, const Params& P) {
    std::sort(G.vEdges.begin(), G.vEdges.end(), [](Edge_t a, Edge_t b) { return a.Weight > b.Weight; });  // unstable: ledger B8
    std::ofstream output(outputfile, std::ios::out);
    output << "# chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tnum_concordantfrag_bp1\tnum_concordantfrag_bp2\n";
    for (size_t i = 0; i < G.vEdges.size(); i++) {
        const Edge_t& e = G.vEdges[i];
        const Node_t &n1 = G.vNodes[e.Ind1], &n2 = G.vNodes[e.Ind2];
        bool flag_chr = (n1.Chr == n2.Chr);
        bool flag_ori = (e.Head1 == false && e.Head2 == true);
        bool flag_dist = (n2.Position - n1.Position - n1.Length <= P.Concord_Dist_Pos || e.Ind2 - e.Ind1 <= P.Concord_Dist_Idx);
        if (flag_chr && flag_ori && flag_dist) continue;
        pii pos1 = Node_NewChr[e.Ind1], pos2 = Node_NewChr[e.Ind2];
        bool flag = false;
        if (pos1.first == pos2.first && pos1.second < pos2.second && e.Head1 == (Components[pos1.first][pos1.second] < 0) && e.Head2 == (Components[pos2.first][pos2.second] > 0)) flag = true;
        else if (pos1.first == pos2.first && pos1.second > pos2.second && e.Head2 == (Components[pos2.first][pos2.second] < 0) && e.Head1 == (Components[pos1.first][pos1.second] > 0)) flag = true;
        if (!flag) continue;
        EdgeBPMap::iterator itmap = ExactBP.find(e);
        EdgeBPMap::const_iterator itsup = Support.find(e);
        std::vector<pii> BP;
        if (itmap == ExactBP.end() || itmap->second.size() == 0) BP.push_back(pii(e.Head1 ? n1.Position : (n1.Position + n1.Length), e.Head2 ? n2.Position : (n2.Position + n2.Length)));
        else BP = itmap->second;
        const std::vector<pii>& Sup = itsup->second;
        for (size_t k = 0; k < BP.size(); k++) {
            output << RefName[n1.Chr] << '\t';
            if (e.Head1) output << BP[k].first << '\t' << (n1.Position + n1.Length) << '\t';
            else output << n1.Position << '\t' << BP[k].first << '\t';
            output << RefName[n2.Chr] << '\t';
            if (e.Head2) output << BP[k].second << '\t' << (n2.Position + n2.Length) << '\t';
            else output << n2.Position << '\t' << BP[k].second << '\t';
            output << ".\t" << e.Weight << "\t" << (e.Head1 ? "-\t" : "+\t") << (e.Head2 ? "-\t" : "+\t") << Sup[k].first << "\t" << Sup[k].second << std::endl;
        }
    }
}

// cross-check of the two exact solvers (enumeration vs branch-and-bound) on random small instances
static int solver_selftest(int rounds) {
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
    for (int r = 0; r < rounds; r++) {
        int n = 2 + rnd() % 6, m = 1 + rnd() % (2 * n);
        std::vector<LocalEdge> E;
        for (int i = 0; i < m; i++) {
            int u = rnd() % n, v = rnd() % n;
            if (u == v) continue;
            if (u > v) std::swap(u, v);
            E.push_back(LocalEdge{u, v, (bool)(rnd() & 1), (bool)(rnd() & 1), 1 + (int)(rnd() % 9), true});
        }
        std::vector<int> o1, o2;
        unsigned m1, m2;
        long v1, v2;
        SolveBrute(n, E, o1, m1, v1, nullptr);
        SolveDP(n, E, o2, m2, v2);
        if (v1 != v2 || m1 != m2 || o1 != o2) {
            std::printf("selftest MISMATCH round %d n=%d: brute val=%ld mask=%u, bnb val=%ld mask=%u\n", r, n, v1, m1, v2, m2);
            return 1;
        }
    }
    std::printf("selftest OK (%d instances)\n", rounds);
    return 0;
}

// one ordering problem from stdin ("n m" then m lines "u v head_u head_v weight", u < v): prints "value s_1 ... s_n" with
// s = +(node + 1) / -(node + 1) left to right.  method: brute (all signed permutations), bnb (n <= 26), wide / wide_seeded (n <= 128),
// kdvalue (the value alone)
static int solve_order_cli(const std::string& method) {
    int n = 0, m = 0;
    if (std::scanf("%d %d", &n, &m) != 2 || n < 2 || n > 128) return 2;
    std::vector<LocalEdge> E;
    for (int i = 0; i < m; i++) {
        int u, v, hu, hv, w;
        if (std::scanf("%d %d %d %d %d", &u, &v, &hu, &hv, &w) != 5 || u < 0 || u >= v || v >= n) return 2;
        E.push_back(LocalEdge{u, v, hu != 0, hv != 0, w, true});
    }
    std::vector<int> order;
    std::vector<bool> rev(n, false);
    long val = -1;
    if (method == "kdvalue") {  // the optimum value alone, by the edge search (KeepDrop)
        KeepDrop kd(n, E, 50000000L);
        const long v = kd.Run();
        if (!kd.complete) { std::printf("FAILED\n"); return 0; }
        std::printf("%ld\n", v);
        return 0;
    }
    if (method == "wide" || method == "wide_seeded") {  // wide: the orientation search from nothing; wide_seeded: from the edge search's value, as Orderer does
        WideSolver ws(n, E, 50000000L);
        if (method == "wide_seeded") { KeepDrop kd(n, E, 5000000L); const long v = kd.Run(); if (v > 0) ws.bestval = v - 1; }
        ws.Run();
        if (!ws.ok) { std::printf("FAILED\n"); return 0; }
        order = ws.bestorder; rev = ws.bestrev; val = ws.bestval;
    } else {
        unsigned mask = 0;
        if (method == "brute") { if (n > 9) return 2; SolveBrute(n, E, order, mask, val, nullptr); }
        else { if (n > 26) return 2; SolveDP(n, E, order, mask, val); }
        for (int i = 0; i < n; i++) rev[i] = (mask >> i) & 1;
    }
    std::printf("%ld", val);
    for (int p = 0; p < n; p++) std::printf(" %d", rev[order[p]] ? -(order[p] + 1) : order[p] + 1);
    std::printf("\n");
    return 0;
}

int main(int argc, char* argv[]) {
    if (argc >= 2 && std::string(argv[1]) == "--selftest") return solver_selftest(argc >= 3 ? std::atoi(argv[2]) : 2000);
    if (argc >= 3 && std::string(argv[1]) == "--solve-order") return solve_order_cli(argv[2]);
    if (argc >= 6 && std::string(argv[1]) == "--junction") return oracle::junction::Run(argv[2], argv[3], argv[4], argv[5]);  // utils/JunctionSequence.cpp
    Params P;
    std::string dumpdir;
    if (argc >= 2 && std::string(argv[1]) == "--print-config") {  // same line as oracle/ref_config_driver.cpp prints for the real parser
        bool ok = parse_arguments(argc - 1, argv + 1, P, dumpdir);
        std::printf("ok=%d star=%d pt=%d pl=%d pm=%d mq=%d dp=%d di=%d w=%d r=%.17g a=%d b=%s c=%s f=%s o=%s G=%d CO=%d TO=%d RG=%d\n", (int)ok, (int)P.UsingSTAR, (int)P.Phred_Type,
                    (int)P.Max_LowPhred_Len, (int)P.Min_Phred, (int)P.Min_MapQual, P.Concord_Dist_Pos, P.Concord_Dist_Idx, P.Min_Edge_Weight, P.DiscordantRatio, P.MaxAllowedDegree,
                    P.Input_BAM.c_str(), P.Input_Chim_BAM.c_str(), P.Input_FASTA.c_str(), P.Output_Prefix.c_str(), (int)P.Print_Graph, (int)P.Print_Components_Ordering,
                    (int)P.Print_Total_Ordering, (int)P.Print_Rearranged_Genome);
        return 0;
    }
    if (!parse_arguments(argc, argv, P, dumpdir)) return 0;  // the reference's main returns 0 either way
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
        std::cout << "[oracle +" << s << "s] " << what << std::endl;
    };
    StageSink sink{dumpdir};
    std::map<std::string, int> RefTable;
    std::vector<std::string> RefName;
    std::vector<int> RefLength;
    BuildRefName(P.Input_BAM, RefName, RefTable, RefLength);
    for (auto& kv : RefTable) std::cout << "Reference name " << kv.first << "\t-->\t" << kv.second << std::endl;

    SBamrecord_t Chimrecord;
    if (P.Input_Chim_BAM.size() != 0) BuildChimericSBamRecord(Chimrecord, P.Input_Chim_BAM, P);  // (src/main.cpp:34-36: --bwa runs without one)
    lap("chimeric records merged");
    auto dump_chimrecord = [&]() {
        if (!dumpdir.empty()) {
            std::ofstream o(dumpdir + "/chimrecord.txt");
            o << "# ReadLen=" << P.ReadLen << "\n";
            for (const ReadRec_t& r : Chimrecord) {
                o << r.Qname << '\t' << r.FirstTotalLen << '\t' << r.SecondTotalLen << '\t' << (r.FirstRead.empty() ? 0 : (int)r.FirstLowPhred) << '\t'
                  << (r.SecondMate.empty() ? 0 : (int)r.SecondLowPhred);
                for (int m = 0; m < 2; m++) {
                    o << (m ? "\tS" : "\tF");
                    for (const SingleBamRec_t& b : (m ? r.SecondMate : r.FirstRead))
                        o << ' ' << b.RefID << ',' << b.RefPos << ',' << b.ReadPos << ',' << b.MatchRef << ',' << b.MatchRead << ',' << (int)b.IsReverse;
                }
                o << '\n';
            }
        }
    };
    if (P.UsingSTAR) dump_chimrecord();
    SegmentGraph_t G(P);
    G.Construct(RefLength, Chimrecord, P.Input_BAM, dumpdir.empty() ? nullptr : &sink, &P.ReadLen);
    if (!P.UsingSTAR) dump_chimrecord();  // (--bwa: RawEdges has rebuilt the fragments from the partially aligned reads)
    lap("segment graph built");
    if (P.Print_Graph) OutputGraph(P.Output_Prefix + "_graph.txt", G);
    Orderer ord(G);
    std::vector<std::vector<int>> Components = ord.Ordering();
    lap("components ordered");
    if (P.Print_Components_Ordering) WriteComponents(P.Output_Prefix + "_component_pri.txt", Components);
    if (const char* sf = std::getenv("ORACLE_STATS_FILE")) {  // (bench.py's cpu_baseline leg: the uniqueness gate without the stage dumps)
        size_t ge20 = 0, largest = 0;
        for (const auto& comp : Components) { ge20 += comp.size() >= 20; largest = std::max(largest, comp.size()); }
        std::ofstream o(sf);
        o << "components\t" << ord.stats.components << "\nambiguous\t" << ord.stats.ambiguous << "\ntoo_large\t" << ord.stats.too_large << "\nmincut_splits\t" << ord.stats.mincut_splits
          << "\nn_components_ge20\t" << ge20 << "\nlargest\t" << largest << "\n";
        for (const std::string& ln : ord.ambiguous_notes) o << "ambiguous_problem\t" << ln << "\n";
    }
    if (!dumpdir.empty()) {
        WriteComponents(dumpdir + "/orders.txt", Components);
        { std::ofstream a(dumpdir + "/ambiguous.txt"); for (const std::string& ln : ord.ambiguous_notes) a << ln << "\n"; }
        std::ofstream o(dumpdir + "/order_stats.txt");
        o << "components\t" << ord.stats.components << "\nsolved\t" << ord.stats.solved << "\nambiguous\t" << ord.stats.ambiguous << "\ntoo_large\t" << ord.stats.too_large
          << "\nmincut_splits\t" << ord.stats.mincut_splits << "\nkept_records\t" << G.n_kept_records << "\nbreak_record\t" << G.n_break_record << "\n";
    }
    // src/main.cpp:45-48 runs the component post-merge always; its result only reaches -TO / -RG (`_sv.txt` is invariant to it, SURVEY.md
    // A.9) and MergeSingleton_Insert is quadratic (:4154-4225), so the oracle runs it only when one of the two outputs is asked for
    const std::vector<std::vector<int>> PrimaryComponents = Components;
    if (P.Print_Total_Ordering || P.Print_Rearranged_Genome) {
        std::vector<opost::NodeGeo> geo(G.vNodes.size());
        for (size_t i = 0; i < geo.size(); i++) geo[i] = opost::NodeGeo{G.vNodes[i].Chr, G.vNodes[i].Position, G.vNodes[i].Length};
        Components = opost::SortComponents(Components);
        Components = opost::MergeSingleton(geo, Components, RefLength);
        Components = opost::SortComponents(Components);
        Components = opost::MergeComponents(geo, Components);
        lap("components merged");
        if (P.Print_Total_Ordering) WriteComponents(P.Output_Prefix + "_component.txt", Components);
        if (P.Print_Rearranged_Genome) {
            std::map<std::string, int> RefTable;
            for (size_t i = 0; i < RefName.size(); i++) RefTable[RefName[i]] = (int)i;
            std::vector<std::string> RefSequence;
            if (opost::BuildRefSeq(P.Input_FASTA, RefTable, RefLength, RefSequence)) opost::OutputNewGenome(geo, Components, RefSequence, RefName, P.Output_Prefix + "_genome.fa");
            else std::cout << "FASTA file doesn't match BAM file" << std::endl;
        }
    }

    std::vector<pii> Node_NewChr(G.vNodes.size());
    for (size_t i = 0; i < Components.size(); i++)
        for (size_t j = 0; j < Components[i].size(); j++) Node_NewChr[std::abs(Components[i][j]) - 1] = pii((int)i, (int)j);

    EdgeBPMap ExactBP, ExactBP_concord_support;
    G.ExactBreakpoint(Chimrecord, ExactBP);
    G.ExactBPConcordantSupport(P.Input_BAM, Chimrecord, ExactBP, ExactBP_concord_support);
    lap("breakpoint support counted");
    if (!dumpdir.empty()) {
        std::ofstream o(dumpdir + "/breakpoints.txt");
        o << "# ind1\thead1\tind2\thead2\t[bp1,bp2:sup1,sup2]...\n";
        for (const Edge_t& e : G.vEdges) {
            o << e.Ind1 << '\t' << (int)e.Head1 << '\t' << e.Ind2 << '\t' << (int)e.Head2;
            EdgeBPMap::const_iterator b = ExactBP.find(e), s = ExactBP_concord_support.find(e);
            size_t nb = (b == ExactBP.end()) ? 0 : b->second.size();
            for (size_t k = 0; k < s->second.size(); k++) {
                if (nb) o << '\t' << b->second[k].first << ',' << b->second[k].second;
                else o << "\t-,-";
                o << ':' << s->second[k].first << ',' << s->second[k].second;
            }
            o << '\n';
        }
    }
    G.DeMultiplyDisEdges();
    WriteBEDPE(P.Output_Prefix + "_sv.txt", G, Components, Node_NewChr, RefName, ExactBP, ExactBP_concord_support, P);
    lap("Done.");
    return 0;
}
