// host-only check of the --bwa record loops in stretches against the loops in one go (no device needed): nodes and raw edges of
// bwa_nodes_and_edges with and without SQUID_BWA_PIECE.   usage: bwa_pieces_check <bam> <piece> [threads]
#include "../squid_amd/csrc/sq_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace sq;
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    sq_ctx c;
    sq_default_params(&c.P);
    c.P.min_mapqual = 1;
    c.pool.reset(new HostPool(argc > 3 ? std::atoi(argv[3]) : 7));
    std::vector<std::string> names;
    std::string err;
    if (read_bam_header(argv[1], names, c.ref_len, err)) { std::printf("header: %s\n", err.c_str()); return 1; }
    auto all = std::make_shared<HostBatch>();
    all->blk_off.assign(1, 0); all->name_off.assign(1, 0);
    ParseOpts o{c.P.phred_type, c.P.min_phred, c.P.max_lowphred_len, true, nullptr};
    if (parse_bam_file(argv[1], o, (size_t)1 << 21, 4, err, [&](const HostBatch& hb) { all->append(hb); return 0; })) { std::printf("parse: %s\n", err.c_str()); return 1; }
    c.bwa = all;
    std::vector<Node> nodes[2];
    std::vector<Edge> raw[2];
    std::vector<Frag> frags[2];
    for (int mode = 0; mode < 2; ++mode) {
        if (mode) setenv("SQUID_BWA_PIECE", argv[2], 1); else unsetenv("SQUID_BWA_PIECE");
        c.read_len = 0; c.nodes.clear(); c.frags.clear();
        const int rc = bwa_nodes_and_edges(&c, raw[mode]);
        if (rc) { std::printf("mode %d: rc %d %s\n", mode, rc, c.err.c_str()); return 1; }
        nodes[mode] = c.nodes; frags[mode] = c.frags;
        { std::vector<Edge> summed; reduce_edges(raw[mode], summed); raw[mode].swap(summed); }  // (BuildEdges' sort + sum + drop, :1943-1957: what the list is for)
        for (size_t t = 0; t < c.timer.names.size(); ++t) if (std::strstr(c.timer.names[t], "stretches")) { std::printf("   %s: %lld\n", c.timer.names[t], (long long)c.timer.launches[t]); }
        c.timer.clear();
        std::printf("mode %d: %zu records, %zu nodes, %zu edges after BuildEdges' reduction, %zu fragments, read_len %d\n", mode, all->size(), nodes[mode].size(), raw[mode].size(), frags[mode].size(), c.read_len);
    }
    int bad = 0;
    if (nodes[0].size() != nodes[1].size()) { std::printf("node counts differ\n"); bad = 1; }
    for (size_t i = 0; i < std::min(nodes[0].size(), nodes[1].size()) && bad < 10; ++i) {
        const Node &a = nodes[0][i], &b = nodes[1][i];
        if (a.chr != b.chr || a.pos != b.pos || a.len != b.len || a.support != b.support || a.depth != b.depth) { std::printf("node %zu: (%d %d %d s%d d%g) vs (%d %d %d s%d d%g)\n", i, a.chr, a.pos, a.len, a.support, a.depth, b.chr, b.pos, b.len, b.support, b.depth); ++bad; }
    }
    if (raw[0].size() != raw[1].size()) { std::printf("raw edge counts differ\n"); bad = 1; }
    for (size_t i = 0; i < std::min(raw[0].size(), raw[1].size()) && bad < 10; ++i)
        if (!edge_key_eq(raw[0][i], raw[1][i]) || raw[0][i].w != raw[1][i].w) { std::printf("edge %zu differs\n", i); ++bad; }
    if (frags[0].size() != frags[1].size()) { std::printf("fragment counts differ\n"); bad = 1; }
    for (size_t i = 0; i < std::min(frags[0].size(), frags[1].size()) && bad < 10; ++i) if (frags[0][i].name != frags[1][i].name) { std::printf("fragment %zu: %s vs %s\n", i, frags[0][i].name.c_str(), frags[1][i].name.c_str()); ++bad; }
    std::printf(bad ? "DIFFERENT\n" : "same\n");
    return bad ? 1 : 0;
}
