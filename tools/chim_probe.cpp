// Host-only probe of the chimeric side of the ingest (BGZF inflate + record decode + BuildChimericSBamRecord counterpart) with a
// hash of the result: timing on the GPU box's host cores and old-against-new comparisons.  Not part of the product.
//   hipcc -O3 -std=c++17 -o build/chim_probe tools/chim_probe.cpp squid_amd/csrc/sq_bam.cpp squid_amd/csrc/sq_chimeric.cpp -lz -lpthread -ldl
//   SQUID_INGEST_TIMING=1 SQUID_CHIM_PROF=1 build/chim_probe <chimeric.bam> [threads]
#include "../squid_amd/csrc/sq_internal.h"
#include <iostream>
using namespace sq;
namespace sq { int fail(sq_ctx* c, int code, const std::string& m) { if (c) c->err = m; return code; } }
int main(int argc, char** argv) {
    sq_ctx* c = new sq_ctx();
    c->P.phred_type = 1; c->P.max_lowphred_len = 10; c->P.min_phred = 4; c->P.min_mapqual = 1; c->P.world_size = 1;
    int threads = argc > 2 ? atoi(argv[2]) : 16;
    c->pool.reset(new HostPool(threads));
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* s) { std::cerr << s << " " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl; };
    ParseOpts o{c->P.phred_type, c->P.min_phred, c->P.max_lowphred_len, true, nullptr};
    HostBatch all; all.clear(); bool got = false; std::string err;
    int rc = parse_bam_file(argv[1], o, (size_t)1 << 40, threads, err, [&](const HostBatch& hb) { all = hb; got = true; return 0; });
    lap("parse");
    sq_aln_batch b; all.view(&b, true);
    rc = build_fragments(c, &b);
    lap("fragments");
    std::cerr << rc << " frags " << c->frags.size() << " names " << c->chim_names.size() << std::endl;
    c->frags0 = c->frags;
    lap("copy");
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
    for (const Frag& f : c->frags) { for (char ch : f.name) mix((unsigned char)ch); mix(f.atot); mix(f.btot); mix(f.alow); mix(f.blow);
        for (const std::vector<Blk>* v : {&f.a, &f.b}) { mix(v->size()); for (const Blk& k : *v) { mix(k.refid); mix(k.refpos); mix(k.readpos); mix(k.matchref); mix(k.matchread); mix(k.rev); mix(k.first); } } }
    for (const std::string& n : c->chim_names) { for (char ch : n) mix((unsigned char)ch); mix(255); }
    std::cerr << "hash " << h << " read_len " << c->read_len << std::endl;
}
