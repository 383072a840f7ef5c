// Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (the reference runs its own
// unsafe code under sanitizers / miri; here the library's host code - graph, SSA flattening, dense allocation, RegisterAllocator with
// spills, bytecode producer and importer, root split, term plan, links of the linked prune - compiled with g++ from the very headers
// capi.hip includes, no GPU).  Driven over every model under models/ and a few hostile inputs; exit code 0 = no report.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I fidget_amd/csrc host_frontend_san.cpp
#include <stdio.h>

#include <fstream>
#include <iostream>
#include <sstream>

#include "host_graph.hpp"
#include "host_regtape.hpp"

static int run_model(const char* path) {
    std::ifstream f(path);
    std::stringstream ss;
    ss << f.rdbuf();
    fh::Graph g;
    std::string err;
    const uint32_t root = g.parse(ss.str().c_str(), err);
    if (root == fh::NO_NODE) { fprintf(stderr, "%s: parse failed: %s\n", path, err.c_str()); return 1; }
    fh::SsaProgram prog;
    if (!fh::flatten(g, {root}, prog, err)) { fprintf(stderr, "%s: flatten: %s\n", path, err.c_str()); return 1; }
    fh::HostTape t;
    if (!fh::allocate(prog, t, err)) { fprintf(stderr, "%s: allocate: %s\n", path, err.c_str()); return 1; }
    size_t words_total = 0;
    for (uint32_t N : {255u, 24u, 12u, 8u, 3u}) {
        fh::RegTapeOut rt;
        if (!fh::reg_tape(t, N, rt, err)) { fprintf(stderr, "%s: reg_tape %u: %s\n", path, N, err.c_str()); return 1; }
        std::vector<uint32_t> words;
        uint32_t regs = 0, mem = 0;
        if (!fh::reg_tape_bytecode(rt, N, words, regs, mem)) continue;   // (the wire format has no room for that many slots)
        words_total += words.size();
        // ... and back in through the importer
        fh::SsaProgram back;
        if (!fh::from_bytecode(words.data(), words.size(), back, err)) { fprintf(stderr, "%s: from_bytecode (N = %u): %s\n", path, N, err.c_str()); return 1; }
        fh::drop_dead(back);
        fh::HostTape t2;
        if (!fh::allocate(back, t2, err)) { fprintf(stderr, "%s: allocate after import: %s\n", path, err.c_str()); return 1; }
    }
    std::vector<fh::SsaProgram> groups;
    const int gop = fh::split_root(prog, 32, 4, groups);
    for (auto& gp : groups) { fh::HostTape tg; if (!fh::allocate(gp, tg, err)) { fprintf(stderr, "%s: group allocate: %s\n", path, err.c_str()); return 1; } }
    fh::TermPlan plan;
    const bool planned = fh::plan_terms(prog, 32, 4, 16, plan);
    if (planned) for (auto& gp : plan.groups) { fh::HostTape tg; if (!fh::allocate(gp, tg, err)) { fprintf(stderr, "%s: term group allocate: %s\n", path, err.c_str()); return 1; } }
    std::vector<uint64_t> links, cops;
    const bool linked = fh::compute_links(t, links, cops);
    printf("%s: %zu ops, %u regs, %u choices, bytecode words %zu, root split %d (%zu groups), term plan %d, links %d\n", path, t.ops.size(), t.n_regs,
           t.n_choices, words_total, gop, groups.size(), (int)planned, (int)linked);
    return 0;
}

int main(int argc, char** argv) {
    int bad = 0;
    for (int i = 1; i < argc; i++) bad += run_model(argv[i]);
    // hostile inputs: the parser and the importer must refuse them, not read past them
    const char* texts[] = {"", "# nothing\n", "_0 var-x\n_1 add _0 _9\n", "_0 const\n", "_0 sqrt\n", "_0 const 1e99999\n_1 neg _0\n", "_0 var-x\n_1 bogus _0\n",
                           "_0 var-x\n_0 var-y\n_1 add _0 _0\n"};
    for (const char* tx : texts) {
        fh::Graph g;
        std::string err;
        const uint32_t root = g.parse(tx, err);
        if (root != fh::NO_NODE) {
            fh::SsaProgram p;
            if (fh::flatten(g, {root}, p, err)) { fh::HostTape t; (void)fh::allocate(p, t, err); }
        }
    }
    // bytecode: valid start marker, then ops with small opcodes and registers (so that bodies are really walked), random immediates,
    // mutations of a real tape's words; with and without the end marker
    uint32_t rng = 12345;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (int trial = 0; trial < 20000; trial++) {
        std::vector<uint32_t> w = {0xFFFFFFFFu, 0};
        const int n = rnd() % 40;
        for (int k = 0; k < n; k++) {
            const uint32_t op = rnd() % 48, r1 = rnd() % ((trial & 1) ? 8 : 256), r2 = rnd() % ((trial & 2) ? 8 : 256), r3 = rnd() % ((trial & 4) ? 8 : 256);
            w.push_back(op | r1 << 8 | r2 << 16 | r3 << 24);
            w.push_back((trial & 8) ? rnd() % 20 : rnd() * 977u);
        }
        if (trial % 3) { w.push_back(0xFFFFFFFFu); w.push_back(0xFFFFFFFFu); }
        fh::SsaProgram p;
        std::string err;
        if (fh::from_bytecode(w.data(), w.size(), p, err)) { fh::drop_dead(p); fh::HostTape t; if (fh::allocate(p, t, err)) { fh::RegTapeOut rt; (void)fh::reg_tape(t, 3 + trial % 9, rt, err); } }
    }
    printf("hostile inputs refused or handled\n");
    return bad != 0;
}
