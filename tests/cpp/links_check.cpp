// links_check.cpp — CPU check of chalkydri_amd/csrc/ck_links.h, the bit-level statement of the segmentation stage's
// connectivity rule that k_tile (k_ccl.hip) executes on the GPU.
//
// For random tri-state maps (and the shapes that stress the rule: single columns, widths around word boundaries, frame
// edges, checkerboards, long diagonals) it cuts every row into 32-pixel words exactly as the kernel does, takes the links
// ck_links_of_run() reports for every run, replays them through a sequential union-find over run nodes, and compares the
// resulting partition (canonical label = smallest pixel index, size per pixel) with the oracle's ora_segment().  The tile
// is the whole image here (any width, any height): what is checked is the completeness and soundness of the link rule,
// not the kernel's parallel machinery — the -m gpu parity tests do that.
//
// Build / run: tests/test_links_host.py (g++ -O2, links with oracle/libck_oracle.so).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../chalkydri_amd/csrc/ck_links.h"
#include "../../oracle/ck_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}

static uint32_t uf_find(std::vector<uint32_t> &p, uint32_t a) {
    while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; }
    return a;
}
static void uf_union(std::vector<uint32_t> &p, uint32_t a, uint32_t b) {
    a = uf_find(p, a); b = uf_find(p, b);
    if (a == b) return;
    if (a < b) p[b] = a; else p[a] = b;
}

static size_t g_pair_nodes, g_pair_links, g_runs;

// ---- the PAIR formulation (nodes = components of 2 x 32 blocks; ck_links.h, "nodes over PAIRS of rows") ---------------------
// Replays exactly what k_tile does with it: every node gets a lookup entry and a min entry, the pixels of a node find its
// lookup entry, pairs are joined by the word-level links of their facing rows (ck_links_of_word on origin-masked words) with
// the targets resolved to the upper pair's nodes, the non-origin columns are single-pixel nodes.
static int check_pairs(const std::vector<uint8_t> &t, int w, int h, const std::vector<uint32_t> &m, const std::vector<uint32_t> &lab,
                       const char *what, size_t n_runs) {
    const int nw = (w + 31) / 32, np = (h + 1) / 2;
    auto mask = [&](int y, int wd, int c) -> uint32_t {
        if (y < 0 || y >= h || wd < 0 || wd >= nw) return 0u;
        return m[((size_t)y * nw + wd) * 2 + c] & ck_origin32(wd * 32, w);
    };
    struct PW { uint32_t Mt, Mb, link, S2; };
    auto pw = [&](int p, int wd, int c) -> PW {
        PW r;
        r.Mt = mask(2 * p, wd, c); r.Mb = mask(2 * p + 1, wd, c);
        r.link = ck_pair_link32(c == 0, r.Mt, r.Mb, (mask(2 * p, wd - 1, c) >> 31) != 0, (mask(2 * p + 1, wd - 1, c) >> 31) != 0);
        r.S2 = ck_pair_starts32(r.Mt, r.Mb, r.link);
        return r;
    };
    // frame pixel of a word-local entry (row << 5 | column) of pair p, word wd
    auto pix = [&](int p, int wd, int e) -> uint32_t { return (uint32_t)((2 * p + (e >> 5)) * w + wd * 32 + (e & 31)); };
    // lookup pixel of the node of pair p, word wd, colour c that holds column x (the column must be occupied)
    auto lookup_of = [&](int p, int wd, int c, int x) -> uint32_t {
        const PW r = pw(p, wd, c);
        return pix(p, wd, ck_pair_lookup(r.Mt, ck_run_start32(r.S2, x)));
    };
    auto last_lookup = [&](int p, int wd, int c) -> uint32_t {
        const PW r = pw(p, wd, c);
        return pix(p, wd, ck_pair_lookup(r.Mt, ck_last_start32(r.S2)));
    };
    std::vector<uint32_t> parent((size_t)w * h), node_of((size_t)w * h, 0xFFFFFFFFu);
    for (size_t i = 0; i < parent.size(); i++) parent[i] = (uint32_t)i;
    size_t n_nodes = 0, n_links = 0;
    for (int p = 0; p < np; p++)
        for (int wd = 0; wd < nw; wd++)
            for (int c = 0; c < 2; c++) {
                const PW r = pw(p, wd, c);
                for (uint32_t S = r.S2; S; S &= S - 1) {
                    const int s = ck_ctz32(S);
                    const uint32_t span = ck_span32(r.S2, s);
                    const uint32_t lk = pix(p, wd, ck_pair_lookup(r.Mt, s)), mn = pix(p, wd, ck_pair_min(r.Mt, span, s));
                    if (mn > lk) { fprintf(stderr, "FAIL(pairs) %s: min entry behind the lookup entry\n", what); return 1; }
                    uf_union(parent, lk, mn);
                    n_nodes++;
                    for (uint32_t R = r.Mt & span; R; R &= R - 1) node_of[(size_t)(2 * p) * w + wd * 32 + ck_ctz32(R)] = lk;
                    for (uint32_t R = r.Mb & span; R; R &= R - 1) node_of[(size_t)(2 * p + 1) * w + wd * 32 + ck_ctz32(R)] = lk;
                }
                if ((r.link & 1u) && ((r.Mt | r.Mb) & 1u)) { uf_union(parent, lookup_of(p, wd, c, 0), last_lookup(p, wd - 1, c)); n_links++; }
                if (p == 0 || !r.Mt) continue;
                const uint32_t U = mask(2 * p - 1, wd, c);
                const bool up31 = (mask(2 * p - 1, wd - 1, c) >> 31) != 0, un0 = (mask(2 * p - 1, wd + 1, c) & 1u) != 0;
                const ck_word_links L = ck_links_of_word(c == 0, r.Mt, U, 0xFFFFFFFFu, false, up31, un0, true);
                for (uint32_t E = L.Ev; E; E &= E - 1) {
                    const int x = ck_ctz32(E);
                    uf_union(parent, lookup_of(p, wd, c, x), lookup_of(p - 1, wd, c, x)); n_links++;
                }
                for (uint32_t E = L.DL; E; E &= E - 1) {
                    const int x = ck_ctz32(E);
                    uf_union(parent, lookup_of(p, wd, c, x), x > 0 ? lookup_of(p - 1, wd, c, x - 1) : last_lookup(p - 1, wd - 1, c)); n_links++;
                }
                for (uint32_t E = L.DR; E; E &= E - 1) {
                    const int x = ck_ctz32(E);
                    uf_union(parent, lookup_of(p, wd, c, x), x < 31 ? lookup_of(p - 1, wd, c, x + 1) : lookup_of(p - 1, wd + 1, c, 0)); n_links++;
                }
            }
    // the non-origin columns: single-pixel nodes, joined by what the rule lets their origin neighbours do
    for (int y = 0; y < h; y++)
        for (int side = 0; side < 2; side++) {
            const int xn = side ? w - 1 : 0;
            if (side && w == 1) continue;
            const size_t i = (size_t)y * w + xn;
            const uint8_t v = t[i];
            if (v == 127) continue;
            node_of[i] = (uint32_t)i;
            if (w < 3) continue;
            if (xn == 0 && t[i + 1] == v) uf_union(parent, (uint32_t)i, node_of[i + 1]);                       // (1, y) joins left
            if (v == 255 && y + 1 < h) {
                const size_t j = (size_t)(y + 1) * w + (xn == 0 ? 1 : w - 2);                                   // (1, y + 1) joins up-left, (w - 2, y + 1) up-right
                if (t[j] == 255) uf_union(parent, (uint32_t)i, node_of[j]);
            }
        }
    for (size_t i = 0; i < lab.size(); i++) {
        const uint32_t mine = t[i] == 127 ? 0xFFFFFFFFu : uf_find(parent, node_of[i]);
        if (mine != lab[i]) {
            fprintf(stderr, "FAIL(pairs) %s %dx%d: pixel (%d,%d) label %u, oracle %u\n", what, w, h, (int)(i % w), (int)(i / w), mine, lab[i]);
            return 1;
        }
    }
    if (n_nodes > n_runs) { fprintf(stderr, "FAIL(pairs) %s: more nodes (%zu) than one-row runs (%zu)\n", what, n_nodes, n_runs); return 1; }
    // (the links between two pairs are one per pair of facing ONE-ROW runs: several of them can join the same two nodes — those are
    // unions that find one root twice on the GPU — but never more than the one-row formulation has)
    if (n_links > 2 * n_runs + 8) { fprintf(stderr, "FAIL(pairs) %s: %zu links for %zu one-row runs\n", what, n_links, n_runs); return 1; }
    g_pair_nodes += n_nodes; g_pair_links += n_links; g_runs += n_runs;
    return 0;
}

static int check(const std::vector<uint8_t> &t, int w, int h, const char *what) {
    const int nw = (w + 31) / 32;
    // per row, per word, per colour (0 white, 1 black)
    std::vector<uint32_t> m((size_t)h * nw * 2, 0u);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t v = t[(size_t)y * w + x];
            if (v == 255) m[((size_t)y * nw + x / 32) * 2 + 0] |= 1u << (x & 31);
            if (v == 0) m[((size_t)y * nw + x / 32) * 2 + 1] |= 1u << (x & 31);
        }
    std::vector<uint32_t> parent((size_t)w * h);
    for (size_t i = 0; i < parent.size(); i++) parent[i] = (uint32_t)i;
    size_t n_links = 0, n_runs = 0;
    for (int y = 0; y < h; y++)
        for (int wd = 0; wd < nw; wd++)
            for (int c = 0; c < 2; c++) {
                const uint32_t M = m[((size_t)y * nw + wd) * 2 + c];
                const uint32_t O = ck_origin32(wd * 32, w);
                const uint32_t U = y > 0 ? m[((size_t)(y - 1) * nw + wd) * 2 + c] : 0u;
                const uint32_t Su = ck_starts32(U, O);
                uint32_t S = ck_starts32(M, O);
                const bool mp31 = wd > 0 && (m[((size_t)y * nw + wd - 1) * 2 + c] >> 31);
                const bool up31 = y > 0 && wd > 0 && (m[((size_t)(y - 1) * nw + wd - 1) * 2 + c] >> 31);
                const bool un0 = y > 0 && wd < nw - 1 && (m[((size_t)(y - 1) * nw + wd + 1) * 2 + c] & 1u);
                const bool on0 = wd < nw - 1 && (ck_origin32((wd + 1) * 32, w) & 1u);
                while (S) {
                    const int i = ck_ctz32(S);
                    S &= S - 1;
                    n_runs++;
                    const ck_run_links L = ck_links_of_run(c == 0, M, U, O, i, mp31, up31, un0, on0);
                    const uint32_t node = (uint32_t)(y * w + wd * 32 + i);
                    // the pixels of the run hang on its node
                    for (uint32_t R = L.R; R; R &= R - 1) parent[(size_t)y * w + wd * 32 + ck_ctz32(R)] = node;
                    if (L.flags & CK_LINK_HLEFT) {
                        const uint32_t Mp = m[((size_t)y * nw + wd - 1) * 2 + c];
                        const uint32_t Sp = ck_starts32(Mp, ck_origin32((wd - 1) * 32, w));
                        uf_union(parent, node, (uint32_t)(y * w + (wd - 1) * 32 + ck_last_start32(Sp)));
                        n_links++;
                    }
                    for (uint32_t G = L.G; G; G &= G - 1) {
                        const int j = ck_ctz32(G);
                        uf_union(parent, node, (uint32_t)((y - 1) * w + wd * 32 + ck_run_start32(Su, j)));
                        n_links++;
                    }
                    if (L.flags & CK_LINK_CROSS_L) {
                        const uint32_t Up = m[((size_t)(y - 1) * nw + wd - 1) * 2 + c];
                        const uint32_t Sp = ck_starts32(Up, ck_origin32((wd - 1) * 32, w));
                        uf_union(parent, node, (uint32_t)((y - 1) * w + (wd - 1) * 32 + ck_last_start32(Sp)));
                        n_links++;
                    }
                    if (L.flags & CK_LINK_CROSS_R) {
                        uf_union(parent, node, (uint32_t)((y - 1) * w + (wd + 1) * 32));
                        n_links++;
                    }
                }
            }
    // the word-level statement of the same links (ck_links_of_word): replayed into a second forest, which must give the same partition
    std::vector<uint32_t> parent2((size_t)w * h);
    for (size_t i = 0; i < parent2.size(); i++) parent2[i] = (uint32_t)i;
    size_t n_links2 = 0;
    for (int y = 0; y < h; y++)
        for (int wd = 0; wd < nw; wd++)
            for (int c = 0; c < 2; c++) {
                const uint32_t M = m[((size_t)y * nw + wd) * 2 + c];
                if (!M) continue;
                const uint32_t O = ck_origin32(wd * 32, w);
                const uint32_t U = y > 0 ? m[((size_t)(y - 1) * nw + wd) * 2 + c] : 0u;
                const uint32_t S = ck_starts32(M, O), Su = ck_starts32(U, O);
                const uint32_t Mp = wd > 0 ? m[((size_t)y * nw + wd - 1) * 2 + c] : 0u;
                const uint32_t Up = (y > 0 && wd > 0) ? m[((size_t)(y - 1) * nw + wd - 1) * 2 + c] : 0u;
                const bool un0 = y > 0 && wd < nw - 1 && (m[((size_t)(y - 1) * nw + wd + 1) * 2 + c] & 1u);
                const bool on0 = wd < nw - 1 && (ck_origin32((wd + 1) * 32, w) & 1u);
                const ck_word_links L = ck_links_of_word(c == 0, M, U, O, (Mp >> 31) != 0, (Up >> 31) != 0, un0, on0);
                const uint32_t rowb = (uint32_t)(y * w + wd * 32), upb = (uint32_t)((y - 1) * w + wd * 32);
                for (uint32_t R = M; R; R &= R - 1) { const int x = ck_ctz32(R); parent2[rowb + x] = rowb + ck_run_start32(S, x); }
                const uint32_t Opv = wd > 0 ? ck_origin32((wd - 1) * 32, w) : 0u;
                if (L.hleft) { uf_union(parent2, rowb, rowb - 32 + ck_last_start32(ck_starts32(Mp, Opv))); n_links2++; }
                for (uint32_t E = L.Ev; E; E &= E - 1) {
                    const int x = ck_ctz32(E);
                    uf_union(parent2, rowb + ck_run_start32(S, x), upb + ck_run_start32(Su, x)); n_links2++;
                }
                for (uint32_t E = L.DL; E; E &= E - 1) {
                    const int x = ck_ctz32(E);
                    const uint32_t tgt = x > 0 ? upb + ck_run_start32(Su, x - 1) : upb - 32 + ck_last_start32(ck_starts32(Up, Opv));
                    uf_union(parent2, rowb + ck_run_start32(S, x), tgt); n_links2++;
                }
                for (uint32_t E = L.DR; E; E &= E - 1) {
                    const int x = ck_ctz32(E);
                    uf_union(parent2, rowb + ck_run_start32(S, x), upb + x + 1); n_links2++;
                }
            }
    std::vector<uint32_t> lab((size_t)w * h), sz((size_t)w * h), cnt((size_t)w * h, 0u);
    ora_segment(t.data(), w, h, lab.data(), sz.data());
    for (size_t i = 0; i < lab.size(); i++) {
        uint32_t mine = t[i] == 127 ? 0xFFFFFFFFu : uf_find(parent2, (uint32_t)i);
        if (mine != lab[i]) {
            fprintf(stderr, "FAIL(word links) %s %dx%d: pixel (%d,%d) label %u, oracle %u\n", what, w, h, (int)(i % w), (int)(i / w), mine, lab[i]);
            return 1;
        }
    }
    if (n_links2 > 2 * n_runs + 8) { fprintf(stderr, "FAIL %s: %zu word links for %zu runs\n", what, n_links2, n_runs); return 1; }
    for (size_t i = 0; i < lab.size(); i++) {
        uint32_t mine = t[i] == 127 ? 0xFFFFFFFFu : uf_find(parent, (uint32_t)i);
        if (mine != lab[i]) {
            fprintf(stderr, "FAIL %s %dx%d: pixel (%d,%d) label %u, oracle %u\n", what, w, h, (int)(i % w), (int)(i / w), mine, lab[i]);
            return 1;
        }
        if (mine != 0xFFFFFFFFu) cnt[mine]++;
    }
    for (size_t i = 0; i < lab.size(); i++)
        if (lab[i] != 0xFFFFFFFFu && cnt[lab[i]] != sz[i]) { fprintf(stderr, "FAIL %s: size\n", what); return 1; }
    // soundness of the economy: never more links than runs + upper runs touched would justify (every run at most one link
    // per distinct earlier run); a gross over-count would mean duplicated links, i.e. wasted unions on the GPU
    if (n_links > 2 * n_runs + 8) { fprintf(stderr, "FAIL %s: %zu links for %zu runs\n", what, n_links, n_runs); return 1; }
    return check_pairs(t, w, h, m, lab, what, n_runs);
}

int main(int argc, char **argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 300;
    int fails = 0, cases = 0;
    const int widths[] = {1, 2, 3, 5, 31, 32, 33, 34, 63, 64, 65, 66, 95, 96, 97, 127, 128, 129, 130, 160, 161, 255, 257, 300};
    for (int r = 0; r < rounds && !fails; r++) {
        int w = widths[rnd() % (sizeof widths / sizeof widths[0])], h = 1 + (int)(rnd() % 40);
        std::vector<uint8_t> t((size_t)w * h);
        int kind = (int)(rnd() % 8);
        uint32_t pw = rnd() % 100, pb = rnd() % (101 - pw);           // colour probabilities in per cent
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                uint8_t v;
                switch (kind) {
                case 0: v = ((x + y) & 1) ? 255 : 0; break;                           // checkerboard: every pixel its own run
                case 1: v = ((x - y) % 3 == 0) ? 255 : ((rnd() & 1) ? 0 : 127); break; // white diagonals down-right
                case 2: v = ((x + y) % 3 == 0) ? 255 : ((rnd() & 1) ? 0 : 127); break; // white diagonals down-left
                case 3: v = (rnd() & 1) ? 255 : 0; break;                             // dense binary noise
                case 4: v = (y & 1) ? 255 : 0; break;                                 // stripes
                case 5: v = (x & 1) ? 255 : 0; break;                                 // columns
                default: { uint32_t q = rnd() % 100; v = q < pw ? 255 : (q < pw + pb ? 0 : 127); } break;
                }
                t[(size_t)y * w + x] = v;
            }
        fails += check(t, w, h, "random");
        cases++;
    }
    if (fails) return 1;
    printf("OK %d maps (%zu one-row runs, %zu pair nodes, %zu pair links)\n", cases, g_runs, g_pair_nodes, g_pair_links);
    return 0;
}
