// TEST INFRASTRUCTURE — built only where /root/reference exists (oracle/Makefile target `ref`), into oracle/_ref/.
//
// Pins the PROPOSAL SEQUENCE of the reference's check-matrix local search: optimize_H.cpp is one translation unit with its
// own main(), so it is included here whole with main renamed (-Dmain is spelled below) and `private` opened for
// PermutationsMatrix only; nothing of it is copied.  ref_opt_proposals() starts from PermutationsMatrix(Z, read_pcm(path))
// (optimize_H.cpp:27-52) and applies random_permute (optimize_H.cpp:66-75) `count` times under std::mt19937(seed)
// (optimize_H.cpp:132), either always from the start matrix (accept = 0: every proposal rejected, optimize_H.cpp:96) or chained
// (accept = 1: every proposal accepted), and reports (i, j, present, shift) of the mutated block of every proposal.
// (i, j) are not observable from outside random_permute, so a second generator replays the draw order the reference's code is
// expected to use; the shim CHECKS that expectation against the real objects: the proposal differs from its parent at most in
// block (i, j), its block (i, j) holds the reported values, and both generators are in the same state afterwards.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <utility>
#include <vector>

#define main acg_ref_optimize_h_main
#define private public
#include "optimize_H.cpp"
#undef private
#undef main

extern "C" int ref_opt_proposals(const char *h_path, int Z, unsigned seed, int count, int accept, int32_t *out /* count*4 */) {
    TMatrix H = read_pcm(h_path);
    PermutationsMatrix cur(Z, H);
    std::mt19937 rnd(seed), shadow(seed);
    for (int k = 0; k < count; k++) {
        PermutationsMatrix nxt = cur.random_permute(rnd);
        // expected draw order (optimize_H.cpp:67-73): block row, block column, [a coin only if the block is present], shift
        const int i = (int) (shadow() % (unsigned) cur._blocks.size());
        const int j = (int) (shadow() % (unsigned) cur._blocks[0].size());
        if (cur._blocks[i][j]) (void) shadow();
        (void) shadow();
        if (!(rnd == shadow)) return 100 + k;                                  // draw count / order differs from the expectation
        for (size_t a = 0; a < cur._blocks.size(); a++)
            for (size_t b = 0; b < cur._blocks[a].size(); b++)
                if (((int) a != i || (int) b != j) && (cur._blocks[a][b] != nxt._blocks[a][b] || cur._diagonals[a][b] != nxt._diagonals[a][b]))
                    return 200 + k;                                            // something else than block (i, j) changed
        out[4 * k + 0] = i;
        out[4 * k + 1] = j;
        out[4 * k + 2] = nxt._blocks[i][j] ? 1 : 0;
        out[4 * k + 3] = nxt._diagonals[i][j];
        if (accept) cur = nxt;
    }
    return 0;
}
