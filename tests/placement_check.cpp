// Host-only check of acg::placement_optimise (csrc/code.cpp): random gather pattern shaped like the QP-ADMM v-update
// (every item in three 32-lane sets); prints "<cycles before> <cycles after> <sets> <milliseconds>".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "../acg_alp_ldpc_amd/csrc/ldpc_internal.hpp"

namespace acg {
void set_error(const std::string &) {}  // code.cpp reports through the library's error slot (api.hip)
}

int main(int argc, char **argv) {
    const int n_items = argc > 1 ? atoi(argv[1]) : 544, rounds = argc > 2 ? atoi(argv[2]) : 2000;
    std::vector<acg::PlacementSet> sets;
    uint64_t rng = 12345;
    auto next = [&]() {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t) (rng >> 33);
    };
    for (int rep = 0; rep < 3; rep++) {  // three random partitions of the items into sets of 32
        std::vector<int> perm(n_items);
        std::iota(perm.begin(), perm.end(), 0);
        for (int i = n_items - 1; i > 0; i--) std::swap(perm[i], perm[next() % (uint32_t) (i + 1)]);
        for (int base = 0; base < n_items; base += 32) {
            acg::PlacementSet ps;
            ps.modulus = 32;
            for (int i = base; i < std::min(base + 32, n_items); i++) ps.items.push_back(perm[i]);
            sets.push_back(ps);
        }
    }
    std::vector<int> pos(n_items);
    std::iota(pos.begin(), pos.end(), 0);
    std::vector<long> mx;
    std::vector<int> copy = pos;
    acg::placement_optimise(copy, n_items, sets, 0, &mx);
    long before = std::accumulate(mx.begin(), mx.end(), 0l);
    const auto t0 = std::chrono::steady_clock::now();
    acg::placement_optimise(pos, n_items, sets, rounds, &mx);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    long after = std::accumulate(mx.begin(), mx.end(), 0l);
    std::vector<int> seen(n_items, 0);  // still a permutation?
    for (int p : pos) {
        if (p < 0 || p >= n_items || seen[p]++) {
            printf("NOT A PERMUTATION\n");
            return 1;
        }
    }
    printf("%ld %ld %zu %.1f\n", before, after, sets.size(), ms);
    return 0;
}
