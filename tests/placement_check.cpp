// Host-only check of acg::placement_optimise (csrc/code.cpp): random gather pattern shaped like the QP-ADMM v-update
// (every item in three 32-lane sets); prints "<cycles before> <cycles after> <sets> <milliseconds>".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>

#include "../acg_alp_ldpc_amd/csrc/ldpc_internal.hpp"

namespace acg {
void set_error(const std::string &) {}  // code.cpp reports through the library's error slot (api.hip)
}

// "gather" mode: acg::placement_optimise_gather on a random (3,6)-regular structure (readers = variables, items = checks)
static int gather_mode(int n_checks) {
    const int n_vars = 2 * n_checks;
    uint64_t rng = 777;
    auto next = [&]() {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t) (rng >> 33);
    };
    std::vector<int> sockets;
    for (int c = 0; c < n_checks; c++)
        for (int j = 0; j < 6; j++) sockets.push_back(c);
    for (int i = (int) sockets.size() - 1; i > 0; i--) std::swap(sockets[i], sockets[next() % (uint32_t) (i + 1)]);
    std::vector<std::vector<int>> chk_of_var(n_vars);
    for (int v = 0; v < n_vars; v++)
        for (int k = 0; k < 3; k++) chk_of_var[v].push_back(sockets[3 * v + k]);
    std::vector<int> corder(n_checks), vorder(n_vars), clabel(n_checks), vlabel(n_vars, 3);
    std::iota(corder.begin(), corder.end(), 0);
    std::iota(vorder.begin(), vorder.end(), 0);
    for (int c = 0; c < n_checks; c++) clabel[c] = c < n_checks / 2 ? 6 : 7;  // two label classes: swaps must stay inside a class
    long before = 0, after = 0;
    acg::placement_optimise_gather(corder, clabel, vorder, vlabel, chk_of_var, 32, 32, 200, &before, &after);
    std::vector<int> seen(n_checks, 0);
    for (int s = 0; s < n_checks; s++) {
        if (corder[s] < 0 || corder[s] >= n_checks || seen[corder[s]]++) return printf("NOT A PERMUTATION\n"), 1;
        if (clabel[corder[s]] != (s < n_checks / 2 ? 6 : 7)) return printf("LABEL ORDER BROKEN\n"), 1;
    }
    std::vector<int> seenv(n_vars, 0);
    for (int s = 0; s < n_vars; s++)
        if (vorder[s] < 0 || vorder[s] >= n_vars || seenv[vorder[s]]++) return printf("NOT A PERMUTATION\n"), 1;
    printf("%ld %ld %d 0\n", before, after, 3 * ((n_vars + 31) / 32));
    return 0;
}

// "admm <matrix.txt> <L>": acg::admm_block_placement on a real code, annealed (mode 1) vs quasi-cyclic tuples (mode 2):
// prints "<qc 0/1> <Z> <tuple>  <annealed: u_reads v_reads v_writes>  <qc: u_reads v_reads v_writes>  <ideal: ...>
//         <annealed wave costs x4> <qc wave costs x4>" after checking that both placements are valid maps
static int admm_mode(const char *path, int L) {
    std::vector<uint8_t> Hd;
    int m = 0, n = 0;
    if (!acg::code_read_txt(path, Hd, m, n)) return printf("CANNOT READ\n"), 1;
    acg::Code c;
    if (!acg::code_build(c, Hd.data(), m, n)) return printf("CANNOT BUILD\n"), 1;
    acg::AdmmBlockPlacement P[2];
    for (int mode = 1; mode <= 2; mode++) {
        acg::AdmmBlockPlacement &p = P[mode - 1];
        acg::admm_block_placement(c, L, false, mode, p);
        // validity: every variable in exactly one thread slot, distinct cells, every group in a distinct slot,
        // the zero slot / zero cell unused
        std::vector<int> seen(c.admm.n_var, 0), cells(p.n_cells, 0), slots((size_t) p.n_gpass * L, 0);
        for (int v : p.var_of_slot)
            if (v >= 0 && (v >= c.admm.n_var || seen[v]++)) return printf("BAD var_of_slot\n"), 1;
        for (int v = 0; v < c.admm.n_var; v++) {
            if (!seen[v]) return printf("VARIABLE WITHOUT SLOT\n"), 1;
            if (p.cell_of_var[v] < 0 || p.cell_of_var[v] >= p.n_cells || p.cell_of_var[v] == p.zero_cell || cells[p.cell_of_var[v]]++)
                return printf("BAD cell\n"), 1;
        }
        for (int g = 0; g < c.admm.n_grp; g++)
            if (p.slot_of_grp[g] < 0 || p.slot_of_grp[g] >= p.n_gpass * L || p.slot_of_grp[g] == p.zero_gslot || slots[p.slot_of_grp[g]]++)
                return printf("BAD slot\n"), 1;
    }
    printf("%d %d %d  %ld %ld %ld  %ld %ld %ld  %ld %ld %ld  %d %d %d %d  %d %d %d %d\n", P[1].qc ? 1 : 0, P[1].Z, P[1].tuple, P[0].cyc_u_reads,
           P[0].cyc_v_reads, P[0].cyc_v_writes, P[1].cyc_u_reads, P[1].cyc_v_reads, P[1].cyc_v_writes, P[1].ideal_u_reads,
           P[1].ideal_v_reads, P[1].ideal_v_writes, P[0].wave_cost[0], P[0].wave_cost[1], P[0].wave_cost[2], P[0].wave_cost[3],
           P[1].wave_cost[0], P[1].wave_cost[1], P[1].wave_cost[2], P[1].wave_cost[3]);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 3 && std::string(argv[1]) == "admm") return admm_mode(argv[2], atoi(argv[3]));
    if (argc > 3 && std::string(argv[3]) == "gather") return gather_mode(atoi(argv[1]));
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
