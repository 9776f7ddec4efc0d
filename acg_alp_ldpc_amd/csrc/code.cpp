// Host-side code analysis: parity-matrix text format, Tanner-graph CSR, the LDS message layout
// used by the fused BP kernels, the QP-ADMM constraint groups, and the GF(2) helpers the
// Monte-Carlo callers need.  Plain C++ (no HIP).  Runs once per H — the reference redoes the
// equivalent work for every frame (bp.h:136-153, qp_admm.h:13-102).
#include "ldpc_internal.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

namespace acg {

// ---------------------------------------------------------------- text format
// Same observable behaviour as read_pcm (utils/parse_data.h:6-25): whitespace-separated row
// tokens; inside a token every ',' closes a cell whose value is decided by the last non-','
// character before it ('1' -> 1, anything else -> 0); a missing trailing ',' is implied.
bool code_read_txt(const char *path, std::vector<uint8_t> &H, int &m, int &n) {
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        set_error(std::string("cannot open ") + path);
        return false;
    }
    std::vector<char> buf;
    {
        char tmp[1 << 16];
        size_t got;
        while ((got = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    }
    std::fclose(f);
    H.clear();
    m = 0;
    n = -1;
    bool cell = false;  // carries across rows exactly like the reference's `bool t`
    size_t i = 0, N = buf.size();
    auto is_ws = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    while (i < N) {
        while (i < N && is_ws(buf[i])) i++;
        if (i >= N) break;
        int cols = 0;
        char last = 0;
        while (i < N && !is_ws(buf[i])) {
            last = buf[i];
            if (last == ',') {
                H.push_back(cell ? 1 : 0);
                cols++;
            } else {
                cell = (last == '1');
            }
            i++;
        }
        if (last != ',') {
            H.push_back(cell ? 1 : 0);
            cols++;
        }
        if (n < 0) n = cols;
        else if (cols != n) {
            set_error(std::string("ragged parity matrix in ") + path);
            return false;
        }
        m++;
    }
    if (m == 0 || n <= 0) {
        set_error(std::string("empty parity matrix in ") + path);
        return false;
    }
    return true;
}

// save_matrix (utils/parse_data.h:44-54): "0,1,...,1\n" per row, no trailing comma
bool code_write_txt(const Code &c, const char *path) {
    FILE *f = std::fopen(path, "wb");
    if (!f) {
        set_error(std::string("cannot open for writing ") + path);
        return false;
    }
    std::string line;
    for (int i = 0; i < c.m; i++) {
        line.clear();
        for (int j = 0; j < c.n; j++) {
            line.push_back(c.H[(size_t) i * c.n + j] ? '1' : '0');
            if (j != c.n - 1) line.push_back(',');
        }
        line.push_back('\n');
        std::fwrite(line.data(), 1, line.size(), f);
    }
    std::fclose(f);
    return true;
}

// ---------------------------------------------------------------- graph
bool code_build(Code &c, const uint8_t *H, int m, int n) {
    if (m <= 0 || n <= 0) {
        set_error("parity matrix must have m > 0 and n > 0");
        return false;
    }
    c.m = m;
    c.n = n;
    c.H.assign((size_t) m * n, 0);
    for (size_t i = 0; i < (size_t) m * n; i++) c.H[i] = H[i] ? 1 : 0;
    c.row_ptr.assign(m + 1, 0);
    c.col_ptr.assign(n + 1, 0);
    c.edge_var.clear();
    for (int i = 0; i < m; i++) {
        c.row_ptr[i] = (int) c.edge_var.size();
        for (int j = 0; j < n; j++)
            if (c.H[(size_t) i * n + j]) {
                c.edge_var.push_back(j);
                c.col_ptr[j + 1]++;
            }
    }
    c.E = (int) c.edge_var.size();
    c.row_ptr[m] = c.E;
    for (int j = 0; j < n; j++) c.col_ptr[j + 1] += c.col_ptr[j];
    c.col_edge.assign(c.E, 0);
    std::vector<int> fill(n, 0);
    for (int e = 0; e < c.E; e++) {
        int v = c.edge_var[e];
        c.col_edge[c.col_ptr[v] + fill[v]++] = e;
    }
    c.max_cdeg = c.max_vdeg = 0;
    for (int i = 0; i < m; i++) c.max_cdeg = std::max(c.max_cdeg, c.row_ptr[i + 1] - c.row_ptr[i]);
    for (int j = 0; j < n; j++) c.max_vdeg = std::max(c.max_vdeg, c.col_ptr[j + 1] - c.col_ptr[j]);
    admm_layout_build(c);
    return true;
}

bool code_is_codeword(const Code &c, const uint8_t *bits) {
    for (int i = 0; i < c.m; i++) {
        int s = 0;
        for (int e = c.row_ptr[i]; e < c.row_ptr[i + 1]; e++) s ^= (bits[c.edge_var[e]] & 1);
        if (s) return false;
    }
    return true;
}

// GetOrtogonal (utils/codeword.h:97-128) on 64-bit packed rows: for row i the pivot is its first
// non-zero column; the row is XORed into every other row holding that column; non-pivot column j
// yields generator row e_j + sum_i H'[i][j] e_pos(i).
bool code_generator(const Code &c, uint8_t *G) {
    const int m = c.m, n = c.n, W = (n + 63) / 64;
    std::vector<uint64_t> R((size_t) m * W, 0);
    for (int i = 0; i < m; i++)
        for (int e = c.row_ptr[i]; e < c.row_ptr[i + 1]; e++) {
            int j = c.edge_var[e];
            R[(size_t) i * W + (j >> 6)] |= 1ull << (j & 63);
        }
    std::vector<int> pos(m, -1);
    std::vector<uint8_t> is_main(n, 0);
    for (int i = 0; i < m; i++) {
        const uint64_t *ri = &R[(size_t) i * W];
        for (int w = 0; w < W; w++)
            if (ri[w]) {
                pos[i] = w * 64 + __builtin_ctzll(ri[w]);
                break;
            }
        if (pos[i] < 0) return false;
        const int pw = pos[i] >> 6;
        const uint64_t pb = 1ull << (pos[i] & 63);
        for (int k = 0; k < m; k++)
            if (k != i && (R[(size_t) k * W + pw] & pb)) {
                uint64_t *rk = &R[(size_t) k * W];
                for (int w = 0; w < W; w++) rk[w] ^= ri[w];
            }
        is_main[pos[i]] = 1;
    }
    std::memset(G, 0, (size_t) (n - m) * n);
    int idx = 0;
    for (int j = 0; j < n; j++)
        if (!is_main[j]) {
            G[(size_t) idx * n + j] = 1;
            for (int i = 0; i < m; i++)
                if (R[(size_t) i * W + (j >> 6)] >> (j & 63) & 1) G[(size_t) idx * n + pos[i]] = 1;
            idx++;
        }
    return true;
}

// ---------------------------------------------------------------- BP LDS layout
bool bp_layout_build(const Code &c, int L, BpLayout &o) {
    o = BpLayout();
    o.L = L;
    o.max_cdeg = c.max_cdeg;
    o.max_vdeg = c.max_vdeg;
    const int m = c.m, n = c.n;
    std::vector<int> corder(m), vorder(n);
    std::iota(corder.begin(), corder.end(), 0);
    std::iota(vorder.begin(), vorder.end(), 0);
    auto cdeg = [&](int i) { return c.row_ptr[i + 1] - c.row_ptr[i]; };
    auto vdeg = [&](int j) { return c.col_ptr[j + 1] - c.col_ptr[j]; };
    std::stable_sort(corder.begin(), corder.end(), [&](int a, int b) { return cdeg(a) > cdeg(b); });
    std::stable_sort(vorder.begin(), vorder.end(), [&](int a, int b) { return vdeg(a) > vdeg(b); });
    // The word of edge (check, j) lives at pass offset + j*L + lane of the check, i.e. in LDS bank (lane mod 32) when L is
    // a multiple of 32; the variable sweep gathers and scatters those words 32 lanes at a time.  Checks and variables
    // of equal degree are interchangeable in the order above, so for the workgroup-per-frame kernels (where the LDS
    // array is the busiest unit and more than half of its cycles were bank conflicts) they are placed to spread every
    // such access over the banks (placement_optimise_gather).  Message values do not depend on the placement.
    if (L >= 256 && c.E > 0 && getenv("ACG_BP_NO_PLACEMENT") == nullptr) {
        std::vector<int> clabel(m), vlabel(n);
        for (int i = 0; i < m; i++) clabel[i] = cdeg(i);
        for (int j = 0; j < n; j++) vlabel[j] = vdeg(j);
        std::vector<std::vector<int>> chk_of_var(n);
        for (int j = 0; j < n; j++)
            for (int k = 0; k < vdeg(j); k++) {
                const int e = c.col_edge[c.col_ptr[j] + k];  // edge ids are check-major: find the check by its row range
                chk_of_var[j].push_back((int) (std::upper_bound(c.row_ptr.begin(), c.row_ptr.end(), e) - c.row_ptr.begin()) - 1);
            }
        long before = 0, after = 0;
        placement_optimise_gather(corder, clabel, vorder, vlabel, chk_of_var, 32, 32, c.E > 20000 ? 150 : 600, &before, &after);
        if (getenv("ACG_BP_PLACEMENT_DEBUG"))
            fprintf(stderr, "[acg_ldpc] BP placement L=%d: modelled LDS cycles of the variable-sweep gathers %ld -> %ld\n", L, before, after);
    }

    o.n_cpass = (m + L - 1) / L;
    o.n_vpass = (n + L - 1) / L;
    o.c_chk.assign((size_t) o.n_cpass * L, -1);
    o.v_var.assign((size_t) o.n_vpass * L, -1);
    for (int s = 0; s < m; s++) o.c_chk[s] = corder[s];
    for (int s = 0; s < n; s++) o.v_var[s] = vorder[s];
    o.c_cnt_ge.assign(c.max_cdeg + 2, 0);
    o.v_cnt_ge.assign(c.max_vdeg + 2, 0);
    for (int i = 0; i < m; i++)
        for (int d = 0; d <= cdeg(i); d++) o.c_cnt_ge[d]++;
    for (int j = 0; j < n; j++)
        for (int d = 0; d <= vdeg(j); d++) o.v_cnt_ge[d]++;

    // A layout
    std::vector<int> edge_pos(c.E, -1);
    int off = 0;
    o.c_maxdeg.resize(o.n_cpass);
    o.c_off.resize(o.n_cpass);
    for (int p = 0; p < o.n_cpass; p++) {
        int md = cdeg(corder[(size_t) p * L]);  // sorted descending: first slot of the pass is the max
        o.c_maxdeg[p] = md;
        o.c_off[p] = off;
        for (int l = 0; l < L; l++) {
            int s = p * L + l;
            if (s >= m) break;
            int chk = corder[s];
            for (int j = 0; j < cdeg(chk); j++) edge_pos[c.row_ptr[chk] + j] = off + j * L + l;
        }
        off += md * L;
    }
    o.zero_pos = off;
    o.a_words = (off + 1 + 3) & ~3;
    if (o.a_words > 65535) {
        set_error("code too large for the fused LDS kernels (message words per frame exceed 65535)");
        return false;
    }
    // variable-side index table
    o.v_maxdeg.resize(o.n_vpass);
    o.v_idx_off.resize(o.n_vpass);
    int ioff = 0;
    for (int p = 0; p < o.n_vpass; p++) {
        int md = vdeg(vorder[(size_t) p * L]);
        o.v_maxdeg[p] = md;
        o.v_idx_off[p] = ioff;
        ioff += md * L;
    }
    o.v_apos_len = ioff;
    o.v_apos.assign((size_t) std::max(ioff, 1), (uint16_t) o.zero_pos);
    for (int p = 0; p < o.n_vpass; p++)
        for (int l = 0; l < L; l++) {
            int s = p * L + l;
            if (s >= n) break;
            int v = vorder[s];
            for (int k = 0; k < vdeg(v); k++)
                o.v_apos[(size_t) o.v_idx_off[p] + (size_t) k * L + l] = (uint16_t) edge_pos[c.col_edge[c.col_ptr[v] + k]];
        }
    return true;
}

// ---------------------------------------------------------------- QP-ADMM groups
// ConstructADMMProblem (qp_admm.h:13-102): a check of degree d >= 3 becomes the chain
// (x1,x2,a1),(a1,x3,a2),...,(a_{d-3},x_{d-1},x_d) with auxiliaries numbered from n upward in row
// order (:84-91); each triple contributes rows (+,-,-)<=0, (-,+,-)<=0, (-,-,+)<=0, (+,+,+)<=2.
void admm_layout_build(Code &c) {
    AdmmLayout &a = c.admm;
    a = AdmmLayout();
    a.n = c.n;
    int pos = c.n;
    auto add_group = [&](int type, int v0, int v1, int v2) {
        a.grp_type.push_back((uint8_t) type);
        a.grp_var.push_back(v0);
        a.grp_var.push_back(v1);
        a.grp_var.push_back(v2);
        a.n_con += (type == 3) ? 4 : type;
        a.nnz += (type == 3) ? 12 : (type == 2 ? 4 : 1);
    };
    for (int i = 0; i < c.m; i++) {
        const int *idx = &c.edge_var[c.row_ptr[i]];
        int d = c.row_ptr[i + 1] - c.row_ptr[i];
        if (d == 0) continue;
        if (d == 1) {
            add_group(1, idx[0], -1, -1);
            continue;
        }
        if (d == 2) {
            add_group(2, idx[0], idx[1], -1);
            continue;
        }
        int last = idx[0];
        for (int j = 1; j < d - 2; j++) {
            int aux = pos++;
            add_group(3, last, idx[j], aux);
            last = aux;
        }
        add_group(3, last, idx[d - 2], idx[d - 1]);
    }
    a.n_var = pos;
    a.n_grp = (int) a.grp_type.size();
    a.e.assign(a.n_var, 0.0);
    a.var_ptr.assign(a.n_var + 1, 0);
    for (int g = 0; g < a.n_grp; g++) {
        int t = a.grp_type[g];
        for (int w = 0; w < t; w++) {
            int v = a.grp_var[(size_t) g * 3 + w];
            a.var_ptr[v + 1]++;
            a.e[v] += (t == 3) ? 4.0 : (t == 2 ? 2.0 : 1.0);
        }
    }
    for (int v = 0; v < a.n_var; v++) a.var_ptr[v + 1] += a.var_ptr[v];
    a.var_grp.assign(a.var_ptr[a.n_var], 0);
    std::vector<int> fill(a.n_var, 0);
    for (int g = 0; g < a.n_grp; g++) {
        int t = a.grp_type[g];
        for (int w = 0; w < t; w++) {
            int v = a.grp_var[(size_t) g * 3 + w];
            a.var_grp[a.var_ptr[v] + fill[v]++] = g * 4 + w;
        }
    }
    a.e_min = 1e300;
    a.e_max = -1e300;
    for (int v = 0; v < a.n_var; v++) {
        a.e_min = std::min(a.e_min, a.e[v]);
        a.e_max = std::max(a.e_max, a.e[v]);
    }
    if (a.n_var == 0) a.e_min = a.e_max = 0;
}

long placement_optimise(std::vector<int> &pos_of_item, const int n_pos, const std::vector<PlacementSet> &sets,
                               const int rounds, std::vector<long> *per_set_max) {
    const int n_items = (int) pos_of_item.size();
    std::vector<int> item_at(n_pos, -1);
    for (int i = 0; i < n_items; i++) item_at[pos_of_item[i]] = i;
    std::vector<std::vector<int>> sets_of(n_items);
    std::vector<std::vector<int>> cnt(sets.size());
    long total = 0;
    for (size_t si = 0; si < sets.size(); si++) {
        cnt[si].assign(sets[si].modulus, 0);
        for (int it : sets[si].items) {
            sets_of[it].push_back((int) si);
            const int c = pos_of_item[it] % sets[si].modulus;
            total += 2 * cnt[si][c] + 1;
            cnt[si][c]++;
        }
    }
    auto move = [&](int it, int newpos) {  // updates counts and `total`
        for (int si : sets_of[it]) {
            const int m = sets[si].modulus, co = pos_of_item[it] % m, cn = newpos % m;
            if (co == cn) continue;
            total -= 2 * cnt[si][co] - 1;
            cnt[si][co]--;
            total += 2 * cnt[si][cn] + 1;
            cnt[si][cn]++;
        }
        pos_of_item[it] = newpos;
    };
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        return rng;
    };
    // simulated annealing, temperature 1.5 -> 0.05 (in units of the objective), deterministic
    const long moves = (long) rounds * n_items;
    if (n_items > 1 && n_pos > 1)
        for (long r = 0; r < moves; r++) {
            const double T = 1.5 * std::pow(0.05 / 1.5, (double) r / (double) moves);
            const int a = (int) (next() % (uint64_t) n_items);
            const int q = (int) (next() % (uint64_t) n_pos), pa = pos_of_item[a];
            if (q == pa) continue;
            const int b = item_at[q];
            const long before = total;
            move(a, q);
            if (b >= 0) move(b, pa);
            const long delta = total - before;
            if (delta > 0 && (double) (next() >> 11) * (1.0 / 9007199254740992.0) >= std::exp(-(double) delta / T)) {  // revert
                if (b >= 0) move(b, q);
                move(a, pa);
            } else {
                item_at[q] = a;
                item_at[pa] = b;
            }
        }
    if (per_set_max) {
        per_set_max->clear();
        for (size_t si = 0; si < sets.size(); si++) per_set_max->push_back(*std::max_element(cnt[si].begin(), cnt[si].end()));
    }
    return total;
}


// Joint placement for a gather (the variable sweep of the fused BP kernels): reader r at position s (lane s % lanes of
// lane group s / lanes) touches, in step k, the item items_of_reader[r][k]; the bank class of an item is its slot mod
// `modulus`.  Two kinds of move, both keeping the degree-sorted order the kernels rely on: two items with the same label
// swap slots, or two readers with the same label swap positions.  Same objective (sum of squared bank multiplicities)
// and annealing schedule as placement_optimise; deterministic.
long placement_optimise_gather(std::vector<int> &item_at_slot, const std::vector<int> &item_label, std::vector<int> &reader_at_pos,
                               const std::vector<int> &reader_label, const std::vector<std::vector<int>> &items_of_reader,
                               const int lanes, const int modulus, const int rounds, long *cycles_before, long *cycles_after) {
    const int n_items = (int) item_at_slot.size(), n_readers = (int) reader_at_pos.size();
    size_t kmax = 0;
    for (const auto &v : items_of_reader) kmax = std::max(kmax, v.size());
    if (cycles_before) *cycles_before = 0;
    if (cycles_after) *cycles_after = 0;
    if (n_readers == 0 || n_items == 0 || kmax == 0) return 0;
    const int n_lane_groups = (n_readers + lanes - 1) / lanes;
    std::vector<int> cnt((size_t) n_lane_groups * kmax * modulus, 0);
    auto cell = [&](int pos, int k, int slot) -> int & { return cnt[((size_t) (pos / lanes) * kmax + k) * modulus + slot % modulus]; };
    std::vector<int> pos_of_reader(n_readers, -1), slot_of_item(n_items, -1);
    for (int s = 0; s < n_readers; s++) pos_of_reader[reader_at_pos[s]] = s;
    for (int s = 0; s < n_items; s++) slot_of_item[item_at_slot[s]] = s;
    std::vector<std::vector<std::pair<int, int>>> uses(n_items);  // item -> (reader, k)
    long total = 0;
    for (int s = 0; s < n_readers; s++) {
        const int r = reader_at_pos[s];
        for (size_t k = 0; k < items_of_reader[r].size(); k++) {
            const int it = items_of_reader[r][k];
            uses[it].push_back({r, (int) k});
            int &cc = cell(s, (int) k, slot_of_item[it]);
            total += 2 * cc + 1;
            cc++;
        }
    }
    auto add = [&](int pos, int k, int slot, int sign) {
        int &cc = cell(pos, k, slot);
        if (sign > 0) {
            total += 2 * cc + 1;
            cc++;
        } else {
            total -= 2 * cc - 1;
            cc--;
        }
    };
    auto cycles = [&]() {
        long cyc = 0;
        for (size_t base = 0; base < cnt.size(); base += modulus) cyc += *std::max_element(cnt.begin() + base, cnt.begin() + base + modulus);
        return cyc;
    };
    if (cycles_before) *cycles_before = cycles();
    auto move_item = [&](int it, int newslot) {
        if (slot_of_item[it] % modulus != newslot % modulus)
            for (const auto &u : uses[it]) {
                add(pos_of_reader[u.first], u.second, slot_of_item[it], -1);
                add(pos_of_reader[u.first], u.second, newslot, +1);
            }
        slot_of_item[it] = newslot;
    };
    auto move_reader = [&](int r, int newpos) {
        if (pos_of_reader[r] / lanes != newpos / lanes)
            for (size_t k = 0; k < items_of_reader[r].size(); k++) {
                add(pos_of_reader[r], (int) k, slot_of_item[items_of_reader[r][k]], -1);
                add(newpos, (int) k, slot_of_item[items_of_reader[r][k]], +1);
            }
        pos_of_reader[r] = newpos;
    };
    uint64_t rng = 0xD1B54A32D192ED03ull;
    auto next = [&]() {
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        return rng;
    };
    const long moves = (long) rounds * (n_items + n_readers);
    for (long mv = 0; mv < moves; mv++) {
        const double T = 1.5 * std::pow(0.05 / 1.5, (double) mv / (double) moves);
        const long before = total;
        const bool item_move = (next() & 1) != 0;
        int a, b, sa, sb;
        if (item_move) {
            sa = (int) (next() % (uint64_t) n_items);
            sb = (int) (next() % (uint64_t) n_items);
            a = item_at_slot[sa];
            b = item_at_slot[sb];
            if (sa == sb || item_label[a] != item_label[b] || sa % modulus == sb % modulus) continue;
            move_item(a, sb);
            move_item(b, sa);
        } else {
            sa = (int) (next() % (uint64_t) n_readers);
            sb = (int) (next() % (uint64_t) n_readers);
            a = reader_at_pos[sa];
            b = reader_at_pos[sb];
            if (sa == sb || reader_label[a] != reader_label[b] || sa / lanes == sb / lanes) continue;
            move_reader(a, sb);
            move_reader(b, sa);
        }
        const long delta = total - before;
        const bool reject = delta > 0 && (double) (next() >> 11) * (1.0 / 9007199254740992.0) >= std::exp(-(double) delta / T);
        if (item_move) {
            if (reject) {
                move_item(b, sb);
                move_item(a, sa);
            } else {
                item_at_slot[sb] = a;
                item_at_slot[sa] = b;
            }
        } else {
            if (reject) {
                move_reader(b, sb);
                move_reader(a, sa);
            } else {
                reader_at_pos[sb] = a;
                reader_at_pos[sa] = b;
            }
        }
    }
    if (cycles_after) *cycles_after = cycles();
    return total;
}


// ---------------------------------------------------------------- quasi-cyclic structure
bool code_detect_qc(const Code &c, QcInfo &q) {
    const int g0 = std::gcd(c.m, c.n);
    for (int Z = g0; Z >= 2; --Z) {
        if (g0 % Z) continue;
        const int mb = c.m / Z, nb = c.n / Z;
        std::vector<int> shift((size_t) mb * nb, -1), ones((size_t) mb * nb, 0);
        bool ok = true;
        for (int r = 0; r < c.m && ok; r++) {
            const int R = r / Z, k = r % Z;
            int lastC = -1;
            for (int e = c.row_ptr[r]; e < c.row_ptr[r + 1] && ok; e++) {
                const int v = c.edge_var[e], C = v / Z, l = v % Z;
                const int sft = (l - k + Z) % Z;  // optimize_H.cpp:41
                int &cur = shift[(size_t) R * nb + C];
                if (C == lastC || (cur >= 0 && cur != sft)) ok = false;  // two ones of a row in one block / not a shift
                cur = sft;
                ones[(size_t) R * nb + C]++;
                lastC = C;
            }
        }
        for (size_t b = 0; b < ones.size() && ok; b++) ok = (ones[b] == 0 || ones[b] == Z);
        if (ok) {
            q.Z = Z;
            q.mb = mb;
            q.nb = nb;
            q.shift = shift;
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------- QP-ADMM block-kernel placement
// Task tables of the LDS-DMA ring engine (see RingTasks; consumed by bp_streamed_ring_kernel).
void ring_tasks_build(const Code &c, RingTasks &o) {
    struct Task { int first, cnt, base, lines; };
    std::vector<Task> ct, vt;
    for (int i = 0; i < c.m;) {
        Task k{i, 0, c.row_ptr[i], 0};
        while (i < c.m && k.cnt < 16 && k.lines + (c.row_ptr[i + 1] - c.row_ptr[i]) <= RING_SLOT_LINES) {
            k.lines += c.row_ptr[i + 1] - c.row_ptr[i];
            k.cnt++;
            i++;
        }
        ct.push_back(k);
    }
    for (int v = 0; v < c.n;) {
        Task k{v, 0, c.col_ptr[v], 0};
        while (v < c.n && k.cnt < 4 && k.lines + (c.col_ptr[v + 1] - c.col_ptr[v]) <= RING_VAR_EDGE_LINES) {
            k.lines += c.col_ptr[v + 1] - c.col_ptr[v];
            k.cnt++;
            v++;
        }
        vt.push_back(k);
    }
    // vector-memory operations a task certainly issues: loads (one LDS-DMA instruction per four lines, + the LLR lines of
    // a variable task) and message stores (one per line); the hard-decision byte of a variable task is predicated and
    // therefore not counted.  wait(i) for the i-th task of a wavefront: see bp_streamed_ring_kernel.
    auto pack = [&](const std::vector<Task> &tk, bool var) {
        std::vector<int32_t> out(4 * std::max<size_t>(tk.size(), 1), 0);
        for (int w = 0; w < RING_WAVES; w++) {
            std::vector<int> seq;
            for (int i = w; i < (int) tk.size(); i += RING_WAVES) seq.push_back(i);
            auto opsL = [&](int i) { return (tk[seq[i]].lines + 3) / 4 + (var ? 1 : 0); };
            auto opsS = [&](int i) { return tk[seq[i]].lines; };
            for (int i = 0; i < (int) seq.size(); i++) {
                int wl = 0, wsn = 0;
                for (int k = i + 1; k <= std::min<int>(i + RING_SLOTS - 1, (int) seq.size() - 1); k++) wl += opsL(k);
                for (int j = std::max(0, i - RING_SLOTS + 1); j <= i - 1; j++) wsn += opsS(j);
                const Task &k = tk[seq[i]];
                out[4 * seq[i] + 0] = k.first;
                out[4 * seq[i] + 1] = k.cnt;
                out[4 * seq[i] + 2] = k.base;
                out[4 * seq[i] + 3] = k.lines | (std::min(wl + wsn, 63) << 8) | (std::min(wl, 63) << 16);
            }
        }
        return out;
    };
    const int nwords = (c.n + 31) / 32;
    o.ctask = pack(ct, false);
    o.vtask = pack(vt, true);
    o.vtask_of_word.assign((size_t) std::max(nwords, 1), 0);
    {
        size_t ti = 0;
        for (int k = 0; k < nwords; k++) {
            while (ti + 1 < vt.size() && vt[ti].first + vt[ti].cnt <= 32 * k) ti++;
            o.vtask_of_word[(size_t) k] = (int32_t) ti;
        }
    }
    o.n_ctask = (int) ct.size();
    o.n_vtask = (int) vt.size();
}

// Layers of the layered min-sum schedule (see LayeredLayout).
bool bp_layered_build(const Code &c, LayeredLayout &o) {
    o = LayeredLayout();
    if (c.m <= 0 || c.n <= 0 || c.n >= 16000) {
        set_error("layered schedule: n must be below 16000 (16-bit byte offsets of the posterior cells)");
        return false;
    }
    // conflict-free same-degree check sets, in processing order
    struct Set { int deg; std::vector<int> chk; int R; int row0; };
    std::vector<Set> sets;
    QcInfo q;
    const bool qc = code_detect_qc(c, q) && q.Z >= 2;
    if (qc) {
        for (int R = 0; R < q.mb; R++) {
            Set s{c.row_ptr[R * q.Z + 1] - c.row_ptr[R * q.Z], {}, R, 0};
            for (int k = 0; k < q.Z; k++) s.chk.push_back(R * q.Z + k);
            if (s.deg > 0) sets.push_back(s);
        }
    } else {
        // greedy colouring: a check joins the first set of its degree in which none of its variables occurs yet
        std::vector<std::vector<uint8_t>> used;  // per set: variable occupied
        for (int r = 0; r < c.m; r++) {
            const int deg = c.row_ptr[r + 1] - c.row_ptr[r];
            if (deg == 0) continue;
            size_t k = 0;
            for (; k < sets.size(); k++) {
                if (sets[k].deg != deg || (int) sets[k].chk.size() >= 64) continue;
                bool free_ = true;
                for (int e = c.row_ptr[r]; e < c.row_ptr[r + 1] && free_; e++) free_ = !used[k][c.edge_var[e]];
                if (free_) break;
            }
            if (k == sets.size()) {
                sets.push_back(Set{deg, {}, -1, 0});
                used.emplace_back((size_t) c.n, 0);
            }
            sets[k].chk.push_back(r);
            for (int e = c.row_ptr[r]; e < c.row_ptr[r + 1]; e++) used[k][c.edge_var[e]] = 1;
        }
    }
    int maxdeg = 0;
    for (auto &s : sets) maxdeg = std::max(maxdeg, s.deg);
    if (maxdeg > 8) {
        set_error("layered schedule: check degree above 8 is not supported");
        return false;
    }
    // group width: the candidate that needs the fewest wavefront-steps per frame (layers / frames per wavefront)
    int bestG = 64;
    double best = 1e30;
    for (int G : {16, 20, 32, 64}) {
        long layers = 0;
        for (auto &s : sets) layers += ((long) s.chk.size() + G - 1) / G;
        const double cost = (double) layers / (64 / G);
        if (cost < best - 1e-9) {
            best = cost;
            bestG = G;
        }
    }
    const int G = bestG;
    o.G = G;
    o.qc = qc;
    o.Z = qc ? q.Z : 0;
    int off = 0;
    for (auto &s : sets) {
        int proto0 = 0;
        if (qc) {
            proto0 = (int) o.proto.size() / 2;
            for (int C = 0; C < q.nb; C++)
                if (q.shift[(size_t) s.R * q.nb + C] >= 0) {
                    o.proto.push_back(C);
                    o.proto.push_back(q.shift[(size_t) s.R * q.nb + C]);
                }
        }
        for (size_t i0 = 0; i0 < s.chk.size(); i0 += (size_t) G) {
            const int cnt = (int) std::min<size_t>((size_t) G, s.chk.size() - i0);
            o.layer.push_back(s.deg);
            o.layer.push_back(off);
            o.layer.push_back(cnt);
            o.layer.push_back(proto0 | ((int) i0 << 16));  // QC: first proto entry | first row of the chunk inside its block row
            for (int l = 0; l < G; l++) o.chk.push_back(l < cnt ? s.chk[i0 + l] : -1);
            o.pos.resize((size_t) off + (size_t) s.deg * G, (uint16_t) c.n);
            for (int l = 0; l < cnt; l++) {
                const int r = s.chk[i0 + l];
                for (int j = 0; j < s.deg; j++) o.pos[(size_t) off + (size_t) j * G + l] = (uint16_t) c.edge_var[c.row_ptr[r] + j];
            }
            off += s.deg * G;
            o.n_layers++;
        }
    }
    o.e_pad = off;
    if (o.n_layers == 0) {
        set_error("layered schedule: the matrix has no checks");
        return false;
    }
    if (qc && q.Z >= (1 << 15)) o.qc = false;
    return true;
}

namespace {

int admm_llen(const AdmmLayout &A, int i) { return A.var_ptr[i + 1] - A.var_ptr[i]; }

// members of group g in ASCENDING variable id (the order of the row phase of admm_block_kernel)
int admm_members(const AdmmLayout &A, int g, int out[3]) {
    const int ty = A.grp_type[g];
    for (int k = 0; k < ty; k++) out[k] = A.grp_var[(size_t) g * 3 + k];
    std::sort(out, out + ty);
    return ty;
}

// modelled LDS cycles of one frame-sweep for a given placement
void admm_model_cycles(const Code &c, AdmmBlockPlacement &P, bool f32) {
    const AdmmLayout &A = c.admm;
    const int L = P.L;
    P.cyc_u_reads = P.cyc_v_reads = P.cyc_v_writes = 0;
    P.ideal_u_reads = P.ideal_v_reads = P.ideal_v_writes = 0;
    for (int w = 0; w < 4; w++) P.wave_cost[w] = 0;
    std::vector<int> seen;  // distinct addresses per bank
    auto cycles_of = [&](std::vector<int> &addrs, int modulus) {  // addrs: word-granular addresses, -1 = lane inactive
        std::sort(addrs.begin(), addrs.end());
        addrs.erase(std::unique(addrs.begin(), addrs.end()), addrs.end());
        seen.assign(modulus, 0);
        int mx = 0;
        for (int a : addrs)
            if (a >= 0) mx = std::max(mx, ++seen[a % modulus]);
        return mx;
    };
    // v-update: lanes [32h, 32h+32) of (pass, wavefront) read entry k < ml(pass, wavefront) of their variables
    for (int p = 0; p < P.n_vpass; p++)
        for (int w = 0; w < L / 64 + (L % 64 ? 1 : 0); w++) {
            int ml = 0;
            for (int l = 64 * w; l < std::min(L, 64 * w + 64); l++) {
                const int i = P.var_of_slot[(size_t) p * L + l];
                if (i >= 0) ml = std::max(ml, admm_llen(A, i));
            }
            if (w < 4) P.wave_cost[w] += ml;
            for (int h = 0; h < 2; h++)
                for (int k = 0; k < ml; k++) {
                    std::vector<int> addrs;
                    for (int l = 64 * w + 32 * h; l < std::min(L, 64 * w + 32 * h + 32); l++) {
                        const int i = P.var_of_slot[(size_t) p * L + l];
                        if (i >= 0 && k < admm_llen(A, i)) addrs.push_back(P.slot_of_grp[A.var_grp[A.var_ptr[i] + k] >> 2]);
                        else addrs.push_back(P.pad_gslot[((size_t) p * L + l) * P.max_list + k]);  // padding entries read an all-zero slot
                    }
                    if (addrs.empty()) continue;
                    P.cyc_u_reads += cycles_of(addrs, 32);
                    P.ideal_u_reads += 1;
                }
        }
    // row phase: lanes of a 32-slot service group read member k of their groups (predicated off for empty slots)
    std::vector<int> grp_of((size_t) P.n_gpass * L, -1);
    for (int g = 0; g < A.n_grp; g++) grp_of[P.slot_of_grp[g]] = g;
    for (int base = 0; base < P.n_gpass * L; base += 32)
        for (int k = 0; k < 3; k++) {
            std::vector<int> addrs;
            for (int sl = base; sl < base + 32; sl++) {
                const int g = grp_of[sl];
                if (g < 0) continue;
                int mem[3];
                const int ty = admm_members(A, g, mem);
                addrs.push_back(k < ty ? P.cell_of_var[mem[k]] : P.zero_cell);
            }
            if (addrs.empty()) continue;
            P.cyc_v_reads += cycles_of(addrs, 32);
            P.ideal_v_reads += 1;
        }
    // v-update stores: ds_write_b64 is served 16 lanes at a time (16 bank pairs), ds_write_b32 32 lanes at a time
    const int wl = f32 ? 32 : 16;
    for (int base = 0; base < P.n_vpass * L; base += wl) {
        std::vector<int> addrs;
        for (int s = base; s < base + wl; s++)
            if (P.var_of_slot[s] >= 0) addrs.push_back(P.cell_of_var[P.var_of_slot[s]]);
        if (addrs.empty()) continue;
        P.cyc_v_writes += cycles_of(addrs, wl);
        P.ideal_v_writes += 1;
    }
}

// Padding reads (see AdmmBlockPlacement::pad_gslot).  spread: give every padded lane an unoccupied (all-zero) slot whose
// bank no real read of its 32-lane service group uses at that entry, so that padding never adds an LDS cycle.
void admm_assign_pads(const Code &c, AdmmBlockPlacement &P, bool spread) {
    const AdmmLayout &A = c.admm;
    const int L = P.L;
    P.max_list = 1;
    for (int i = 0; i < A.n_var; i++) P.max_list = std::max(P.max_list, admm_llen(A, i));
    P.pad_gslot.assign((size_t) P.n_vpass * L * P.max_list, -1);
    std::vector<std::vector<int>> free_of_bank(32);
    if (spread) {
        std::vector<char> used((size_t) P.n_gpass * L, 0);
        for (int g = 0; g < A.n_grp; g++) used[P.slot_of_grp[g]] = 1;
        for (int sl = 0; sl < P.u_slots; sl++)
            if (!used[sl]) free_of_bank[sl % 32].push_back(sl);
    }
    for (int p = 0; p < P.n_vpass; p++)
        for (int base = 0; base < L; base += 32)
            for (int k = 0; k < P.max_list; k++) {
                bool bank_used[32] = {};
                for (int l = base; l < std::min(L, base + 32); l++) {
                    const int i = P.var_of_slot[(size_t) p * L + l];
                    if (i >= 0 && k < admm_llen(A, i)) bank_used[P.slot_of_grp[A.var_grp[A.var_ptr[i] + k] >> 2] % 32] = true;
                }
                for (int l = base; l < std::min(L, base + 32); l++) {
                    const int i = P.var_of_slot[(size_t) p * L + l];
                    if (i >= 0 && k < admm_llen(A, i)) continue;
                    int pick = P.zero_gslot;
                    if (spread) {
                        // all padded lanes of the service group may share ONE address (a broadcast costs nothing extra):
                        // the first free bank that owns an unoccupied slot
                        for (int b = 0; b < 32; b++)
                            if (!bank_used[b] && !free_of_bank[b].empty()) {
                                pick = free_of_bank[b][0];
                                break;
                            }
                    }
                    P.pad_gslot[((size_t) p * L + l) * P.max_list + k] = pick;
                }
            }
}

// ---- mode 1: list-length order + simulated annealing of the two address maps (any code) ----------------------------
void admm_placement_annealed(const Code &c, AdmmBlockPlacement &P, bool f32, bool tune) {
    const AdmmLayout &A = c.admm;
    const int L = P.L;
    // variables sorted by list length (descending) so a pass has a uniform trip count
    std::vector<int> vorder(A.n_var);
    std::iota(vorder.begin(), vorder.end(), 0);
    std::stable_sort(vorder.begin(), vorder.end(), [&](int x, int y) { return admm_llen(A, x) > admm_llen(A, y); });
    P.var_of_slot.assign((size_t) P.n_vpass * L, -1);
    for (int s = 0; s < A.n_var; s++) P.var_of_slot[s] = vorder[s];
    // slots: three-variable checks first, then the one- and two-variable ones (their wavefronts run the GENERIC
    // instance of admm_group_update), then padding; slot n_grp is the all-zero slot list padding points to
    P.slot_of_grp.assign(A.n_grp, 0);
    {
        int sl = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (int g = 0; g < A.n_grp; g++)
                if ((A.grp_type[g] == 3) == (pass == 0)) P.slot_of_grp[g] = sl++;
            if (pass == 0) P.n3 = sl;
        }
    }
    P.zero_gslot = A.n_grp;
    P.u_slots = P.n_gpass * L;
    P.zero_cell = A.n_var;
    P.n_cells = A.n_var + 1;
    P.cell_of_var.resize(A.n_var);
    std::iota(P.cell_of_var.begin(), P.cell_of_var.end(), 0);
    if (tune && P.n3 > 1) {
        // v-update: lanes [32h, 32h+32) of (pass, wavefront) read entry k of their variables: bank = slot mod 32
        std::vector<int> item_of_grp(A.n_grp, -1), pos3;
        for (int g = 0; g < A.n_grp; g++)
            if (A.grp_type[g] == 3) {
                item_of_grp[g] = (int) pos3.size();
                pos3.push_back(P.slot_of_grp[g]);
            }
        std::vector<PlacementSet> sets;
        for (int p_ = 0; p_ < P.n_vpass; p_++) {
            const int maxlist = admm_llen(A, vorder[(size_t) p_ * L]);
            for (int h = 0; h < L / 32; h++)
                for (int k = 0; k < maxlist; k++) {
                    PlacementSet ps;
                    ps.modulus = 32;
                    for (int l = 32 * h; l < 32 * h + 32; l++) {
                        const int sidx = p_ * L + l;
                        if (sidx >= A.n_var) continue;
                        const int i = vorder[sidx];
                        if (k >= admm_llen(A, i)) continue;
                        const int g = A.var_grp[A.var_ptr[i] + k] >> 2;
                        if (item_of_grp[g] >= 0) ps.items.push_back(item_of_grp[g]);
                    }
                    if (ps.items.size() > 1) sets.push_back(std::move(ps));
                }
        }
        placement_optimise(pos3, P.n3, sets, 800);
        for (int g = 0; g < A.n_grp; g++)
            if (item_of_grp[g] >= 0) P.slot_of_grp[g] = pos3[item_of_grp[g]];
    }
    if (tune && A.n_var > 1) {
        std::vector<int> grp_of((size_t) P.n_gpass * L, -1);
        for (int g = 0; g < A.n_grp; g++) grp_of[P.slot_of_grp[g]] = g;
        std::vector<PlacementSet> sets;
        // row phase: lanes [32h, 32h+32) of a group pass read member k of their groups: bank = cell mod 32
        for (int base = 0; base < P.n_gpass * L; base += 32)
            for (int k = 0; k < 3; k++) {
                PlacementSet ps;
                ps.modulus = 32;
                for (int sl = base; sl < base + 32; sl++) {
                    const int g = grp_of[sl];
                    if (g < 0) continue;
                    int mem[3];
                    if (k < admm_members(A, g, mem)) ps.items.push_back(mem[k]);
                }
                if (ps.items.size() > 1) sets.push_back(std::move(ps));
            }
        const int wl = f32 ? 32 : 16;
        for (int base = 0; base < A.n_var; base += wl) {
            PlacementSet ps;
            ps.modulus = wl;
            for (int sidx = base; sidx < std::min(base + wl, A.n_var); sidx++) ps.items.push_back(vorder[sidx]);
            if (ps.items.size() > 1) sets.push_back(std::move(ps));
        }
        placement_optimise(P.cell_of_var, A.n_var, sets, 800);
    }
}

// ---- mode 2: tuples of a quasi-cyclic code ---------------------------------------------------------------------------
struct TupleSide {
    int n_items = 0;
    std::vector<std::vector<int>> refs;  // refs[item][k] = item of the OTHER side read by entry k
    std::vector<int> label;              // items may only trade places with items of the same label
    std::vector<int> pos;                // tuple position
    std::vector<std::vector<std::pair<int, int>>> readers;  // (item of the other side, k) that read this item
};

// Joint search over the tuple positions of both sides.  Position = svc * M + cls: a service group holds M tuples, the
// class of a tuple is its position inside the service group.  Energy = sum over (side, service group, k) of the squared
// class multiplicities of the tuples read — minimal (= number of reads) iff every service group reads M distinct classes.
// cost_unit: positions [u*cost_unit, (u+1)*cost_unit) form a wavefront pass whose v-update cost is its largest label;
// a move into an empty position is allowed only where it does not raise that maximum (side 0 only).
long tuple_joint_placement(TupleSide (&S)[2], const int n_pos, const int M, const int cost_unit, const int rounds) {
    std::vector<int> at[2];
    size_t kmax[2] = {0, 0};
    for (int s = 0; s < 2; s++) {
        at[s].assign(n_pos, -1);
        for (int i = 0; i < S[s].n_items; i++) {
            at[s][S[s].pos[i]] = i;
            kmax[s] = std::max(kmax[s], S[s].refs[i].size());
        }
        S[s].readers.assign(S[s].n_items, {});
    }
    for (int s = 0; s < 2; s++)
        for (int i = 0; i < S[s].n_items; i++)
            for (size_t k = 0; k < S[s].refs[i].size(); k++) S[1 - s].readers[S[s].refs[i][k]].push_back({i, (int) k});
    const int n_svc = (n_pos + M - 1) / M;
    std::vector<int> cnt[2];
    long total = 0;
    auto cell = [&](int s, int svc, int k, int cls) -> int & { return cnt[s][((size_t) svc * kmax[s] + k) * M + cls]; };
    auto add = [&](int s, int svc, int k, int cls, int sign) {
        int &cc = cell(s, svc, k, cls);
        if (sign > 0) {
            total += 2 * cc + 1;
            cc++;
        } else {
            total -= 2 * cc - 1;
            cc--;
        }
    };
    for (int s = 0; s < 2; s++) {
        cnt[s].assign((size_t) n_svc * std::max<size_t>(kmax[s], 1) * M, 0);
        for (int i = 0; i < S[s].n_items; i++)
            for (size_t k = 0; k < S[s].refs[i].size(); k++) add(s, S[s].pos[i] / M, (int) k, S[1 - s].pos[S[s].refs[i][k]] % M, +1);
    }
    auto move = [&](int s, int x, int np) {
        const int op = S[s].pos[x];
        if (op / M != np / M)  // as a reader: its entries change service group
            for (size_t k = 0; k < S[s].refs[x].size(); k++) {
                const int cls = S[1 - s].pos[S[s].refs[x][k]] % M;
                add(s, op / M, (int) k, cls, -1);
                add(s, np / M, (int) k, cls, +1);
            }
        if (op % M != np % M)  // as a target: its class changes for everybody who reads it
            for (const auto &r : S[s].readers[x]) {
                const int svc = S[1 - s].pos[r.first] / M;
                add(1 - s, svc, r.second, op % M, -1);
                add(1 - s, svc, r.second, np % M, +1);
            }
        S[s].pos[x] = np;
    };
    auto unit_max_without = [&](int s, int unit, int skip_pos) {
        int mx = 0;
        for (int q = unit * cost_unit; q < std::min(n_pos, (unit + 1) * cost_unit); q++)
            if (q != skip_pos && at[s][q] >= 0) mx = std::max(mx, S[s].label[at[s][q]]);
        return mx;
    };
    uint64_t rng = 0xA24BAED4963EE407ull;
    auto next = [&]() {
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        return rng;
    };
    const long moves = (long) rounds * (S[0].n_items + S[1].n_items);
    const long floor_e = [&] {  // energy of a conflict-free placement: every read counted once
        long e = 0;
        for (int s = 0; s < 2; s++)
            for (int i = 0; i < S[s].n_items; i++) e += (long) S[s].refs[i].size();
        return e;
    }();
    for (long mv = 0; mv < moves && total > floor_e; mv++) {
        const double T = 1.5 * std::pow(0.05 / 1.5, (double) mv / (double) moves);
        const int s = (int) (next() & 1);
        if (S[s].n_items < 1) continue;
        const int a = (int) (next() % (uint64_t) S[s].n_items);
        const int q = (int) (next() % (uint64_t) n_pos), pa = S[s].pos[a];
        if (q == pa) continue;
        const int b = at[s][q];
        if (b >= 0 && S[s].label[a] != S[s].label[b]) continue;
        if (b < 0 && q / cost_unit != pa / cost_unit) {
            if (s == 1) continue;  // group tuples keep their wavefront pass (the row phase stays balanced)
            if (unit_max_without(s, q / cost_unit, -1) < S[s].label[a]) continue;  // would lengthen that wavefront pass
        }
        const long before = total;
        move(s, a, q);
        if (b >= 0) move(s, b, pa);
        const long delta = total - before;
        if (delta > 0 && (double) (next() >> 11) * (1.0 / 9007199254740992.0) >= std::exp(-(double) delta / T)) {
            if (b >= 0) move(s, b, q);
            move(s, a, pa);
        } else {
            at[s][q] = a;
            at[s][pa] = b;
        }
    }
    return total - floor_e;
}

bool admm_placement_qc(const Code &c, AdmmBlockPlacement &P) {
    const AdmmLayout &A = c.admm;
    const int L = P.L;
    QcInfo qc;
    if (!code_detect_qc(c, qc)) return false;
    const int Z = qc.Z;
    const int g = std::gcd(Z, 32), q = Z / g;
    if (g < 2 || std::gcd(q, g) != 1 || L % g || 32 % g) return false;
    for (int gi = 0; gi < A.n_grp; gi++)
        if (A.grp_type[gi] != 3) return false;  // one- / two-variable checks: the annealed path handles them
    // ---- labels: group gi = (proto G, copy i), variable v = (proto Pv, copy j) ---------------------------------------
    // admm_layout_build walks the check rows in order; row (R, i) of degree d yields groups t = 0..d-3 and auxiliaries
    // t = 0..d-4, so the proto index is (block row, t) and the copy index the row inside the block.
    std::vector<int> g_proto(A.n_grp), g_copy(A.n_grp), v_proto(A.n_var, -1), v_copy(A.n_var, -1);
    std::vector<int> gbase(qc.mb + 1, 0), abase(qc.mb + 1, 0);
    for (int R = 0; R < qc.mb; R++) {
        const int d = c.row_ptr[R * Z + 1] - c.row_ptr[R * Z];
        if (d < 3) return false;
        gbase[R + 1] = gbase[R] + (d - 2);
        abase[R + 1] = abase[R] + (d - 3);
    }
    const int n_gproto = gbase[qc.mb], n_vproto = qc.nb + abase[qc.mb];
    {
        int gi = 0, aux = c.n;
        for (int r = 0; r < c.m; r++) {
            const int R = r / Z, i = r % Z, d = c.row_ptr[r + 1] - c.row_ptr[r];
            if (d != gbase[R + 1] - gbase[R] + 2) return false;
            for (int t = 0; t < d - 2; t++, gi++) {
                g_proto[gi] = gbase[R] + t;
                g_copy[gi] = i;
            }
            for (int t = 0; t < d - 3; t++, aux++) {
                v_proto[aux] = qc.nb + abase[R] + t;
                v_copy[aux] = i;
            }
        }
        if (gi != A.n_grp || aux != A.n_var) return false;
        for (int v = 0; v < c.n; v++) {
            v_proto[v] = v / Z;
            v_copy[v] = v % Z;
        }
    }
    // ---- tuples: copies {x : x mod q = cq}, ordered by x mod g -------------------------------------------------------
    const int n_vt = n_vproto * q, n_gt = n_gproto * q;
    auto vt_of = [&](int v) { return v_proto[v] * q + v_copy[v] % q; };
    auto gt_of = [&](int gi) { return g_proto[gi] * q + g_copy[gi] % q; };
    TupleSide S[2];  // side 0 = variable tuples (read group tuples, list entry k), side 1 = group tuples (read variable tuples, member k)
    S[0].n_items = n_vt;
    S[1].n_items = n_gt;
    S[0].refs.assign(n_vt, {});
    S[1].refs.assign(n_gt, {});
    S[0].label.assign(n_vt, 0);
    S[1].label.assign(n_gt, 0);
    // every member of a tuple must read the same tuple through entry k, at pairwise different offsets (x mod g):
    // that is what the cyclic structure promises; verified here member by member, any surprise -> annealed path
    for (int v = 0; v < A.n_var; v++) {
        const int tv = vt_of(v), len = admm_llen(A, v);
        if (S[0].refs[tv].empty() && len > 0) {
            S[0].refs[tv].assign(len, -1);
            S[0].label[tv] = len;
        }
        if ((int) S[0].refs[tv].size() != len) return false;
        for (int k = 0; k < len; k++) {
            const int gi = A.var_grp[A.var_ptr[v] + k] >> 2;
            if (S[0].refs[tv][k] < 0) S[0].refs[tv][k] = gt_of(gi);
            if (S[0].refs[tv][k] != gt_of(gi)) return false;
        }
    }
    for (int gi = 0; gi < A.n_grp; gi++) {
        const int tg = gt_of(gi);
        int mem[3];
        admm_members(A, gi, mem);
        if (S[1].refs[tg].empty()) S[1].refs[tg].assign(3, -1);
        for (int k = 0; k < 3; k++) {
            if (S[1].refs[tg][k] < 0) S[1].refs[tg][k] = vt_of(mem[k]);
            if (S[1].refs[tg][k] != vt_of(mem[k])) return false;
        }
    }
    // offsets inside a tuple: the g members read g different offsets of the target tuple
    {
        std::vector<int> mask;
        mask.assign((size_t) n_vt * 8, 0);
        for (int v = 0; v < A.n_var; v++)
            for (int k = 0; k < admm_llen(A, v) && k < 8; k++) {
                const int gi = A.var_grp[A.var_ptr[v] + k] >> 2;
                int &mk = mask[(size_t) vt_of(v) * 8 + k];
                if (mk & (1 << (g_copy[gi] % g))) return false;
                mk |= 1 << (g_copy[gi] % g);
            }
        for (int v = 0; v < A.n_var; v++)
            if (admm_llen(A, v) > 8) return false;
        mask.assign((size_t) n_gt * 3, 0);
        for (int gi = 0; gi < A.n_grp; gi++) {
            int mem[3];
            admm_members(A, gi, mem);
            for (int k = 0; k < 3; k++) {
                int &mk = mask[(size_t) gt_of(gi) * 3 + k];
                if (mk & (1 << (v_copy[mem[k]] % g))) return false;
                mk |= 1 << (v_copy[mem[k]] % g);
            }
        }
    }
    // ---- tuple positions ---------------------------------------------------------------------------------------------
    const int M = 32 / g;            // tuples per 32-lane service group = number of classes
    const int per_pass = L / g;      // tuple positions per pass
    const int unit = 64 / g;         // tuple positions per (pass, wavefront)
    const int n_pass = std::max(P.n_vpass, P.n_gpass);
    const int n_pos = n_pass * per_pass;
    if (n_vt > P.n_vpass * per_pass || n_gt + 1 > P.n_gpass * per_pass || L % 64) return false;
    // start: variable tuples sorted by list length, cut into wavefront-pass units, units dealt to the wavefronts
    // longest first onto the least loaded wavefront (its passes fill in order)
    {
        std::vector<int> order(n_vt);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return S[0].label[x] > S[0].label[y]; });
        const int waves = L / 64;
        std::vector<int> load(waves, 0), used(waves, 0);
        S[0].pos.assign(n_vt, -1);
        for (int u0 = 0; u0 < n_vt; u0 += unit) {
            int best = -1;
            for (int w = 0; w < waves; w++)
                if (used[w] < P.n_vpass && (best < 0 || load[w] < load[best])) best = w;
            if (best < 0) return false;
            const int pass = used[best]++;
            load[best] += S[0].label[order[u0]];
            for (int x = u0; x < std::min(n_vt, u0 + unit); x++) S[0].pos[order[x]] = pass * per_pass + best * unit + (x - u0);
        }
        // group tuples: pass-major.  The last (partial) pass fills the first wavefronts, so that the U array the kernel
        // keeps in LDS can end right behind it (576 instead of 768 slots for H05: a sixth workgroup per CU fits)
        S[1].pos.assign(n_gt, -1);
        int x = 0;
        for (int pass = 0; pass < P.n_gpass && x < n_gt; pass++)
            for (int w = 0; w < waves && x < n_gt; w++)
                for (int o = 0; o < unit && x < n_gt; o++, x++) S[1].pos[x] = pass * per_pass + w * unit + o;
    }
    // group tuples carry no label constraint except "stay in the same pass occupancy": label = pass so the row phase
    // keeps its per-wavefront pass counts; variable tuples: label = list length
    for (int t = 0; t < n_gt; t++) S[1].label[t] = 0;
    const long residual = tuple_joint_placement(S, n_pos, M, unit, 10000);
    if (getenv("ACG_ADMM_PLACEMENT_DEBUG"))
        fprintf(stderr, "[acg_ldpc] QC placement: Z=%d tuple=%d, %d variable / %d group tuples on %d positions, residual conflict energy %ld\n",
                Z, g, n_vt, n_gt, n_pos, residual);
    // ---- back to threads -------------------------------------------------------------------------------------------------
    P.var_of_slot.assign((size_t) P.n_vpass * L, -1);
    P.cell_of_var.assign(A.n_var, 0);
    P.slot_of_grp.assign(A.n_grp, 0);
    for (int v = 0; v < A.n_var; v++) {
        const int slot = S[0].pos[vt_of(v)] * g + v_copy[v] % g;
        if (slot >= P.n_vpass * L || P.var_of_slot[slot] >= 0) return false;
        P.var_of_slot[slot] = v;
        P.cell_of_var[v] = slot;  // V cell = thread slot: consecutive lanes store consecutive words
    }
    std::vector<char> used((size_t) P.n_gpass * L, 0);
    for (int gi = 0; gi < A.n_grp; gi++) {
        const int slot = S[1].pos[gt_of(gi)] * g + g_copy[gi] % g;
        if (slot >= P.n_gpass * L || used[slot]) return false;
        used[slot] = 1;
        P.slot_of_grp[gi] = slot;
    }
    // U slots the kernel has to hold: everything up to the end of the last wavefront-pass unit that holds a group, which
    // must also offer free (all-zero) slots for the list padding
    int top = 0;
    for (int sl = 0; sl < P.n_gpass * L; sl++)
        if (used[sl]) top = sl;
    P.u_slots = std::min(P.n_gpass * L, (top / 64 + 1) * 64);
    P.zero_gslot = -1;
    for (int sl = P.u_slots - 1; sl >= 0 && P.zero_gslot < 0; sl--)
        if (!used[sl]) P.zero_gslot = sl;
    if (P.zero_gslot < 0) {  // the last unit is full: one more tile
        if (P.u_slots + 32 > P.n_gpass * L) return false;
        P.zero_gslot = P.u_slots;
        P.u_slots += 32;
    }
    P.zero_cell = P.n_vpass * L;
    P.n_cells = P.n_vpass * L + 1;
    P.n3 = 0;
    P.qc = true;
    P.Z = Z;
    P.tuple = g;
    return true;
}

}  // namespace

bool admm_block_placement(const Code &c, int L, bool f32, int mode, AdmmBlockPlacement &out) {
    const AdmmLayout &A = c.admm;
    // the placement depends on the matrix only (not on alpha / mu / iteration counts): one search per matrix and process
    // (the (alpha, mu) grid search creates 3721 decoders for one H, qpadmm_params.cpp:64-77)
    static std::mutex cache_mu;
    static std::map<std::array<uint64_t, 2>, AdmmBlockPlacement> cache;
    uint64_t h = 1469598103934665603ull;
    for (int v : c.edge_var) h = (h ^ (uint64_t) (uint32_t) v) * 1099511628211ull;
    for (int v : c.row_ptr) h = (h ^ (uint64_t) (uint32_t) v) * 1099511628211ull;
    const std::array<uint64_t, 2> key = {h, ((uint64_t) c.m << 40) ^ ((uint64_t) c.n << 16) ^ ((uint64_t) L << 4) ^ ((uint64_t) mode << 1) ^ (f32 ? 1u : 0u)};
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        auto it = cache.find(key);
        if (it != cache.end()) {
            out = it->second;
            return true;
        }
    }
    out = AdmmBlockPlacement();
    out.L = L;
    out.n_gpass = (A.n_grp + 1 + L - 1) / L;  // +1: at least one padding slot that stays all-zero
    out.n_vpass = (A.n_var + L - 1) / L;
    bool done = false;
    if (mode >= 2) done = admm_placement_qc(c, out);
    if (!done) {
        AdmmBlockPlacement fresh;
        fresh.L = L;
        fresh.n_gpass = out.n_gpass;
        fresh.n_vpass = out.n_vpass;
        out = fresh;
        admm_placement_annealed(c, out, f32, mode >= 1);
    }
    admm_assign_pads(c, out, done);
    admm_model_cycles(c, out, f32);
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        if (cache.size() >= 64) cache.clear();  // bounded: a local search over matrices must not grow it without limit
        cache[key] = out;
    }
    return true;
}

}  // namespace acg
