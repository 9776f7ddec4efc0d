// Host-side code analysis: parity-matrix text format, Tanner-graph CSR, the LDS message layout
// used by the fused BP kernels, the QP-ADMM constraint groups, and the GF(2) helpers the
// Monte-Carlo callers need.  Plain C++ (no HIP).  Runs once per H — the reference redoes the
// equivalent work for every frame (bp.h:136-153, qp_admm.h:13-102).
#include "ldpc_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

}  // namespace acg
