// Internal host-side structures shared by code.cpp (graph analysis, no HIP) and the HIP
// translation units.  Not part of the ABI.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace acg {

void set_error(const std::string &msg);

// Layout of one frame's message state for a given number of cooperating lanes L.
//
// Check side ("A layout"): checks sorted by degree (descending, stable), dealt L per pass.
// The j-th edge (variables ascending, the order of bp.h:144-147) of the check in slot (p, l)
// lives at word  c_off[p] + j*L + l : every check-phase access is a unit-stride, bank-conflict
// free LDS access with no index table.  Slots j >= degree are padding and hold +0.0 forever
// (neutral for the phi-sum, the sign product and the syndrome bit).
//
// Variable side: variables sorted by degree (descending, stable), L per pass.  The k-th edge
// (checks ascending) of the variable in slot (p, l) is found through v_apos[v_idx_off[p] + k*L + l],
// the A-layout word of that edge; padding entries point at `zero_pos`, a word that is always +0.0.
// Messages are updated IN PLACE: a word holds v->c before the check phase and c->v after it.
struct BpLayout {
    int L = 0;
    int n_cpass = 0, n_vpass = 0;
    int a_words = 0;  // message words incl. the zero cell (rounded up to a multiple of 4)
    int zero_pos = 0;
    int max_cdeg = 0, max_vdeg = 0;
    std::vector<int32_t> c_maxdeg, c_off;   // [n_cpass]
    std::vector<int32_t> c_cnt_ge;          // [max_cdeg+2]: number of checks with degree >= d
    std::vector<int32_t> c_chk;             // [n_cpass*L] check id per slot, -1 = none
    std::vector<int32_t> v_maxdeg, v_idx_off;  // [n_vpass]
    std::vector<int32_t> v_cnt_ge;          // [max_vdeg+2]
    std::vector<int32_t> v_var;             // [n_vpass*L] variable id per slot, -1 = none
    std::vector<uint16_t> v_apos;           // [sum_p v_maxdeg[p]*L]
    int v_apos_len = 0;
};

// QP-ADMM problem structure (qp_admm.h:13-102) regrouped by "constraint group": one group per
// three-variable check (4 rows, qp_admm.h:34-57), per degree-2 check (2 rows, :75-83) or per
// degree-1 check (1 row, :70-74).  Rows of a group only touch the group's <=3 variables, so the
// per-row state never leaves the lane that owns the group.
struct AdmmLayout {
    int n = 0, n_var = 0, n_con = 0, nnz = 0, n_grp = 0;
    double e_min = 0, e_max = 0;
    std::vector<int32_t> grp_var;   // [n_grp*3] variable ids (-1 = unused)
    std::vector<uint8_t> grp_type;  // 3 / 2 / 1
    std::vector<double> e;          // [n_var] = sum_j A_ji^2
    // per variable: list of (group, position-in-group) in construction order
    std::vector<int32_t> var_ptr;   // [n_var+1]
    std::vector<int32_t> var_grp;   // [sum] group*4 + position
};

struct Code {
    int m = 0, n = 0, E = 0;
    std::vector<uint8_t> H;          // dense m*n
    std::vector<int32_t> row_ptr, edge_var;   // CSR by check (variables ascending)
    std::vector<int32_t> col_ptr, col_edge;   // CSR by variable (checks ascending) -> edge id
    int max_cdeg = 0, max_vdeg = 0;
    AdmmLayout admm;
};

bool code_build(Code &c, const uint8_t *H, int m, int n);
bool code_read_txt(const char *path, std::vector<uint8_t> &H, int &m, int &n);
bool code_write_txt(const Code &c, const char *path);
bool code_generator(const Code &c, uint8_t *G);
bool code_is_codeword(const Code &c, const uint8_t *bits);
bool bp_layout_build(const Code &c, int L, BpLayout &out);
void admm_layout_build(Code &c);

// Static placement against LDS bank conflicts.  A wave64 LDS access is served in fixed lane groups and takes as many
// LDS cycles as the busiest bank has distinct addresses (MI355X_MICROARCH.md, LDS).  Which group slot / variable cell a
// lane touches in every instruction of the sweep is known when the decoder is created, so the two free permutations
// (constraint group -> U slot, variable -> V cell) are chosen to spread every lane group over the banks:
// items = groups or variables, position = slot or cell, a "set" = the items one lane group touches with one
// instruction, bank class = position mod `modulus`.  Objective: sum over sets of sum over classes of count^2
// (minimal when every class is hit at most once); deterministic simulated annealing over swaps of two positions.
struct PlacementSet {
    std::vector<int> items;
    int modulus;
};
long placement_optimise(std::vector<int> &pos_of_item, int n_pos, const std::vector<PlacementSet> &sets, int rounds,
                        std::vector<long> *per_set_max = nullptr);

long placement_optimise_gather(std::vector<int> &item_at_slot, const std::vector<int> &item_label, std::vector<int> &reader_at_pos,
                               const std::vector<int> &reader_label, const std::vector<std::vector<int>> &items_of_reader, int lanes,
                               int modulus, int rounds, long *cycles_before = nullptr, long *cycles_after = nullptr);

// ---- quasi-cyclic structure -----------------------------------------------------------------------------
// H as an (m/Z) x (n/Z) array of Z x Z blocks, each zero or a cyclic-shift permutation: row k of block (R, C) has its
// one at column (k + shift) mod Z — the reference's PermutationsMatrix (optimize_H.cpp:27-63); data/H05.txt and
// data/optimalH.txt are 8 x 14 arrays of 20 x 20 such blocks.
struct QcInfo {
    int Z = 0, mb = 0, nb = 0;
    std::vector<int> shift;  // [mb*nb], -1 = zero block
};
// largest Z >= 2 for which the matrix has this form (false: none)
bool code_detect_qc(const Code &c, QcInfo &q);

// ---- task tables of the LDS-DMA ring engine (bp_streamed_ring_kernel) ---------------------------------------------------
// The sweeps cut into tasks of at most RING_SLOT_LINES message lines; task i of a sweep belongs to wavefront i mod RING_WAVES.
// int4 per task (see StreamTables in kernels.hpp); the counted waits are what the kernel hands to s_waitcnt vmcnt(N):
// N = vector-memory operations CERTAINLY issued behind the task's loads when its data is needed.
struct RingTasks {
    std::vector<int32_t> ctask, vtask, vtask_of_word;
    int n_ctask = 0, n_vtask = 0;
};
void ring_tasks_build(const Code &c, RingTasks &out);

// ---- layered schedule for min-sum (bp_layered.hip) --------------------------------------------------------
// A layer is a set of at most G checks of ONE degree that share no variable, handled by the G lanes of a frame group in one
// step; the posteriors are updated in place after every layer.  Quasi-cyclic H: the layers are the block rows (a block
// row of cyclic-shift blocks touches every variable at most once and all its checks have the degree "non-zero blocks of the
// row"), cut into chunks of G when Z > G, and the device needs nothing but the (block column, shift) list of every block
// row — edge j of check (R, k) is variable C_j*Z + (k + s_j) mod Z.  Any other H: greedy colouring of the checks into
// conflict-free same-degree sets, with an explicit position table.
struct LayeredLayout {
    int G = 0;                 // lanes per frame (16, 20, 32 or 64)
    int n_layers = 0;
    bool qc = false;
    int Z = 0;
    int e_pad = 0;             // message words per frame: G * sum of the layers' degrees
    std::vector<int32_t> layer;      // [n_layers][4] = {degree, message offset (words), checks in the layer, first proto entry | first row << 16}
    std::vector<int32_t> proto;      // QC: per block row in use, degree x {block column, shift}
    std::vector<int32_t> chk;        // [n_layers*G] check id of (layer, lane), -1 = none
    std::vector<uint16_t> pos;       // [e_pad] variable of edge j of (layer, lane) at layer offset + j*G + lane (n = the neutral cell)
};
bool bp_layered_build(const Code &c, LayeredLayout &out);

// ---- placement of the QP-ADMM problem on the threads / LDS of admm_block_kernel ------------------------------------
// One workgroup of L threads owns a frame: thread l handles, in pass p, the variable var_of_slot[p*L + l] (v-update) and
// the constraint group living in U slot p*L + l (row phase).  LDS: V[cell] (fp64/fp32 words) and U[slot][4 rows], tiled so
// that the bank of an access is `slot mod 32` / `cell mod 32`.  A wavefront's LDS instruction is served 32 lanes at a
// time (ds_read_b64 / ds_read_b32) or 16 (ds_write_b64) and costs as many cycles as its busiest bank has distinct
// addresses, so the three free maps (variable -> thread slot, group -> U slot, variable -> V cell) decide the cost:
//   mode 1 (any code)  variables sorted by list length; group slots and V cells by simulated annealing over the
//                      modelled cycles (placement_optimise);
//   mode 2 (QC codes)  constructive: with g = gcd(Z, 32) and gcd(Z/g, g) = 1, the g copies {i : i mod (Z/g) = c} of a
//                      proto-variable / proto-group ("tuple") sit in g consecutive lanes ordered by i mod g.  A cyclic
//                      shift permutes the members of a tuple among themselves in the low log2(g) bank bits, so a
//                      tuple's accesses never collide with each other and the problem shrinks to placing tuples:
//                      32/g tuples per 32-lane service group, whose referenced tuples must sit in distinct classes
//                      (tuple position mod 32/g).  V cell = thread slot of the variable, U slot = thread slot of the
//                      group, so the stores are conflict-free by construction; the tuple positions are found by a joint
//                      search over both sides that starts from list-length-balanced wavefronts.
struct AdmmBlockPlacement {
    int L = 0, n_gpass = 0, n_vpass = 0;
    int n_cells = 0;      // V cells incl. the zero cell
    int zero_cell = 0;    // a cell that is always 0.0 (absent members of one- / two-variable checks)
    int zero_gslot = 0;   // a U slot that is always all-zero (list padding)
    int u_slots = 0;      // U slots the kernel must hold: every occupied / zero / padding slot is below it (multiple of 32)
    std::vector<int> var_of_slot;  // [n_vpass*L] variable id or -1
    std::vector<int> slot_of_grp;  // [n_grp]
    std::vector<int> cell_of_var;  // [n_var]
    // Entry k of a lane whose own list is shorter than the longest list of its (pass, wavefront) still issues the read:
    // pad_gslot[(p*L + l) * max_list + k] is the all-zero U slot it reads (-1: the lane has a real entry there).  Mode 1:
    // always zero_gslot; mode 2: an unoccupied slot in a bank none of the real reads of that service group uses.
    int max_list = 0;
    std::vector<int> pad_gslot;
    int n3 = 0;           // mode 1: groups in slots [0, n3) are three-variable checks, the rest one-/two-variable ones
    bool qc = false;
    int Z = 0, tuple = 1;
    // modelled LDS cycles per frame-sweep (one per 32-/16-lane service group and instruction; U reads count one row)
    long cyc_u_reads = 0, cyc_v_reads = 0, cyc_v_writes = 0;
    long ideal_u_reads = 0, ideal_v_reads = 0, ideal_v_writes = 0;  // the same with no bank conflict at all
    int wave_cost[4] = {0, 0, 0, 0};  // v-update: sum over passes of the longest list in (pass, wavefront)
};
bool admm_block_placement(const Code &c, int L, bool f32, int mode, AdmmBlockPlacement &out);

}  // namespace acg
