// Device-facing argument blocks shared by the .hip kernels and the host API.  Not part of the ABI.
#pragma once

#include <cstdint>

namespace acg {

// Device copies of BpLayout (see ldpc_internal.hpp for the meaning of each table).
struct BpTables {
    const int32_t *c_pass;     // [n_cpass][2] = {max degree in pass, A offset of pass}
    const int32_t *c_cnt_ge;   // [34] number of checks with degree >= d (zero padded)
    const int32_t *v_pass;     // [n_vpass][2] = {max degree in pass, index-table offset of pass}
    const int32_t *v_cnt_ge;   // [34]
    const int32_t *v_var;      // [n_vpass*L] variable id per slot (-1 none)
    const uint16_t *v_apos;    // [v_apos_len] A word of (pass, k, lane)
    int32_t v_apos_len;
    int32_t idx_lds_bytes;     // bytes of the block-shared LDS copy of v_apos (0 = read it from global)
    int32_t n_cpass, n_vpass;
    int32_t a_words, zero_pos;
    int32_t m, n, nwords;      // nwords = (n+31)/32 packed output words per frame
    int32_t llr_words;         // n_vpass * L
    int32_t lds_bytes_per_frame;
};

// per-launch MC statistics (one row per launch; experiment.h:25-68)
enum { MC_CORRECT = 0, MC_PSEUDO, MC_TOTAL, MC_HAM, MC_HAM_OK, MC_HAM_WRONG, MC_ITERS, MC_NCOUNTERS };

struct DecodeArgs {
    // input
    const void *y;      // frames*n float or double (null in MC/device-noise mode)
    int32_t y_is_f64;
    int64_t frames;
    double inv_var2;    // 2 / sigma^2  (llr = y * inv_var2; fp64 path divides exactly like channel.h:14-16)
    double var;         // sigma^2
    // output (any may be null in MC mode)
    uint32_t *out_bits;
    uint8_t *out_ok;
    int32_t *out_iters;
    // params
    int32_t max_iter;
    int32_t early_exit;
    float ms_scale;
    // Monte-Carlo mode
    int32_t mc;          // 0 = decode y; 1 = generate y on device + classify
    uint64_t seed;
    int64_t first_frame;
    const uint32_t *cw_packed;  // n_cw * nwords, null = all-zero codeword
    int64_t n_cw;
    float sigma;
    unsigned long long *counters;  // MC_NCOUNTERS
    unsigned long long *work_counter;  // fused BP: next unassigned frame (zeroed before every launch)
    // diagnostics (streamed engine, first 64-frame tile only): raw message words after the LAST executed
    // check sweep / variable sweep, [E][64], and posteriors [n][64]; null = off
    void *dbg_c2v;
    void *dbg_v2c;
    void *dbg_post;
};

// Layered min-sum (bp_layered.hip; LayeredLayout in ldpc_internal.hpp)
struct LayerTables {
    const int32_t *layer;    // [n_layers][4] = {degree, message offset (words), checks, first proto entry | first row << 16}
    const int32_t *proto;    // quasi-cyclic H: {block column, shift} pairs per block row; null = use pos
    const int32_t *proto_packed;  // the same, one word per pair: block column * Z * 4 << 16 | shift * 4 (byte units, for the hot loop)
    const uint16_t *pos;     // any other H: [e_pad] variable of (layer, edge, lane), n = neutral cell; null when proto is set
    int32_t n_layers, Z, n, nwords;
    int32_t e_pad;           // message words per frame
    int32_t p_words;         // posterior words per frame (n + the neutral cell, rounded up)
    int32_t r_words;         // 32-bit words the e_pad messages of a frame occupy (e_pad, or half of it rounded up for fp16 storage)
    int32_t tab_lds_bytes;   // bytes of the workgroup's position table
    int32_t lds_bytes_per_frame;
};

// Streamed ("HBM") BP engine: plain CSR of the Tanner graph, read through scalar loads.
struct StreamTables {
    const int32_t *row_ptr;   // [m+1] edges in check-major order (variables ascending)
    const int32_t *col_ptr;   // [n+1]
    const int32_t *col_edge;  // [E] edge ids per variable (checks ascending)
    int32_t m, n, E, nwords;
    int64_t ws_words_per_wave;  // workspace words (of 4 bytes) per wavefront
    // LDS-DMA ring engine (bp_streamed_ring_kernel, fp32): the sweeps cut into tasks of at most RING_SLOT_LINES message
    // lines; task i of a sweep belongs to wavefront i mod RING_WAVES.  int4 per task:
    //   check task {first check, checks, first line, lines | wait << 8 | wait_nostore << 16}
    //   variable task {first variable, variables, first col_ptr entry, edge lines | wait << 8 | wait_nostore << 16}
    // wait = vector-memory operations GUARANTEED to be issued behind the task's loads when its data is needed (see the kernel)
    const int32_t *ctask;
    const int32_t *vtask;
    const int32_t *vtask_of_word;  // [nwords] first variable task holding a variable >= 32 * word
    int32_t n_ctask, n_vtask;
};
#ifndef ACG_RING_SLOTS
#define ACG_RING_SLOTS 3
#endif
#ifndef ACG_RING_SLOT_LINES
#define ACG_RING_SLOT_LINES 16
#endif
#ifndef ACG_RING_MAX_PER_CU
#define ACG_RING_MAX_PER_CU 4
#endif
constexpr int RING_WAVES = 4;        // wavefronts per workgroup of the ring engine
constexpr int RING_SLOTS = ACG_RING_SLOTS;            // ring slots per wavefront (tasks in flight: RING_SLOTS - 1 ahead of the one computed)
constexpr int RING_SLOT_LINES = ACG_RING_SLOT_LINES;  // 256-byte lines per slot
constexpr int RING_VAR_EDGE_LINES = RING_SLOT_LINES - 4;  // variable task: edge lines + up to 4 LLR lines
constexpr int RING_MAX_CDEG = RING_SLOT_LINES < 16 ? RING_SLOT_LINES : 16;          // a node must fit in one slot
constexpr int RING_MAX_VDEG = RING_VAR_EDGE_LINES < 12 ? RING_VAR_EDGE_LINES : 12;  // (and in the kernels' degree switches)
constexpr int RING_LDS_BYTES = RING_WAVES * RING_SLOTS * RING_SLOT_LINES * 256;
// the counted waits must stay within the 6-bit vmcnt field: (slots - 1) tasks of loads + stores behind a task's loads
static_assert((RING_SLOTS - 1) * (RING_SLOT_LINES + RING_SLOT_LINES / 4 + 1) <= 63, "ring too deep for vmcnt");

}  // namespace acg
