// Workgroup-per-frame fused BP — the same sweeps as bp_fused_kernel (bp_core.inc) with L = 256 or 1024 lanes per frame:
// for codes whose message array is too large for a useful number of wavefront-sized frames per CU (E above ~3k edges)
// but still fits in LDS (E up to ~36k: the 5000 x 10000 (3,6) code of BASELINE configs[4] needs 120 KB).  One frame per
// workgroup, all state in LDS for the whole decode, sweeps separated by __syncthreads(), frames handed out dynamically.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"

// IDXREG: the variable-side index table in registers (variable degree <= 4, table too large for LDS): see var_phase_regs
// DBG: tests only, see bp_fused_kernel
template <typename T, int L, int ALGO, bool MC, bool IDXLDS, bool IDXREG = false, bool DBG = false, bool REG = false>
__global__ void __launch_bounds__(L) bp_block_kernel(const BpTables t, const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MAXD = 8;
    constexpr int NVP = 12;
    using Core = BpCore<T, MAXD, L, ALGO, IDXLDS, NVP>;
    __shared__ unsigned long long fr_lds;
    __shared__ int ham_lds;
    const int l = threadIdx.x;
    const uint16_t *IDX = t.v_apos;
    if (IDXLDS) {
        uint16_t *idx_lds = reinterpret_cast<uint16_t *>(smem);
        for (int i = l; i < t.v_apos_len; i += L) idx_lds[i] = t.v_apos[i];
        IDX = idx_lds;
    }
    T *A = reinterpret_cast<T *>(smem + t.idx_lds_bytes);
    uint32_t *OB = reinterpret_cast<uint32_t *>(A + t.a_words);
    Core core(t, A, A, OB, IDX, l, a.ms_scale);
    typename Core::LlrRegs lr;
    typename Core::VarIds vi;
    typename Core::IdxRegs ir;
#pragma unroll
    for (int p = 0; p < NVP; ++p) {
        lr[p] = (T) 0;
        vi[p] = (p < t.n_vpass) ? t.v_var[p * L + l] : -1;
    }
    if (IDXREG) core.load_idx_regs(ir);
    unsigned long long acc_correct = 0, acc_pseudo = 0, acc_total = 0, acc_ham = 0, acc_ham_ok = 0, acc_ham_wrong = 0, acc_iters = 0;
    // padding words and the zero cell are +0.0 for the whole launch
    for (int w = l; w < t.a_words; w += L) A[w] = (T) 0;
#ifdef ACG_BLOCK_STAMPS
    unsigned long long st_chk = 0, st_b1 = 0, st_var = 0, st_b2 = 0, st_n = 0;
#endif

    for (;;) {
        __syncthreads();
        if (l == 0) {
            fr_lds = atomicAdd(a.work_counter, 1ull);
            ham_lds = 0;
        }
        __syncthreads();
        const int64_t frame = (int64_t) fr_lds;
        if (frame >= a.frames) break;
        const int64_t gf = a.first_frame + frame;
        const uint32_t *cw = (MC && a.cw_packed) ? a.cw_packed + (size_t) (gf % a.n_cw) * t.nwords : nullptr;
        // ---- start of a frame: symbols -> LLRs (registers) -> first v->c sweep ------------------------------
        if (MC) {
            const int nq = (t.n + 3) >> 2;
            for (int q = l; q < nq; q += L) {
                uint32_t r[4];
                philox4x32_10((uint32_t) gf, (uint32_t) (gf >> 32), (uint32_t) q, 0u, (uint32_t) a.seed,
                              (uint32_t) (a.seed >> 32), r);
                float z[4];
                box_muller(r[0], r[1], z[0], z[1]);
                box_muller(r[2], r[3], z[2], z[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = 4 * q + e;
                    if (v < t.n) {
                        const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                        A[v] = (T) __builtin_fmaf(a.sigma, z[e], bit ? -1.0f : 1.0f);  // channel.h:24 (explicit fma: same symbol in every kernel)
                    }
                }
            }
            __syncthreads();
        }
        int my_ham = 0;
        for (int p = 0; p < t.n_vpass; ++p) {
            const int slot = p * L + l;
            const int v = core.var_id(vi, p);
            T llr = (T) 0;
            if (v >= 0) {
                if (MC) {
                    const T yv = A[v];
                    const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                    my_ham += ((!bit && yv <= (T) 0) || (bit && yv > (T) 0)) ? 1 : 0;  // experiment.h:33-46
                    llr = (T) ((double) yv * (a.inv_var2 * Dom<T>::scale));
                } else if (a.y_is_f64) {
                    llr = (T) (2 * reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + v] / a.var * Dom<T>::scale);
                } else {
                    llr = (T) ((double) reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + v] * (a.inv_var2 * Dom<T>::scale));
                }
            }
            core.set_llr(lr, p, slot, llr);
        }
        if (MC) {
            my_ham = group_sum<64>(my_ham, l & 63);
            if ((l & 63) == 0) atomicAdd(&ham_lds, my_ham);
            __syncthreads();  // staged symbols consumed
            for (int w = l; w < ((t.n + 3) & ~3); w += L) A[w] = (T) 0;
        }
        __syncthreads();
        core.var_init_phase(lr, true);  // bp.h:184 on empty mailboxes
        __syncthreads();
        // ---- sweeps (bp.h:186-197) -----------------------------------------------------------------------------
#ifdef ACG_BLOCK_STAMPS
        st_chk = st_b1 = st_var = st_b2 = st_n = 0;
#endif
        int it = 0;
        bool latched = false;
        // decode instances (MERGED): the syndrome comes out of the check sweep and the hard decisions of my variables live
        // in a register bit mask (bit per pass); the Monte-Carlo instances keep the separate syndrome pass
        constexpr bool MERGED = !MC;
        uint32_t hard = MERGED ? core.llr_hard_mask(lr) : 0u;
        for (;;) {
            bool bad;
            if (MERGED) {
                // the check sweep reads every v->c word anyway: it also delivers the syndrome of the hard bits riding in
                // them (no separate syndrome pass, one barrier less per sweep).  Its c->v output is wasted on the last trip.
#ifdef ACG_BLOCK_STAMPS
                const unsigned long long t0 = __builtin_readcyclecounter();
                const bool myb = core.template check_phase_syndrome<REG>(true);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long t1 = __builtin_readcyclecounter();
                bad = __syncthreads_or(myb ? 1 : 0) != 0;
                const unsigned long long t2 = __builtin_readcyclecounter();
                st_chk += t1 - t0;
                st_b1 += t2 - t1;
#else
                bad = __syncthreads_or(core.template check_phase_syndrome<REG>(true) ? 1 : 0) != 0;
#endif
            } else {
                bad = __syncthreads_or(core.syndrome_bad() ? 1 : 0) != 0;
            }
            const bool conv = it > 0 && it <= a.max_iter && !bad;  // bp.h:195 (max_iter = 0: never)
            const bool out_now = conv && !latched;
            const bool finish = (a.early_exit && conv) || it >= a.max_iter;
            const bool fail_now = finish && !conv && !latched;
            if (out_now || fail_now) {  // block-uniform
                if (out_now) {
                    if (MERGED) core.pack_bits_mask(hard, vi);
                    else core.pack_bits(lr, vi);
                } else {
                    for (int w = l; w < t.nwords; w += L) OB[w] = 0u;  // bp.h:198
                    __syncthreads();
                }
                if (a.out_bits)
                    for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) frame * t.nwords + w] = OB[w];
                if (l == 0) {
                    if (a.out_ok) a.out_ok[frame] = out_now ? 1 : 0;
                    if (a.out_iters) a.out_iters[frame] = it < a.max_iter ? it : a.max_iter;
                }
                if (MC) {
                    bool neq = false;
                    for (int w = l; w < t.nwords; w += L) neq |= (OB[w] != (cw ? cw[w] : 0u));
                    const bool differ = __syncthreads_or(neq ? 1 : 0) != 0;
                    if (l == 0) {
                        const bool correct = out_now && !differ;  // experiment.h:110-116
                        const int ham = ham_lds;
                        acc_correct += correct;
                        acc_pseudo += (out_now && differ);
                        acc_total += 1;
                        acc_ham += ham;
                        acc_ham_ok += correct ? ham : 0;
                        acc_ham_wrong += correct ? 0 : ham;
                        acc_iters += it < a.max_iter ? it : a.max_iter;
                    }
                }
                latched = true;
            }
            if (finish) break;
            if (DBG && MERGED && it == a.max_iter - 1 && a.dbg_c2v)  // (the barrier inside __syncthreads_or ordered the sweep)
                for (int w = l; w < t.a_words; w += L) reinterpret_cast<T *>(a.dbg_c2v)[(size_t) frame * t.a_words + w] = A[w];
            if (DBG) __syncthreads();
#ifdef ACG_BLOCK_STAMPS
            const unsigned long long t3 = __builtin_readcyclecounter();
#endif
            if (MERGED) {
                if (IDXREG) {
#pragma unroll
                    for (int i = 0; i < 2 * NVP; ++i) asm volatile("" : "+v"(ir[i]));  // see BpCore::var_phase_regs
                }
                hard = IDXREG ? core.template var_phase_regs<REG>(lr, ir, true) : core.var_phase_hard(lr, true);
            } else {
                core.check_phase(true);
                __syncthreads();
                core.var_phase(lr, true);
            }
#ifdef ACG_BLOCK_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t4 = __builtin_readcyclecounter();
            __syncthreads();
            st_var += t4 - t3;
            st_b2 += __builtin_readcyclecounter() - t4;
            st_n += 1;
#else
            __syncthreads();
#endif
            if (DBG && it == a.max_iter - 1 && a.dbg_v2c) {
                for (int w = l; w < t.a_words; w += L) reinterpret_cast<T *>(a.dbg_v2c)[(size_t) frame * t.a_words + w] = A[w];
                for (int p = 0; p < t.n_vpass; ++p)
                    reinterpret_cast<T *>(a.dbg_post)[(size_t) frame * (t.n_vpass * L) + p * L + l] = core.get_llr(lr, p, p * L + l);
                __syncthreads();
            }
            it += 1;
        }
    }
#ifdef ACG_BLOCK_STAMPS
        // developer build: cycles of the LAST frame of this workgroup, per wavefront: {check sweep, barrier, variable sweep, barrier, sweeps}
        if (a.dbg_post && blockIdx.x == 0 && (l & 63) == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dbg_post) + (size_t) (l >> 6) * 5;
            o[0] = st_chk; o[1] = st_b1; o[2] = st_var; o[3] = st_b2; o[4] = st_n;
        }
#endif
    if (MC && l == 0 && acc_total) {
        atomicAdd(&a.counters[MC_CORRECT], acc_correct);
        atomicAdd(&a.counters[MC_PSEUDO], acc_pseudo);
        atomicAdd(&a.counters[MC_TOTAL], acc_total);
        atomicAdd(&a.counters[MC_HAM], acc_ham);
        atomicAdd(&a.counters[MC_HAM_OK], acc_ham_ok);
        atomicAdd(&a.counters[MC_HAM_WRONG], acc_ham_wrong);
        atomicAdd(&a.counters[MC_ITERS], acc_iters);
    }
}

template <typename T, int L, int ALGO>
static const void *blk_ptr(bool mc, bool idxlds) {
    if (mc) return idxlds ? (const void *) bp_block_kernel<T, L, ALGO, true, true> : (const void *) bp_block_kernel<T, L, ALGO, true, false>;
    return idxlds ? (const void *) bp_block_kernel<T, L, ALGO, false, true> : (const void *) bp_block_kernel<T, L, ALGO, false, false>;
}

template <typename T, int ALGO>
static const void *blk_ptr_l(int L, bool mc, bool idxlds) {
    if (L == 256) return blk_ptr<T, 256, ALGO>(mc, idxlds);
    if (L == 1024) return blk_ptr<T, 1024, ALGO>(mc, idxlds);
    return nullptr;
}

template <typename T, int ALGO>
static const void *blk_ptr_idxreg(int L, bool regular) {
    if (regular) {  // one check degree, one variable degree: the instance without the other paths (REG)
        if (L == 256) return (const void *) bp_block_kernel<T, 256, ALGO, false, false, true, false, true>;
        if (L == 1024) return (const void *) bp_block_kernel<T, 1024, ALGO, false, false, true, false, true>;
        return nullptr;
    }
    if (L == 256) return (const void *) bp_block_kernel<T, 256, ALGO, false, false, true>;
    if (L == 1024) return (const void *) bp_block_kernel<T, 1024, ALGO, false, false, true>;
    return nullptr;
}

// tests only (acg_ldpc_debug_bp_trace): 256 threads per frame, index table in LDS, message dump compiled in
const void *bp_block_kernel_ptr_dbg(int f64) {
#ifdef ACG_FAST_BUILD
    if (f64) return nullptr;
#else
    if (f64) return (const void *) bp_block_kernel<double, 256, 0, false, true, false, true>;
#endif
    return (const void *) bp_block_kernel<float, 256, 0, false, true, false, true>;
}

// workgroup-per-frame kernels exist for node degree <= 8, L in {256, 1024}; idxreg (decode only, no LDS index copy):
// the index table lives in registers
const void *bp_block_kernel_ptr(int algo, int f64, int L, bool mc, bool idxlds, bool idxreg, bool regular) {
    if (idxreg && !mc && !idxlds) {
#ifdef ACG_FAST_BUILD
        if (f64 || algo) return nullptr;
        return blk_ptr_idxreg<float, 0>(L, regular);
#else
        if (algo == 0) return f64 ? blk_ptr_idxreg<double, 0>(L, regular) : blk_ptr_idxreg<float, 0>(L, regular);
        return f64 ? blk_ptr_idxreg<double, 1>(L, regular) : blk_ptr_idxreg<float, 1>(L, regular);
#endif
    }
#ifdef ACG_FAST_BUILD
    if (f64 || algo) return nullptr;
    return blk_ptr_l<float, 0>(L, mc, idxlds);
#else
    if (algo == 0) return f64 ? blk_ptr_l<double, 0>(L, mc, idxlds) : blk_ptr_l<float, 0>(L, mc, idxlds);
    return f64 ? blk_ptr_l<double, 1>(L, mc, idxlds) : blk_ptr_l<float, 1>(L, mc, idxlds);
#endif
}

}  // namespace acg
