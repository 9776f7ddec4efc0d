// TEST INFRASTRUCTURE — not part of the product path.
//
// Thin extern "C" shim around the *unmodified* reference headers, compiled from
// where they lie under $(ACG_REF) (default /root/reference) by oracle/Makefile
// into oracle/_ref/libacg_ref.so.  No reference source is copied into this repo;
// this file only calls the reference's public (and, for the soft-message trace,
// private) entry points.  Used for:
//   * generating tests/golden/*.npz (oracle/make_golden.py),
//   * validating oracle/ldpc_oracle.c (tests/test_oracle_vs_ref.py, skipped
//     when the .so is absent),
//   * bench.py's cpu_baseline leg (kind "reference").
//
// Reference entry points used (file:line in the reference tree):
//   read_pcm                     utils/parse_data.h:6
//   llr_variance / llr           utils/channel.h:12,14
//   transmit                     utils/channel.h:19
//   gen_random_codewords         utils/channel.h:39
//   GetOrtogonal / IsCodeword    utils/codeword.h:97,90
//   BeliefPropagationDecoder     algo/bp.h:208   (decode: bp.h:183)
//   ConstructADMMProblem         algo/qp_admm.h:13
//   DecodeQPADMM                 algo/qp_admm.h:104
//   multithread_experiment       experiment.h:125
//
// The reference BP keeps a global, non-atomic node counter (bp.h:13,32): it is
// only ever driven single-threaded here.

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <pthread.h>

#include "experiment.h"
#include "utils/parse_data.h"
#include "utils/codeword.h"
#include "algo/algo.h"
// soft-message trace needs BeliefPropagation::_graph and TannerGraph::_v_nodes
#define private public
#define protected public
#include "algo/bp.h"
#undef private
#undef protected
#include "algo/qp_admm.h"

namespace {

TMatrix to_tmatrix(const uint8_t *H, int m, int n) {
    TMatrix r(m);
    for (int i = 0; i < m; i++) {
        r[i].resize(n);
        for (int j = 0; j < n; j++) r[i][j] = H[(size_t) i * n + j] != 0;
    }
    return r;
}

TFVector to_fvec(const double *y, int n) { return TFVector(y, y + n); }

}  // namespace

extern "C" {

// returns 0 on success; m,n filled; out (cap bytes) receives row-major 0/1.
int acgref_read_pcm(const char *path, uint8_t *out, long cap, int *m, int *n) {
    TMatrix H = read_pcm(path);
    if (H.empty()) return 1;
    *m = (int) H.size();
    *n = (int) H[0].size();
    if ((long) *m * *n > cap) return 2;
    for (int i = 0; i < *m; i++) {
        if ((int) H[i].size() != *n) return 3;
        for (int j = 0; j < *n; j++) out[(size_t) i * *n + j] = H[i][j];
    }
    return 0;
}

void acgref_save_matrix(const uint8_t *H, int m, int n, const char *path) {
    save_matrix(to_tmatrix(H, m, n), path);
}

double acgref_llr_variance(double snr) { return llr_variance(snr); }

double acgref_llr(double v, double snr) { return llr(v, snr); }

void acgref_transmit(uint32_t seed, double snr, const uint8_t *cw, int n, double *y) {
    TCodeword c(n);
    for (int i = 0; i < n; i++) c[i] = cw[i] != 0;
    mt19937 rnd(seed);
    TFVector t = transmit(snr, c, rnd);
    for (int i = 0; i < n; i++) y[i] = t[i];
}

// G must hold (n-m)*n bytes. returns 1 if ok, 0 if H is rank deficient in the
// reference's sense (GetOrtogonal second == false).
int acgref_get_orthogonal(const uint8_t *H, int m, int n, uint8_t *G) {
    auto o = GetOrtogonal(to_tmatrix(H, m, n));
    if (!o.second) return 0;
    for (int i = 0; i < n - m; i++)
        for (int j = 0; j < n; j++) G[(size_t) i * n + j] = o.first[i][j];
    return 1;
}

void acgref_gen_codewords(const uint8_t *G, int k, int n, uint32_t seed, int count, uint8_t *out) {
    TMatrix g = to_tmatrix(G, k, n);
    mt19937 rnd(seed);
    vector<TCodeword> cw = gen_random_codewords(g, count, rnd);
    for (int f = 0; f < count; f++)
        for (int j = 0; j < n; j++) out[(size_t) f * n + j] = cw[f][j];
}

int acgref_is_codeword(const uint8_t *H, int m, int n, const uint8_t *c) {
    TCodeword cw(n);
    for (int i = 0; i < n; i++) cw[i] = c[i] != 0;
    return IsCodeword(to_tmatrix(H, m, n), cw) ? 1 : 0;
}

void acgref_reset_node_counter() { Node::counter = 0; }

// One frame through BeliefPropagationDecoder(max_iter).  bits: n bytes, zeroed
// on failure (the reference returns an EMPTY vector then, bp.h:198).
// returns the reference's bool flag.
int acgref_bp_decode(const uint8_t *H, int m, int n, const double *y, double snr, int max_iter,
                     uint8_t *bits) {
    BeliefPropagationDecoder dec(max_iter);
    auto p = dec.decode(to_tmatrix(H, m, n), to_fvec(y, n), snr);
    memset(bits, 0, n);
    if (p.second) {
        assert((int) p.first.size() == n);
        for (int i = 0; i < n; i++) bits[i] = p.first[i];
    } else {
        assert(p.first.empty());
    }
    return p.second ? 1 : 0;
}

// Batch version: frames x n inputs; ok[f] flag; returns seconds spent inside decode().
double acgref_bp_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                              int max_iter, uint8_t *bits, uint8_t *ok) {
    TMatrix h = to_tmatrix(H, m, n);
    BeliefPropagationDecoder dec(max_iter);
    double t = 0;
    for (int f = 0; f < frames; f++) {
        TFVector yy = to_fvec(y + (size_t) f * n, n);
        auto t0 = chrono::steady_clock::now();
        auto p = dec.decode(h, yy, snr);
        t += chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        uint8_t *b = bits + (size_t) f * n;
        memset(b, 0, n);
        if (p.second)
            for (int i = 0; i < n; i++) b[i] = p.first[i];
        ok[f] = p.second;
    }
    return t;
}

// Soft state after `iters` full iterations of the reference schedule
// (bp.h:183-199 without the exit test).  Edge order: check-major, variables
// ascending inside a check.  c2v[e]: signed message check->variable as stored
// in the variable's mailbox (0 before the first iteration); v2c_mag/v2c_sgn:
// the (phi(|x|), sign) pair stored in the check's mailbox; post[v] = estimate().
// returns number of edges.
int acgref_bp_trace(const uint8_t *H, int m, int n, const double *y, double snr, int iters, double *c2v,
                    double *v2c_mag, double *v2c_sgn, double *post) {
    TMatrix h = to_tmatrix(H, m, n);
    TannerGraph graph = from_biadjacency_matrix(h, snr, to_fvec(y, n));
    BeliefPropagation bp(graph, h, iters);
    bp.c_receive_messages();
    for (int it = 0; it < iters; it++) {
        bp.v_receive_messages();
        bp.c_receive_messages();
    }
    int e = 0;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++)
            if (h[i][j]) {
                CNode &c = bp._graph.c_node(i);
                VNode &v = bp._graph.v_node(j);
                Message mc = v._received_messages[c.uuid()];
                Message mv = c._received_messages[v.uuid()];
                c2v[e] = (double) mc.first;
                v2c_mag[e] = (double) mv.first;
                v2c_sgn[e] = (double) mv.second;
                e++;
            }
    for (int j = 0; j < n; j++) post[j] = (double) bp._graph.v_node(j).estimate();
    return e;
}

int acgref_qpadmm_decode(const uint8_t *H, int m, int n, const double *y, double snr, double alpha,
                         double mu, int max_iter, double eps, uint8_t *bits) {
    auto p = DecodeQPADMM(to_tmatrix(H, m, n), to_fvec(y, n), snr, alpha, mu, max_iter, eps);
    for (int i = 0; i < n; i++) bits[i] = p.first[i];
    return p.second ? 1 : 0;
}

double acgref_qpadmm_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                                  double alpha, double mu, int max_iter, double eps, uint8_t *bits,
                                  uint8_t *ok) {
    TMatrix h = to_tmatrix(H, m, n);
    QPADMMDecoder dec(alpha, mu, max_iter, eps);
    double t = 0;
    for (int f = 0; f < frames; f++) {
        TFVector yy = to_fvec(y + (size_t) f * n, n);
        auto t0 = chrono::steady_clock::now();
        auto p = dec.decode(h, yy, snr);
        t += chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < n; i++) bits[(size_t) f * n + i] = p.first[i];
        ok[f] = p.second;
    }
    return t;
}

// Structural pins of ConstructADMMProblem: out = {n_var, n_con, nnz, e_min, e_max}
void acgref_admm_shape(const uint8_t *H, int m, int n, double *out) {
    vector<double> y(n, 1.0);
    ADMMProblem p = ConstructADMMProblem(to_tmatrix(H, m, n), y, 0.0);
    long nnz = 0;
    for (auto &col: p.A) nnz += (long) col.size();
    out[0] = (double) p.q.size();
    out[1] = (double) p.b.size();
    out[2] = (double) nnz;
    out[3] = *min_element(p.e.begin(), p.e.end());
    out[4] = *max_element(p.e.begin(), p.e.end());
}

// Full A in column-list form for structural comparison: for variable i, entries
// (con, coef) in construction order. col_ptr has n_var+1 entries.
void acgref_admm_matrix(const uint8_t *H, int m, int n, int *col_ptr, int *con, double *coef, double *b) {
    vector<double> y(n, 1.0);
    ADMMProblem p = ConstructADMMProblem(to_tmatrix(H, m, n), y, 0.0);
    int k = 0;
    for (int i = 0; i < (int) p.A.size(); i++) {
        col_ptr[i] = k;
        for (auto &f: p.A[i]) {
            con[k] = f.first;
            coef[k] = f.second;
            k++;
        }
    }
    col_ptr[p.A.size()] = k;
    for (int j = 0; j < (int) p.b.size(); j++) b[j] = p.b[j];
}

// experiment.h Monte-Carlo loop, driven with ONE thread (frame i seeded with
// mt19937(i+1), experiment.h:90-97).  kind: 0 = BP(max_iter), 1 = QP-ADMM.
// out = {correct, pseudo, total, sum_hamming, sum_hamming_ok, sum_hamming_wrong}; returns time_sec.
double acgref_experiment(int kind, int max_iter, double alpha, double mu, double eps, const uint8_t *H, int m,
                         int n, const uint8_t *codewords, int count, double snr, long *out) {
    TMatrix h = to_tmatrix(H, m, n);
    vector<TCodeword> cws(count);
    for (int f = 0; f < count; f++) {
        cws[f].resize(n);
        for (int j = 0; j < n; j++) cws[f][j] = codewords[(size_t) f * n + j] != 0;
    }
    shared_ptr<Decoder> dec;
    if (kind == 0)
        dec = make_shared<BeliefPropagationDecoder>(max_iter);
    else
        dec = make_shared<QPADMMDecoder>(alpha, mu, max_iter, eps);
    ExperimentResult r = multithread_experiment(dec, cws, h, snr, 1);
    out[0] = r.correct;
    out[1] = r.pseudo;
    out[2] = r.total;
    out[3] = r.tr.sum_hamming;
    out[4] = r.tr.sum_hamming_ok;
    out[5] = r.tr.sum_hamming_wrong;
    return r.time_sec;
}

}  // extern "C"
