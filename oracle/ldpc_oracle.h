/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (acg_alp_ldpc_amd/) never links, loads or calls it.
 *
 * Parity status: PINNED.  Every function below is checked against the real reference
 * (oracle/_ref/libacg_ref.so, built from /root/reference by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py, and against the committed fixtures in tests/golden/
 * (generated from the real reference by oracle/make_golden.py) in tests/test_oracle_golden.py.
 * The min-sum variant has no reference counterpart: "parity unpinned".
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* utils/parse_data.h:6-25 (read_pcm) and :44-54 (save_matrix) */
int ldo_read_pcm(const char *path, uint8_t *out, long cap, int *m, int *n);
int ldo_save_matrix(const uint8_t *H, int m, int n, const char *path);

/* utils/channel.h:12-16 */
double ldo_llr_variance(double snr);
double ldo_llr(double v, double snr);

/* utils/channel.h:18-26 with std::mt19937(seed) + libstdc++ normal_distribution<double> */
void ldo_transmit(uint32_t seed, double snr, const uint8_t *cw, int n, double *y);

/* utils/codeword.h:97-128 (GetOrtogonal): G gets (n-m) x n; returns 1 ok / 0 failure */
int ldo_get_orthogonal(const uint8_t *H, int m, int n, uint8_t *G);
/* utils/channel.h:28-44 with std::mt19937(seed) */
void ldo_gen_codewords(const uint8_t *G, int k, int n, uint32_t seed, int count, uint8_t *out);
/* utils/codeword.h:90-95 */
int ldo_is_codeword(const uint8_t *H, int m, int n, const uint8_t *c);

/* algo/bp.h:183-199 via :208-222.  bits zeroed on failure; *iters = iteration (1-based) of the
 * first zero syndrome, or max_iter on failure.  returns ok flag. */
int ldo_bp_decode(const uint8_t *H, int m, int n, const double *y, double snr, int max_iter,
                  uint8_t *bits, int *iters);
/* threads: OpenMP threads over frames (the restatement has no global state). returns wall seconds. */
double ldo_bp_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                           int max_iter, int threads, uint8_t *bits, uint8_t *ok, int32_t *iters);
/* soft state after `iters` full iterations (no exit test); same layout as acgref_bp_trace */
int ldo_bp_trace(const uint8_t *H, int m, int n, const double *y, double snr, int iters, double *c2v,
                 double *v2c_mag, double *v2c_sgn, double *post);

/* build-added min-sum variant (NOT in the reference; parity unpinned).  Same flooding schedule and
 * exit rule as bp.h:183-199; check message = prod(sign) * scale * min|.| over the other edges. */
int ldo_minsum_decode(const uint8_t *H, int m, int n, const double *y, double snr, int max_iter,
                      double scale, uint8_t *bits, int *iters);
double ldo_minsum_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                               int max_iter, double scale, int threads, uint8_t *bits, uint8_t *ok,
                               int32_t *iters);

/* algo/qp_admm.h:13-102: out = {n_var, n_con, nnz, e_min, e_max} */
void ldo_admm_shape(const uint8_t *H, int m, int n, double *out);
void ldo_admm_matrix(const uint8_t *H, int m, int n, int *col_ptr, int *con, double *coef, double *b);
/* algo/qp_admm.h:104-178; *iters = number of sweeps executed */
int ldo_qpadmm_decode(const uint8_t *H, int m, int n, const double *y, double snr, double alpha, double mu,
                      int max_iter, double eps, uint8_t *bits, int *iters);
double ldo_qpadmm_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                               double alpha, double mu, int max_iter, double eps, int threads,
                               uint8_t *bits, uint8_t *ok, int32_t *iters);

/* experiment.h:80-123 driven single-threaded: frame i uses mt19937(i+1).
 * kind 0 = BP, 1 = QP-ADMM, 2 = min-sum(scale=alpha).
 * out = {correct, pseudo, total, sum_hamming, sum_hamming_ok, sum_hamming_wrong}. returns decode seconds. */
double ldo_experiment(int kind, int max_iter, double alpha, double mu, double eps, const uint8_t *H, int m,
                      int n, const uint8_t *codewords, int count, double snr, long *out);

#ifdef __cplusplus
}
#endif
#endif
