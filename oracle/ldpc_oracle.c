/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference hot path.
 * See ldpc_oracle.h for the usage rules and the parity status.
 *
 * Each function cites the reference file:line it follows.  The arithmetic types are the
 * reference's: long double (x87 80-bit) for BP messages (bp.h:9), double for LLRs and ADMM.
 *
 * Known, documented deviation: the reference sums BP mailboxes in
 * std::unordered_map iteration order (bp.h:51,79,87), which depends on ever-growing node
 * uuids; here neighbours are summed in ascending index order.  The difference is a few ulp
 * of an 80-bit long double and never reaches a hard decision on any fixture.
 */
#define _GNU_SOURCE
#include "ldpc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* ------------------------------------------------------------------ text format */

/* utils/parse_data.h:6-25.  Rows are whitespace separated tokens; a token gets a trailing ','
 * if it has none (:15-16); every ',' emits the value decided by the last non-',' character,
 * and only '1' is true (:17-22).  The flag `t` is declared outside the row loop (:10) so it
 * carries over between rows; it starts uninitialised in the reference, false here. */
int ldo_read_pcm(const char *path, uint8_t *out, long cap, int *m, int *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    long rows = 0, cols = -1, cur = 0, used = 0;
    int t = 0, in_tok = 0, last = 0, c;
    for (;;) {
        c = fgetc(f);
        int ws = (c == EOF || c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f');
        if (!ws) {
            in_tok = 1;
            last = c;
            if (c == ',') {
                if (used >= cap) { fclose(f); return 2; }
                out[used++] = (uint8_t) t;
                cur++;
            } else {
                t = (c == '1');
            }
        } else {
            if (in_tok) {
                if (last != ',') { /* implicit trailing comma */
                    if (used >= cap) { fclose(f); return 2; }
                    out[used++] = (uint8_t) t;
                    cur++;
                }
                if (cols < 0) cols = cur;
                else if (cols != cur) { fclose(f); return 3; }
                rows++;
                cur = 0;
                in_tok = 0;
            }
            if (c == EOF) break;
        }
    }
    fclose(f);
    if (rows == 0) return 1;
    *m = (int) rows;
    *n = (int) cols;
    return 0;
}

/* utils/parse_data.h:44-54 */
int ldo_save_matrix(const uint8_t *H, int m, int n, const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < n; j++) {
            fputc(H[(size_t) i * n + j] ? '1' : '0', f);
            if (j != n - 1) fputc(',', f);
        }
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ channel */

/* utils/channel.h:12 */
double ldo_llr_variance(double snr) { return pow(10, -(snr / 10)) / 2; }

/* utils/channel.h:14-16 */
double ldo_llr(double v, double snr) { return 2 * v / ldo_llr_variance(snr); }

/* std::mt19937 (ISO C++ [rand.eng.mers], 32-bit MT19937) */
typedef struct {
    uint32_t s[624];
    int idx;
} mt19937_t;

static void mt_seed(mt19937_t *g, uint32_t seed) {
    g->s[0] = seed;
    for (int i = 1; i < 624; i++) g->s[i] = 1812433253u * (g->s[i - 1] ^ (g->s[i - 1] >> 30)) + (uint32_t) i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->s[i] & 0x80000000u) | (g->s[(i + 1) % 624] & 0x7fffffffu);
            g->s[i] = g->s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->s[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* libstdc++ std::generate_canonical<double, 53>(mt19937): two 32-bit draws, accumulated in
 * double, divided by 2^64, clamped below 1 (bits/random.tcc). */
static double canonical53(mt19937_t *g) {
    double sum = 0.0, tmp = 1.0;
    const long double r = 4294967296.0L;
    for (int k = 0; k < 2; k++) {
        sum += (double) mt_next(g) * tmp;
        tmp = (double) ((long double) tmp * r);
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* utils/channel.h:18-26: y_i = (bit ? -1 : +1) + N(0, sigma^2), libstdc++
 * normal_distribution<double> = Marsaglia polar method, second value cached. */
static void transmit_gen(mt19937_t *g, double snr, const uint8_t *cw, int n, double *y) {
    double sigma = sqrt(ldo_llr_variance(snr));
    int have = 0;
    double saved = 0.0;
    for (int i = 0; i < n; i++) {
        double ret;
        if (have) {
            have = 0;
            ret = saved;
        } else {
            double x, yy, r2;
            do {
                x = 2.0 * canonical53(g) - 1.0;
                yy = 2.0 * canonical53(g) - 1.0;
                r2 = x * x + yy * yy;
            } while (r2 > 1.0 || r2 == 0.0);
            double mult = sqrt(-2 * log(r2) / r2);
            saved = x * mult;
            have = 1;
            ret = yy * mult;
        }
        ret = ret * sigma + 0.0;
        y[i] = (cw[i] ? -1.0 : 1.0) + ret;
    }
}

void ldo_transmit(uint32_t seed, double snr, const uint8_t *cw, int n, double *y) {
    mt19937_t g;
    mt_seed(&g, seed);
    transmit_gen(&g, snr, cw, n, y);
}

/* ------------------------------------------------------------------ GF(2) support */

/* utils/codeword.h:97-128: Gauss-Jordan, pivot of row i = its first non-zero column */
int ldo_get_orthogonal(const uint8_t *Hin, int m, int n, uint8_t *G) {
    uint8_t *H = (uint8_t *) malloc((size_t) m * n);
    int *pos = (int *) malloc(sizeof(int) * (size_t) m);
    uint8_t *is_main = (uint8_t *) calloc((size_t) n, 1);
    memcpy(H, Hin, (size_t) m * n);
    int ok = 1;
    for (int i = 0; i < m && ok; i++) {
        pos[i] = -1;
        for (int j = 0; j < n; j++)
            if (H[(size_t) i * n + j]) {
                pos[i] = j;
                break;
            }
        if (pos[i] == -1) {
            ok = 0;
            break;
        }
        for (int k = 0; k < m; k++)
            if (k != i && H[(size_t) k * n + pos[i]])
                for (int j = 0; j < n; j++) H[(size_t) k * n + j] ^= H[(size_t) i * n + j];
        is_main[pos[i]] = 1;
    }
    if (ok) {
        memset(G, 0, (size_t) (n - m) * n);
        int idx = 0;
        for (int j = 0; j < n; j++)
            if (!is_main[j]) {
                G[(size_t) idx * n + j] = 1;
                for (int i = 0; i < m; i++)
                    if (H[(size_t) i * n + j]) G[(size_t) idx * n + pos[i]] = 1;
                idx++;
            }
    }
    free(H);
    free(pos);
    free(is_main);
    return ok;
}

/* utils/channel.h:28-44: row i of G is XORed in when rnd() % 2 == 0 */
void ldo_gen_codewords(const uint8_t *G, int k, int n, uint32_t seed, int count, uint8_t *out) {
    mt19937_t g;
    mt_seed(&g, seed);
    for (int f = 0; f < count; f++) {
        uint8_t *res = out + (size_t) f * n;
        memset(res, 0, (size_t) n);
        for (int i = 0; i < k; i++)
            if (mt_next(&g) % 2 == 0)
                for (int j = 0; j < n; j++) res[j] ^= G[(size_t) i * n + j];
    }
}

/* utils/codeword.h:90-95 (dense GF(2) product there; same value) */
int ldo_is_codeword(const uint8_t *H, int m, int n, const uint8_t *c) {
    for (int i = 0; i < m; i++) {
        int s = 0;
        for (int j = 0; j < n; j++) s ^= (H[(size_t) i * n + j] & c[j]);
        if (s) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ Tanner graph (flat) */

typedef struct {
    int m, n, E;
    int *row_ptr; /* m+1, edges in check-major order, variables ascending (bp.h:144-147) */
    int *edge_var;
    int *col_ptr; /* n+1 */
    int *col_edge; /* edge ids per variable, checks ascending */
} graph_t;

static void graph_build(graph_t *g, const uint8_t *H, int m, int n) {
    g->m = m;
    g->n = n;
    int E = 0;
    for (size_t i = 0; i < (size_t) m * n; i++) E += H[i] != 0;
    g->E = E;
    g->row_ptr = (int *) malloc(sizeof(int) * (size_t) (m + 1));
    g->edge_var = (int *) malloc(sizeof(int) * (size_t) (E > 0 ? E : 1));
    g->col_ptr = (int *) calloc((size_t) (n + 1), sizeof(int));
    g->col_edge = (int *) malloc(sizeof(int) * (size_t) (E > 0 ? E : 1));
    int e = 0;
    for (int i = 0; i < m; i++) {
        g->row_ptr[i] = e;
        for (int j = 0; j < n; j++)
            if (H[(size_t) i * n + j]) {
                g->edge_var[e++] = j;
                g->col_ptr[j + 1]++;
            }
    }
    g->row_ptr[m] = e;
    for (int j = 0; j < n; j++) g->col_ptr[j + 1] += g->col_ptr[j];
    int *fill = (int *) calloc((size_t) n, sizeof(int));
    for (int ed = 0; ed < E; ed++) {
        int v = g->edge_var[ed];
        g->col_edge[g->col_ptr[v] + fill[v]++] = ed;
    }
    free(fill);
}

static void graph_free(graph_t *g) {
    free(g->row_ptr);
    free(g->edge_var);
    free(g->col_ptr);
    free(g->col_edge);
}

/* ------------------------------------------------------------------ sum-product BP */

/* bp.h:34 */
static long double phi(long double x) { return -logl(tanhl(x / 2)); }

typedef struct {
    long double *c2v;     /* VNode mailbox value (bp.h:57 returns {sgn*phi(sum), 1}) */
    long double *v2c_mag; /* CNode mailbox .first  (bp.h:82) */
    long double *v2c_sgn; /* CNode mailbox .second (bp.h:82) */
    long double *llr;     /* VNode::_channel_llr (bp.h:66) */
} bp_state_t;

static void bp_alloc(bp_state_t *s, const graph_t *g) {
    size_t E = (size_t) (g->E > 0 ? g->E : 1);
    s->c2v = (long double *) malloc(sizeof(long double) * E);
    s->v2c_mag = (long double *) malloc(sizeof(long double) * E);
    s->v2c_sgn = (long double *) malloc(sizeof(long double) * E);
    s->llr = (long double *) malloc(sizeof(long double) * (size_t) g->n);
}

static void bp_free(bp_state_t *s) {
    free(s->c2v);
    free(s->v2c_mag);
    free(s->v2c_sgn);
    free(s->llr);
}

/* bp.h:136-153: llr in double (channel.h:14), mailboxes start at (0, 1) (bp.h:40-43,70-73) */
static void bp_init(bp_state_t *s, const graph_t *g, const double *y, double snr) {
    for (int j = 0; j < g->n; j++) s->llr[j] = (long double) ldo_llr(y[j], snr);
    for (int e = 0; e < g->E; e++) {
        s->c2v[e] = 0.0L;
        s->v2c_mag[e] = 0.0L;
        s->v2c_sgn[e] = 1.0L;
    }
}

/* bp.h:160-169 with VNode::message bp.h:77-83 */
static void bp_v2c(bp_state_t *s, const graph_t *g) {
    for (int v = 0; v < g->n; v++) {
        int b = g->col_ptr[v], e_ = g->col_ptr[v + 1];
        for (int a = b; a < e_; a++) {
            long double sum = 0;
            for (int o = b; o < e_; o++)
                if (o != a) sum += s->c2v[g->col_edge[o]];
            long double x = s->llr[v] + sum;
            int ed = g->col_edge[a];
            s->v2c_mag[ed] = phi(fabsl(x));
            s->v2c_sgn[ed] = (x <= 0) ? -1.0L : 1.0L;
        }
    }
}

/* bp.h:171-181 with CNode::message bp.h:49-57.  All outputs of a half-iteration are computed
 * from the mailboxes of the other side only, so the in-place write below is the flooding
 * schedule of the reference. */
static void bp_c2v(bp_state_t *s, const graph_t *g) {
    for (int c = 0; c < g->m; c++) {
        int b = g->row_ptr[c], e_ = g->row_ptr[c + 1];
        for (int a = b; a < e_; a++) {
            long double sum = 0, sgn = 1;
            for (int o = b; o < e_; o++)
                if (o != a) {
                    sum += s->v2c_mag[o];
                    sgn *= s->v2c_sgn[o];
                }
            s->c2v[a] = sgn * phi(sum);
        }
    }
}

/* bp.h:85-90 */
static long double bp_estimate(const bp_state_t *s, const graph_t *g, int v) {
    long double sum = 0;
    for (int o = g->col_ptr[v]; o < g->col_ptr[v + 1]; o++) sum += s->c2v[g->col_edge[o]];
    return s->llr[v] + sum;
}

static int syndrome_zero(const graph_t *g, const uint8_t *bits) {
    for (int c = 0; c < g->m; c++) {
        int sy = 0;
        for (int e = g->row_ptr[c]; e < g->row_ptr[c + 1]; e++) sy ^= bits[g->edge_var[e]];
        if (sy) return 0;
    }
    return 1;
}

/* bp.h:183-199 */
static int bp_decode_graph(const graph_t *g, bp_state_t *s, const double *y, double snr, int max_iter,
                           uint8_t *bits, int *iters) {
    uint8_t *est = (uint8_t *) malloc((size_t) g->n);
    bp_init(s, g, y, snr);
    bp_v2c(s, g);
    int ok = 0, it;
    for (it = 0; it < max_iter; it++) {
        bp_c2v(s, g);
        bp_v2c(s, g);
        for (int v = 0; v < g->n; v++) est[v] = bp_estimate(s, g, v) <= 0;
        if (syndrome_zero(g, est)) {
            ok = 1;
            it++;
            break;
        }
    }
    if (ok) memcpy(bits, est, (size_t) g->n);
    else memset(bits, 0, (size_t) g->n); /* reference returns an empty vector (bp.h:198) */
    if (iters) *iters = it;
    free(est);
    return ok;
}

int ldo_bp_decode(const uint8_t *H, int m, int n, const double *y, double snr, int max_iter, uint8_t *bits,
                  int *iters) {
    graph_t g;
    bp_state_t s;
    graph_build(&g, H, m, n);
    bp_alloc(&s, &g);
    int ok = bp_decode_graph(&g, &s, y, snr, max_iter, bits, iters);
    bp_free(&s);
    graph_free(&g);
    return ok;
}

double ldo_bp_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                           int max_iter, int threads, uint8_t *bits, uint8_t *ok, int32_t *iters) {
    graph_t g;
    graph_build(&g, H, m, n);
    if (threads < 1) threads = 1;
    double t0 = now_sec();
#pragma omp parallel num_threads(threads)
    {
        bp_state_t s;
        bp_alloc(&s, &g);
#pragma omp for schedule(dynamic, 16)
        for (int f = 0; f < frames; f++) {
            int it = 0;
            ok[f] = (uint8_t) bp_decode_graph(&g, &s, y + (size_t) f * n, snr, max_iter, bits + (size_t) f * n, &it);
            if (iters) iters[f] = it;
        }
        bp_free(&s);
    }
    double t = now_sec() - t0;
    graph_free(&g);
    return t;
}

int ldo_bp_trace(const uint8_t *H, int m, int n, const double *y, double snr, int iters, double *c2v,
                 double *v2c_mag, double *v2c_sgn, double *post) {
    graph_t g;
    bp_state_t s;
    graph_build(&g, H, m, n);
    bp_alloc(&s, &g);
    bp_init(&s, &g, y, snr);
    bp_v2c(&s, &g);
    for (int it = 0; it < iters; it++) {
        bp_c2v(&s, &g);
        bp_v2c(&s, &g);
    }
    for (int e = 0; e < g.E; e++) {
        c2v[e] = (double) s.c2v[e];
        v2c_mag[e] = (double) s.v2c_mag[e];
        v2c_sgn[e] = (double) s.v2c_sgn[e];
    }
    for (int v = 0; v < n; v++) post[v] = (double) bp_estimate(&s, &g, v);
    int E = g.E;
    bp_free(&s);
    graph_free(&g);
    return E;
}

/* ------------------------------------------------------------------ min-sum (build-added, unpinned) */

static int minsum_decode_graph(const graph_t *g, const double *y, double snr, int max_iter, double scale,
                               uint8_t *bits, int *iters, double *c2v, double *v2c, double *llr) {
    uint8_t *est = (uint8_t *) malloc((size_t) g->n);
    for (int j = 0; j < g->n; j++) llr[j] = ldo_llr(y[j], snr);
    for (int e = 0; e < g->E; e++) c2v[e] = 0.0;
    int ok = 0, it;
    for (it = -1; it < max_iter; it++) {
        if (it >= 0) {
            for (int c = 0; c < g->m; c++) {
                int b = g->row_ptr[c], e_ = g->row_ptr[c + 1];
                for (int a = b; a < e_; a++) {
                    double mn = INFINITY;
                    int neg = 0;
                    for (int o = b; o < e_; o++)
                        if (o != a) {
                            double av = fabs(v2c[o]);
                            if (av < mn) mn = av;
                            neg ^= (v2c[o] <= 0); /* same sign rule as bp.h:82: x <= 0 -> -1 */
                        }
                    double mag = scale * mn;
                    c2v[a] = neg ? -mag : mag;
                }
            }
        }
        for (int v = 0; v < g->n; v++) {
            int b = g->col_ptr[v], e_ = g->col_ptr[v + 1];
            for (int a = b; a < e_; a++) {
                double sum = 0;
                for (int o = b; o < e_; o++)
                    if (o != a) sum += c2v[g->col_edge[o]];
                v2c[g->col_edge[a]] = llr[v] + sum;
            }
        }
        if (it < 0) continue;
        for (int v = 0; v < g->n; v++) {
            double sum = 0;
            for (int o = g->col_ptr[v]; o < g->col_ptr[v + 1]; o++) sum += c2v[g->col_edge[o]];
            est[v] = (llr[v] + sum) <= 0;
        }
        if (syndrome_zero(g, est)) {
            ok = 1;
            it++;
            break;
        }
    }
    if (ok) memcpy(bits, est, (size_t) g->n);
    else memset(bits, 0, (size_t) g->n);
    if (iters) *iters = it;
    free(est);
    return ok;
}

int ldo_minsum_decode(const uint8_t *H, int m, int n, const double *y, double snr, int max_iter, double scale,
                      uint8_t *bits, int *iters) {
    graph_t g;
    graph_build(&g, H, m, n);
    size_t E = (size_t) (g.E > 0 ? g.E : 1);
    double *c2v = (double *) malloc(sizeof(double) * E), *v2c = (double *) malloc(sizeof(double) * E);
    double *llr = (double *) malloc(sizeof(double) * (size_t) n);
    int ok = minsum_decode_graph(&g, y, snr, max_iter, scale, bits, iters, c2v, v2c, llr);
    free(c2v);
    free(v2c);
    free(llr);
    graph_free(&g);
    return ok;
}

double ldo_minsum_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                               int max_iter, double scale, int threads, uint8_t *bits, uint8_t *ok,
                               int32_t *iters) {
    graph_t g;
    graph_build(&g, H, m, n);
    if (threads < 1) threads = 1;
    double t0 = now_sec();
#pragma omp parallel num_threads(threads)
    {
        size_t E = (size_t) (g.E > 0 ? g.E : 1);
        double *c2v = (double *) malloc(sizeof(double) * E), *v2c = (double *) malloc(sizeof(double) * E);
        double *llr = (double *) malloc(sizeof(double) * (size_t) n);
#pragma omp for schedule(dynamic, 16)
        for (int f = 0; f < frames; f++) {
            int it = 0;
            ok[f] = (uint8_t) minsum_decode_graph(&g, y + (size_t) f * n, snr, max_iter, scale,
                                                  bits + (size_t) f * n, &it, c2v, v2c, llr);
            if (iters) iters[f] = it;
        }
        free(c2v);
        free(v2c);
        free(llr);
    }
    double t = now_sec() - t0;
    graph_free(&g);
    return t;
}

/* ------------------------------------------------------------------ QP-ADMM */

typedef struct {
    int n, n_var, n_con, nnz;
    int *col_ptr; /* n_var+1 */
    int *con;     /* nnz: constraint index, in construction order per variable (qp_admm.h:38-56) */
    double *coef; /* nnz: +-1 */
    double *b;    /* n_con */
    double *e;    /* n_var */
} admm_t;

typedef struct {
    int var, con;
    double coef;
} trip_t;

/* qp_admm.h:13-102 (structure only; q is filled per frame) */
static void admm_build(admm_t *p, const uint8_t *H, int m, int n) {
    int n_aux = 0;
    long cap_trip = 0, cap_con = 0;
    for (int i = 0; i < m; i++) {
        int sz = 0;
        for (int t = 0; t < n; t++) sz += H[(size_t) i * n + t] != 0;
        n_aux += (sz - 3 > 0) ? sz - 3 : 0; /* qp_admm.h:20 */
        int three = sz >= 3 ? sz - 2 : 0;
        cap_trip += 12L * three + 4;
        cap_con += 4L * three + 2;
    }
    int n_var = n + n_aux; /* qp_admm.h:22 */
    trip_t *tr = (trip_t *) malloc(sizeof(trip_t) * (size_t) (cap_trip > 0 ? cap_trip : 1));
    double *b = (double *) malloc(sizeof(double) * (size_t) (cap_con > 0 ? cap_con : 1));
    int nt = 0, nb = 0;
    int *idx = (int *) malloc(sizeof(int) * (size_t) n);

#define ADD_THREE(I, J, Hh)                                                                          \
    do { /* qp_admm.h:34-57 */                                                                       \
        int v3[3] = {(I), (J), (Hh)};                                                                \
        b[nb] = 0.0; b[nb + 1] = 0.0; b[nb + 2] = 0.0; b[nb + 3] = 2.0;                              \
        for (int w = 0; w < 3; w++) {                                                                \
            for (int r = 0; r < 3; r++) { tr[nt].var = v3[w]; tr[nt].con = nb + r; tr[nt].coef = (r == w) ? 1.0 : -1.0; nt++; } \
            tr[nt].var = v3[w]; tr[nt].con = nb + 3; tr[nt].coef = 1.0; nt++;                         \
        }                                                                                            \
        nb += 4;                                                                                     \
    } while (0)

    int pos = n;
    for (int i = 0; i < m; i++) {
        int d = 0;
        for (int j = 0; j < n; j++)
            if (H[(size_t) i * n + j]) idx[d++] = j;
        if (d == 0) continue; /* qp_admm.h:67-69 */
        if (d == 1) {         /* qp_admm.h:70-74 */
            tr[nt].var = idx[0]; tr[nt].con = nb; tr[nt].coef = 1.0; nt++;
            b[nb++] = 0.0;
            continue;
        }
        if (d == 2) { /* qp_admm.h:75-83 */
            b[nb] = 0.0; b[nb + 1] = 0.0;
            tr[nt].var = idx[0]; tr[nt].con = nb; tr[nt].coef = 1.0; nt++;
            tr[nt].var = idx[0]; tr[nt].con = nb + 1; tr[nt].coef = -1.0; nt++;
            tr[nt].var = idx[1]; tr[nt].con = nb; tr[nt].coef = -1.0; nt++;
            tr[nt].var = idx[1]; tr[nt].con = nb + 1; tr[nt].coef = 1.0; nt++;
            nb += 2;
            continue;
        }
        int idx_last = idx[0]; /* qp_admm.h:84-91 */
        for (int j = 1; j < d - 2; j++) {
            int idx_mid = idx[j];
            int idx_aux = pos++;
            ADD_THREE(idx_last, idx_mid, idx_aux);
            idx_last = idx_aux;
        }
        ADD_THREE(idx_last, idx[d - 2], idx[d - 1]);
    }
#undef ADD_THREE

    p->n = n;
    p->n_var = n_var;
    p->n_con = nb;
    p->nnz = nt;
    p->col_ptr = (int *) calloc((size_t) (n_var + 1), sizeof(int));
    p->con = (int *) malloc(sizeof(int) * (size_t) (nt > 0 ? nt : 1));
    p->coef = (double *) malloc(sizeof(double) * (size_t) (nt > 0 ? nt : 1));
    p->b = (double *) malloc(sizeof(double) * (size_t) (nb > 0 ? nb : 1));
    p->e = (double *) calloc((size_t) n_var, sizeof(double));
    memcpy(p->b, b, sizeof(double) * (size_t) nb);
    for (int k = 0; k < nt; k++) p->col_ptr[tr[k].var + 1]++;
    for (int i = 0; i < n_var; i++) p->col_ptr[i + 1] += p->col_ptr[i];
    int *fill = (int *) calloc((size_t) n_var, sizeof(int));
    for (int k = 0; k < nt; k++) { /* stable: keeps construction order per variable */
        int v = tr[k].var, at = p->col_ptr[v] + fill[v]++;
        p->con[at] = tr[k].con;
        p->coef[at] = tr[k].coef;
    }
    for (int i = 0; i < n_var; i++) /* qp_admm.h:94-99 */
        for (int k = p->col_ptr[i]; k < p->col_ptr[i + 1]; k++) p->e[i] += p->coef[k] * p->coef[k];
    free(fill);
    free(tr);
    free(b);
    free(idx);
}

static void admm_free(admm_t *p) {
    free(p->col_ptr);
    free(p->con);
    free(p->coef);
    free(p->b);
    free(p->e);
}

void ldo_admm_shape(const uint8_t *H, int m, int n, double *out) {
    admm_t p;
    admm_build(&p, H, m, n);
    double emin = 1e300, emax = -1e300;
    for (int i = 0; i < p.n_var; i++) {
        if (p.e[i] < emin) emin = p.e[i];
        if (p.e[i] > emax) emax = p.e[i];
    }
    out[0] = p.n_var;
    out[1] = p.n_con;
    out[2] = p.nnz;
    out[3] = emin;
    out[4] = emax;
    admm_free(&p);
}

void ldo_admm_matrix(const uint8_t *H, int m, int n, int *col_ptr, int *con, double *coef, double *b) {
    admm_t p;
    admm_build(&p, H, m, n);
    memcpy(col_ptr, p.col_ptr, sizeof(int) * (size_t) (p.n_var + 1));
    memcpy(con, p.con, sizeof(int) * (size_t) p.nnz);
    memcpy(coef, p.coef, sizeof(double) * (size_t) p.nnz);
    memcpy(b, p.b, sizeof(double) * (size_t) p.n_con);
    admm_free(&p);
}

/* qp_admm.h:104-178.  ws: n_var*3 + n_con*3 doubles of scratch. */
static int admm_decode_prob(const admm_t *p, const double *y, double snr, double alpha, double mu,
                            int max_iter, double eps_stop, uint8_t *bits, int *iters, double *ws) {
    int n = p->n, nv = p->n_var, nc = p->n_con;
    double e_min = 1e9; /* qp_admm.h:108-111 */
    for (int i = 0; i < nv; i++)
        if (p->e[i] < e_min) e_min = p->e[i];
    if (e_min * mu <= alpha) { /* qp_admm.h:112-114 */
        memset(bits, 0, (size_t) n);
        if (iters) *iters = 0;
        return 0;
    }
    double *q = ws, *v = q + nv, *inv = v + nv, *z = inv + nv, *yl = z + nc, *r = yl + nc;
    for (int i = 0; i < nv; i++) q[i] = (i < n) ? ldo_llr(y[i], snr) : 0.0; /* qp_admm.h:24-30 */
    for (int i = 0; i < nv; i++) v[i] = q[i] > 0.0 ? 1.0 : 0.0;               /* dead store, :116-119 */
    for (int j = 0; j < nc; j++) z[j] = yl[j] = 0.0;
    for (int i = 0; i < nv; i++) { /* qp_admm.h:123-127 */
        double A = (mu * p->e[i] - alpha) / 2;
        inv[i] = -1.0 / (2 * A);
    }
    int iter;
    for (iter = 0; iter < max_iter; iter++) {
        for (int i = 0; i < nv; i++) { /* qp_admm.h:132-142 */
            double B = q[i] + (alpha / 2);
            for (int k = p->col_ptr[i]; k < p->col_ptr[i + 1]; k++) {
                int j = p->con[k];
                B += p->coef[k] * (yl[j] + mu * (z[j] - p->b[j]));
            }
            double vi = B * inv[i];
            vi = (vi < 0.0) ? 0.0 : vi; /* std::max(v, 0.0), qp_admm.h:140 */
            vi = (1.0 < vi) ? 1.0 : vi; /* std::min(v, 1.0), qp_admm.h:141 */
            v[i] = vi;
        }
        for (int j = 0; j < nc; j++) r[j] = p->b[j]; /* qp_admm.h:144-151 */
        for (int i = 0; i < nv; i++)
            for (int k = p->col_ptr[i]; k < p->col_ptr[i + 1]; k++) r[p->con[k]] -= p->coef[k] * v[i];
        double sum2 = 0; /* qp_admm.h:154-159 */
        for (int j = 0; j < nc; j++) {
            double zn = r[j] - yl[j];
            z[j] = zn > 0.0 ? zn : 0.0;
            double yn = yl[j] - r[j];
            yl[j] = yn > 0.0 ? yn : 0.0;
            sum2 += (z[j] - r[j]) * (z[j] - r[j]);
        }
        if (sum2 < eps_stop) { /* qp_admm.h:161-163 */
            iter++;
            break;
        }
    }
    for (int i = 0; i < n; i++) bits[i] = (v[i] <= 0.5) ? 0 : 1; /* qp_admm.h:166-175 */
    if (iters) *iters = iter;
    return 1; /* qp_admm.h:165,177 */
}

int ldo_qpadmm_decode(const uint8_t *H, int m, int n, const double *y, double snr, double alpha, double mu,
                      int max_iter, double eps, uint8_t *bits, int *iters) {
    admm_t p;
    admm_build(&p, H, m, n);
    double *ws = (double *) malloc(sizeof(double) * (size_t) (3 * p.n_var + 3 * p.n_con + 1));
    int ok = admm_decode_prob(&p, y, snr, alpha, mu, max_iter, eps, bits, iters, ws);
    free(ws);
    admm_free(&p);
    return ok;
}

double ldo_qpadmm_decode_batch(const uint8_t *H, int m, int n, const double *y, int frames, double snr,
                               double alpha, double mu, int max_iter, double eps, int threads,
                               uint8_t *bits, uint8_t *ok, int32_t *iters) {
    admm_t p;
    admm_build(&p, H, m, n);
    if (threads < 1) threads = 1;
    double t0 = now_sec();
#pragma omp parallel num_threads(threads)
    {
        double *ws = (double *) malloc(sizeof(double) * (size_t) (3 * p.n_var + 3 * p.n_con + 1));
#pragma omp for schedule(dynamic, 16)
        for (int f = 0; f < frames; f++) {
            int it = 0;
            ok[f] = (uint8_t) admm_decode_prob(&p, y + (size_t) f * n, snr, alpha, mu, max_iter, eps,
                                               bits + (size_t) f * n, &it, ws);
            if (iters) iters[f] = it;
        }
        free(ws);
    }
    double t = now_sec() - t0;
    admm_free(&p);
    return t;
}

/* ------------------------------------------------------------------ Monte-Carlo harness */

/* experiment.h:80-123 single-threaded; HammingDistanceTracker experiment.h:33-46 */
double ldo_experiment(int kind, int max_iter, double alpha, double mu, double eps, const uint8_t *H, int m,
                      int n, const uint8_t *codewords, int count, double snr, long *out) {
    graph_t g;
    bp_state_t s;
    admm_t p;
    graph_build(&g, H, m, n);
    bp_alloc(&s, &g);
    admm_build(&p, H, m, n);
    double *ws = (double *) malloc(sizeof(double) * (size_t) (3 * p.n_var + 3 * p.n_con + 1 + 2 * g.E + n));
    double *y = (double *) malloc(sizeof(double) * (size_t) n);
    uint8_t *bits = (uint8_t *) malloc((size_t) n);
    long correct = 0, pseudo = 0, total = 0, ham = 0, ham_ok = 0, ham_wrong = 0;
    double tsec = 0;
    for (int f = 0; f < count; f++) {
        const uint8_t *cw = codewords + (size_t) f * n;
        ldo_transmit((uint32_t) (f + 1), snr, cw, n, y); /* experiment.h:97-99 */
        double t0 = now_sec();
        int ok, it;
        if (kind == 0) ok = bp_decode_graph(&g, &s, y, snr, max_iter, bits, &it);
        else if (kind == 1) ok = admm_decode_prob(&p, y, snr, alpha, mu, max_iter, eps, bits, &it, ws);
        else ok = minsum_decode_graph(&g, y, snr, max_iter, alpha, bits, &it, ws, ws + g.E, ws + 2 * g.E);
        tsec += now_sec() - t0;
        int is_correct = 0;
        if (ok && syndrome_zero(&g, bits)) { /* experiment.h:110-117 */
            if (memcmp(bits, cw, (size_t) n) == 0) {
                correct++;
                is_correct = 1;
            } else
                pseudo++;
        }
        total++;
        int h = 0;
        for (int i = 0; i < n; i++) {
            if (!cw[i] && y[i] <= 0) h++;
            if (cw[i] && y[i] > 0) h++;
        }
        ham += h;
        if (is_correct) ham_ok += h;
        else ham_wrong += h;
    }
    out[0] = correct;
    out[1] = pseudo;
    out[2] = total;
    out[3] = ham;
    out[4] = ham_ok;
    out[5] = ham_wrong;
    free(ws);
    free(y);
    free(bits);
    admm_free(&p);
    bp_free(&s);
    graph_free(&g);
    return tsec;
}
