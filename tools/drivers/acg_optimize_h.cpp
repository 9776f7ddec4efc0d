// N3 — check-matrix local search: the reference's optimize_H.cpp (quasi-cyclic protograph of Z x Z circulants, one
// random block mutated per proposal, accepted when the QP-ADMM FER at -3 dB over 1000 frames drops) on the device.
//   acg_optimize_h [--init data/H05.txt | --random 8,14] [--Z 20] [--iters 10000] [--tests 1000] [--snr -3]
//                  [--alpha 1.95 --mu 0.5 --admm-iters 1000] [--seed 239] [--out optimalH.txt] [--noise host|device]
//                  [--dump-proposals N [--accept-all]]   host only: print "row col present shift" of the first N proposals
//                                                        (every one rejected, or every one accepted) and stop — pinned against
//                                                        the reference's own random_permute in tests/test_drivers.py
// Every proposal is a fresh H: graph analysis, generator (GetOrtogonal) and decoder are rebuilt per proposal through
// the C ABI; a proposal whose generator does not exist scores FER = 1 (optimize_H.cpp:17-19).
#include <iostream>
#include <random>

#include "common.hpp"

struct QcMatrix {  // quasi-cyclic description: block mask + one shift per present block
    int Z = 0, R = 0, C = 0;
    std::vector<uint8_t> present;  // R*C
    std::vector<int> shift;        // R*C

    std::vector<uint8_t> dense() const {  // row i*Z+k has its one at column j*Z + (shift + k) mod Z
        std::vector<uint8_t> H((size_t) R * Z * C * Z, 0);
        for (int i = 0; i < R; i++)
            for (int j = 0; j < C; j++)
                if (present[(size_t) i * C + j])
                    for (int k = 0; k < Z; k++)
                        H[(size_t) (i * Z + k) * (C * Z) + (size_t) j * Z + (shift[(size_t) i * C + j] + k) % Z] = 1;
        return H;
    }

    // inverse of dense(): fails (false) if some block is not a single circulant
    bool from_dense(const std::vector<uint8_t> &H, int m, int n, int z) {
        if (m % z || n % z) return false;
        Z = z;
        R = m / z;
        C = n / z;
        present.assign((size_t) R * C, 0);
        shift.assign((size_t) R * C, 0);
        for (int i = 0; i < R; i++)
            for (int j = 0; j < C; j++) {
                int s = -1;
                for (int k = 0; k < Z; k++)
                    for (int l = 0; l < Z; l++)
                        if (H[(size_t) (i * Z + k) * n + (size_t) j * Z + l]) {
                            const int ns = (l - k + Z) % Z;
                            if (s >= 0 && s != ns) return false;
                            s = ns;
                        }
                if (s >= 0) {
                    present[(size_t) i * C + j] = 1;
                    shift[(size_t) i * C + j] = s;
                }
            }
        return dense() == H;
    }

    // one mutation (optimize_H.cpp:66-75): pick a block; flip its presence if it is absent or with probability 1/2;
    // redraw its shift.  at_out (optional): the block that was drawn
    template <typename Gen>
    QcMatrix mutated(Gen &rnd, size_t *at_out = nullptr) const {
        QcMatrix q = *this;
        const int i = (int) (rnd() % (unsigned) R);
        const int j = (int) (rnd() % (unsigned) C);
        const size_t at = (size_t) i * C + j;
        if (!q.present[at] || rnd() % 2 == 0) q.present[at] = !q.present[at];
        q.shift[at] = (int) (rnd() % (unsigned) Z);
        if (at_out) *at_out = at;
        return q;
    }
};

struct Scorer {
    double alpha, mu, snr;
    int iters;
    int64_t tests;
    int noise;
    // FER(H) of optimize_H.cpp:16-25: codewords from GetOrtogonal + mt19937(239)
    double fer(const QcMatrix &q, int64_t ntests) const {
        const int m = q.R * q.Z, n = q.C * q.Z;
        std::vector<uint8_t> H = q.dense();
        acg_ldpc_code *code = nullptr;
        if (acg_ldpc_code_from_dense(H.data(), m, n, &code)) drv::die("code_from_dense");
        bool ok;
        std::vector<uint8_t> cws = drv::make_codewords(code, nullptr, 239u, ntests, &ok);
        double f = 1.0;
        if (ok) {
            int nv, nc, nz;
            double e_min, e_max;
            acg_ldpc_code_admm_shape(code, &nv, &nc, &nz, &e_min, &e_max);
            if (!(e_min * mu <= alpha)) {
                acg_ldpc_params p;
                acg_ldpc_params_default(&p);
                p.algo = ACG_LDPC_QPADMM;
                p.alpha = alpha;
                p.mu = mu;
                p.max_iter = iters;
                p.eps_stop = 1e-5;
                p.fast_setup = 1;  // one decoder per proposal: skip the static LDS placement search
                acg_ldpc_decoder *d = nullptr;
                if (acg_ldpc_decoder_create(code, &p, &d)) drv::die("create");
                f = drv::run_mc(d, cws, n, snr, ntests, noise, 1).fer();
                acg_ldpc_decoder_destroy(d);
            }
        }
        acg_ldpc_code_destroy(code);
        return f;
    }
};

int main(int argc, char **argv) {
    drv::Args a{argc, argv};
    const int Z = (int) a.integer("--Z", 20);
    Scorer sc{a.num("--alpha", 1.95), a.num("--mu", 0.5), a.num("--snr", -3.0), (int) a.integer("--admm-iters", 1000),
              a.integer("--tests", 1000),
              std::strcmp(a.get("--noise", "host"), "device") ? ACG_LDPC_NOISE_HOST_MT19937 : ACG_LDPC_NOISE_DEVICE_PHILOX};
    const char *out = a.get("--out", "optimalH.txt");
    std::mt19937 rnd((uint32_t) a.integer("--seed", 239));  // optimize_H.cpp:132
    QcMatrix q;
    if (a.get("--init")) {
        acg_ldpc_code *code = nullptr;
        if (acg_ldpc_code_load_txt(a.get("--init"), &code)) drv::die("read_pcm");
        int m, n, E;
        acg_ldpc_code_dims(code, &m, &n, &E);
        std::vector<uint8_t> H((size_t) m * n);
        acg_ldpc_code_dense(code, H.data());
        acg_ldpc_code_destroy(code);
        if (!q.from_dense(H, m, n, Z)) {
            std::fprintf(stderr, "%s is not quasi-cyclic with %d x %d circulants\n", a.get("--init"), Z, Z);
            return 1;
        }
    } else {
        // random start (optimize_H.cpp:106-122): density-1/2 block mask, uniform shifts, redrawn until a generator exists
        std::vector<double> rc = drv::parse_list(a.get("--random", "8,14"));
        q.Z = Z;
        q.R = (int) rc[0];
        q.C = (int) rc[1];
        for (;;) {
            q.present.assign((size_t) q.R * q.C, 0);
            q.shift.assign((size_t) q.R * q.C, 0);
            for (size_t k = 0; k < q.present.size(); k++) {
                q.present[k] = rnd() % 2;
                q.shift[k] = (int) (rnd() % (unsigned) Z);
            }
            std::vector<uint8_t> H = q.dense();
            acg_ldpc_code *code = nullptr;
            if (acg_ldpc_code_from_dense(H.data(), q.R * Z, q.C * Z, &code)) drv::die("code");
            std::vector<uint8_t> G((size_t) (q.C - q.R) * Z * q.C * Z);
            const int rcg = acg_ldpc_code_generator(code, G.data());
            acg_ldpc_code_destroy(code);
            if (rcg == 0) break;
        }
    }
    if (a.has("--check-qc")) {  // host-only: print the protograph (shift or -1 per block) and stop
        std::printf("Z=%d R=%d C=%d\n", q.Z, q.R, q.C);
        for (int i = 0; i < q.R; i++) {
            for (int j = 0; j < q.C; j++) std::printf("%d%c", q.present[(size_t) i * q.C + j] ? q.shift[(size_t) i * q.C + j] : -1, j + 1 == q.C ? '\n' : ' ');
        }
        return 0;
    }
    if (a.has("--dump-proposals")) {  // host-only: the proposal sequence of optimize_H.cpp:89-104 without scoring
        const int cnt = (int) a.integer("--dump-proposals", 16);
        for (int it = 0; it < cnt; it++) {
            size_t at = 0;
            QcMatrix cand = q.mutated(rnd, &at);
            std::printf("%d %d %d %d\n", (int) (at / (size_t) q.C), (int) (at % (size_t) q.C), (int) cand.present[at], cand.shift[at]);
            if (a.has("--accept-all")) q = cand;
        }
        return 0;
    }
    std::cout.precision(5);
    std::cout << std::fixed;
    double error = sc.fer(q, sc.tests);
    std::cout << "initial FER=" << error << std::endl;
    const int iters = (int) a.integer("--iters", 10000);
    for (int it = 0; it < iters; it++) {
        QcMatrix cand = q.mutated(rnd);
        const double e = sc.fer(cand, sc.tests);
        std::cout << "\tproposal: FER=" << e << std::endl;
        if (e < error) {  // optimize_H.cpp:96-101
            q = cand;
            error = e;
            std::cout << "accept, FER=" << error << std::endl;
            std::vector<uint8_t> H = q.dense();
            acg_ldpc_code *code = nullptr;
            if (acg_ldpc_code_from_dense(H.data(), q.R * Z, q.C * Z, &code)) drv::die("code");
            if (acg_ldpc_code_save_txt(code, out)) drv::die("save_matrix");
            acg_ldpc_code_destroy(code);
        }
    }
    if (a.has("--final-tests")) std::cout << sc.fer(q, a.integer("--final-tests", 10000)) << std::endl;  // optimize_H.cpp:135
    return 0;
}
