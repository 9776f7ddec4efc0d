// N1 — QP-ADMM (alpha, mu) grid search: the reference's qpadmm_params.cpp:36-85 on the device.
//   acg_qpadmm_params [--H data/optimalH.txt] [--snr -3] [--tests 1000] [--iters 1000] [--alpha 0,3,61] [--mu 0,3,61]
//                     [--noise host|device]
// Defaults are the reference's: optimalH, codewords from mt19937(239), 1000 frames at -3 dB, QPADMMDecoder(a, mu,
// 1000, 1e-5) on a 61 x 61 grid over [0,3]^2 (qpadmm_params.cpp:12-17,51-58).  Grid points where the decoder's guard
// e_min*mu <= alpha fires decode nothing and score FER = 1 (every frame fails, qp_admm.h:112-114).
#include <iostream>

#include "common.hpp"

static double linear_function(double L, double R, int cnt, int i) { return L + ((R - L) / (cnt - 1)) * i; }  // :32-34

int main(int argc, char **argv) {
    drv::Args a{argc, argv};
    const char *hpath = a.get("--H", "data/optimalH.txt");
    const int64_t tests = a.integer("--tests", 1000);
    const double snr = a.num("--snr", -3.0);
    const int iters = (int) a.integer("--iters", 1000);
    std::vector<double> ga = drv::parse_list(a.get("--alpha", "0,3,61")), gm = drv::parse_list(a.get("--mu", "0,3,61"));
    const int noise = std::strcmp(a.get("--noise", "host"), "device") ? ACG_LDPC_NOISE_HOST_MT19937 : ACG_LDPC_NOISE_DEVICE_PHILOX;
    acg_ldpc_code *code = nullptr;
    if (acg_ldpc_code_load_txt(hpath, &code)) drv::die("read_pcm");
    int m, n, E;
    acg_ldpc_code_dims(code, &m, &n, &E);
    bool ok;
    std::vector<uint8_t> cws = drv::make_codewords(code, a.get("--G"), 239u, tests, &ok);  // qpadmm_params.cpp:46-47
    if (!ok) return 1;
    int nv, nc, nz;
    double e_min, e_max;
    acg_ldpc_code_admm_shape(code, &nv, &nc, &nz, &e_min, &e_max);
    std::cerr << "n=" << n << " k=" << m << std::endl;
    std::cout.precision(5);
    std::cout << std::fixed;
    double best_fer = 2.0, best_alpha = -1, best_mu = -1;
    const int acnt = (int) ga[2], mcnt = (int) gm[2];
    for (int ai = 0; ai < acnt; ++ai)
        for (int mi = 0; mi < mcnt; ++mi) {
            const double alpha = linear_function(ga[0], ga[1], acnt, ai), mu = linear_function(gm[0], gm[1], mcnt, mi);
            double fer = 1.0;
            if (!(e_min * mu <= alpha)) {  // otherwise every decode returns (zeros, false): FER 1 without launching
                acg_ldpc_params p;
                acg_ldpc_params_default(&p);
                p.algo = ACG_LDPC_QPADMM;
                p.alpha = alpha;
                p.mu = mu;
                p.max_iter = iters;
                p.eps_stop = 1e-5;
                acg_ldpc_decoder *d = nullptr;
                if (acg_ldpc_decoder_create(code, &p, &d)) drv::die("create");
                fer = drv::run_mc(d, cws, n, snr, tests, noise, 1).fer();
                acg_ldpc_decoder_destroy(d);
            }
            std::cerr << "alpha=" << alpha << ", mu=" << mu << ": fer=" << fer << std::endl;
            if (fer < best_fer) {
                best_fer = fer;
                best_alpha = alpha;
                best_mu = mu;
                std::cout << "new best fer found: " << fer << "| alpha=" << alpha << ", mu=" << mu << std::endl;
            }
        }
    std::cout << "Best parameters:" << std::endl;
    std::cout << "alpha=" << best_alpha << std::endl;
    std::cout << "mu=" << best_mu << std::endl;
    std::cout << "fer=" << best_fer << std::endl;
    acg_ldpc_code_destroy(code);
    return 0;
}
