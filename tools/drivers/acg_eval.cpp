// N2 — evaluation driver: the reference's main.cpp:42-92 loop (decoder list x SNR list -> stdout + report.csv with
// the same header and 12-digit fixed precision, main.cpp:47-49,79-86) on the device.
//
//   acg_eval --H data/optimalH.txt [--G data/G05.txt] [--snrs -5,-4.5,...,0] [--tests 10000] [--bp-iters 100]
//            [--alpha 1.2 --mu 0.55 --admm-iters 10000 --eps 1e-5] [--noise host|device] [--seed 1] [--out report.csv]
//            [--gpus N]   (one decoder handle + host thread per GPU, contiguous frame ranges, counters summed)
//            [--minsum-iters N [--ms-scale 0.75] [--layered] [--ms-f16]]   also evaluate the build-added normalised min-sum
//                         (NOT in the reference: parity unpinned), flooding or with the layered schedule
// Defaults are main.cpp's (OPTIMAL build): optimalH, BP(100), QP-ADMM(1.2, 0.55, 10000, 1e-5), 10000 codewords from
// mt19937(239'239'239), SNRs -5..0 step 0.5.  --noise host reproduces the reference's frames bit for bit
// (frame i <- mt19937(i+1)); --noise device keeps generation, decoding and classification on the GPU.
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "common.hpp"

int main(int argc, char **argv) {
    drv::Args a{argc, argv};
    const char *hpath = a.get("--H", "data/optimalH.txt");
    const int64_t tests = a.integer("--tests", 10000);                     // TESTS_NUM, main.cpp:25
    std::vector<double> snrs = drv::parse_list(a.get("--snrs", "-5,-4.5,-4,-3.5,-3,-2.5,-2,-1.5,-1,-0.5,0"));  // main.cpp:27
    const int noise = std::strcmp(a.get("--noise", "host"), "device") ? ACG_LDPC_NOISE_HOST_MT19937 : ACG_LDPC_NOISE_DEVICE_PHILOX;
    acg_ldpc_code *code = nullptr;
    if (acg_ldpc_code_load_txt(hpath, &code)) drv::die("read_pcm");
    int m, n, E;
    acg_ldpc_code_dims(code, &m, &n, &E);
    bool ok;
    std::vector<uint8_t> cws = drv::make_codewords(code, a.get("--G"), 239239239u, tests, &ok);  // main.cpp:63-64
    if (!ok) {
        std::fprintf(stderr, "GetOrtogonal failed for %s\n", hpath);
        return 1;
    }
    std::cerr << "n=" << n << " k=" << m << "\n";  // main.cpp:66 (prints H.size() as k)

    const int gpus = (int) a.integer("--gpus", 1);
    const int ndev = (int) a.integer("--device-count", 1 << 30);  // tests: fold N handles onto fewer devices
    std::vector<drv::MultiGpu> decs;
    acg_ldpc_params p;
    if (!a.has("--no-bp")) {
        acg_ldpc_params_default(&p);
        p.algo = ACG_LDPC_BP_SUMPRODUCT;
        p.max_iter = (int) a.integer("--bp-iters", 100);  // main.cpp:29
        decs.emplace_back();
        decs.back().create(code, p, gpus, ndev);
    }
    if (!a.has("--no-admm")) {
        acg_ldpc_params_default(&p);
        p.algo = ACG_LDPC_QPADMM;
        p.alpha = a.num("--alpha", 1.2);  // main.cpp:31
        p.mu = a.num("--mu", 0.55);
        p.max_iter = (int) a.integer("--admm-iters", 10000);
        p.eps_stop = a.num("--eps", 1e-5);
        decs.emplace_back();
        decs.back().create(code, p, gpus, ndev);
    }

    if (a.has("--minsum-iters")) {  // build-added variant (north_star); not one of main.cpp's decoders
        acg_ldpc_params_default(&p);
        p.algo = ACG_LDPC_BP_MINSUM;
        p.max_iter = (int) a.integer("--minsum-iters", 50);
        p.ms_scale = a.num("--ms-scale", 0.75);
        if (a.has("--layered")) p.schedule = ACG_LDPC_SCHEDULE_LAYERED;
        if (a.has("--ms-f16")) p.precision = ACG_LDPC_PREC_F16;
        decs.emplace_back();
        decs.back().create(code, p, gpus, ndev);
    }

    std::cout.precision(5);
    std::cout << std::fixed;
    std::ofstream fdata(a.get("--out", "report.csv"));
    fdata << "Method,SNR,Sigma,FER,Time,AvgHamming,AvgHammingCorrect,AvgHammingWrong" << std::endl;  // main.cpp:48
    fdata << std::fixed << std::setprecision(12);
    for (double snr : snrs) std::cerr << "snr=" << snr << ": var=" << acg_ldpc_llr_variance(snr) << std::endl;
    for (auto &d : decs) {
        std::cout << "Algo: " << d.name() << std::endl;
        for (double snr : snrs) {
            drv::McOut r = d.run(cws, n, snr, tests, noise, (uint64_t) a.integer("--seed", 1));
            std::cout << "\tSNR: " << snr << ", FER: " << r.fer() << ", (time=" << r.avg_time() << "s)" << std::endl;
            std::cerr << "\t\tAverage hamming distance: " << r.mean_hamming() << std::endl;
            fdata << d.name() << "," << snr << "," << std::sqrt(acg_ldpc_llr_variance(snr)) << "," << r.fer() << ","
                  << r.avg_time() << "," << r.mean_hamming() << "," << r.mean_hamming_ok() << ","
                  << r.mean_hamming_wrong() << std::endl;
        }
        std::cerr << std::string(30, '_') << std::endl;
    }
    for (auto &d : decs) d.destroy();
    acg_ldpc_code_destroy(code);
    return 0;
}
