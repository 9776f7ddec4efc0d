// Compiled (and, on a GPU box, run) by tests/test_cxx_adaptor.py: the C++ mirror of the reference's
// Decoder interface must compile with plain g++ against include/ and behave like algo/algo.h:6-11.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <string>
#include <thread>

#include "acg_ldpc_decoder.hpp"

using namespace acg_ldpc;

static int thread_count() {
    std::ifstream f("/proc/self/status");
    std::string line;
    while (std::getline(f, line))
        if (line.rfind("Threads:", 0) == 0) return std::atoi(line.c_str() + 8);
    return -1;
}

// The optimize_H.cpp:16-25,89-104 pattern: ONE decoder object, a new H per proposal.  100 distinct matrices (column
// rotations of the base matrix: a codeword of H rotated the same way is a codeword of the rotated H), driven from 4 host
// threads at once like experiment.h:127-130 drives one decoder from THREADS_NUM pthreads.
static int lru_check(const TMatrix &H0, const std::vector<uint8_t> &cw0) {
    const int m = (int) H0.size(), n = (int) H0[0].size();
    BeliefPropagationDecoder bp(30);
    QPADMMDecoder admm(1.2, 0.55, 200, 1e-5);
    HipDecoderBase *decs[2] = {&bp, &admm};
    int fails = 0, t_mid = 0;
    std::mutex mu;
    auto work = [&](int tid) {
        for (int r = tid; r < 100; r += 4) {
            TMatrix H(m, TCodeword(n));
            std::vector<uint8_t> cw((size_t) n);
            for (int i = 0; i < m; i++)
                for (int j = 0; j < n; j++) H[i][(j + r) % n] = H0[i][j];
            for (int j = 0; j < n; j++) cw[(size_t) ((j + r) % n)] = cw0[(size_t) j];
            TFVector y((size_t) n);
            acg_ldpc_transmit_host(cw.data(), 1, n, r, 1, 5.0, y.data());
            for (auto *d : decs) {
                auto p = d->decode(H, y, 5.0);
                int diff = p.second ? 0 : 1;
                if (p.second)
                    for (int i = 0; i < n; i++) diff += (p.first[i] != (bool) cw[i]);
                std::lock_guard<std::mutex> lk(mu);
                if (diff) fails++;
                if (d->live_handles() > HipDecoderBase::kMaxHandles + 4) fails += 1000;  // + the handles pinned by the 4 threads
            }
            if (r == 40) t_mid = thread_count();
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.emplace_back(work, t);
    for (auto &t : th) t.join();
    const int t_end = thread_count();
    std::printf("lru: fails=%d live=%zu/%zu threads mid=%d end=%d\n", fails, bp.live_handles(), admm.live_handles(), t_mid, t_end);
    if (fails || bp.live_handles() > HipDecoderBase::kMaxHandles || admm.live_handles() > HipDecoderBase::kMaxHandles) return 7;
    if (t_end > t_mid + 2) return 8;  // threads must not accumulate with the number of matrices seen
    std::printf("lru ok\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    acg_ldpc_code *code = nullptr;
    if (acg_ldpc_code_load_txt(argv[1], &code)) {
        std::fprintf(stderr, "%s\n", acg_ldpc_last_error());
        return 3;
    }
    int m, n, E;
    acg_ldpc_code_dims(code, &m, &n, &E);
    std::vector<uint8_t> dense((size_t) m * n);
    acg_ldpc_code_dense(code, dense.data());
    TMatrix H(m, TCodeword(n));
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) H[i][j] = dense[(size_t) i * n + j];
    std::printf("m=%d n=%d E=%d\n", m, n, E);
    if (!acg_ldpc_device_available()) {
        std::printf("no device: compile/link check only\n");
        return 0;
    }
    if (argc > 2 && std::string(argv[2]) == "lru") {
        std::vector<uint8_t> G0((size_t) (n - m) * n), cw0((size_t) n);
        if (acg_ldpc_code_generator(code, G0.data())) return 4;
        acg_ldpc_gen_codewords(G0.data(), n - m, n, 239239239u, 1, cw0.data());
        return lru_check(H, cw0);
    }
    // same shape as main.cpp:28-40: a list of shared_ptr<Decoder>
    std::vector<std::shared_ptr<Decoder>> decoders{std::make_shared<BeliefPropagationDecoder>(50),
                                                   std::make_shared<QPADMMDecoder>(1.95, 0.5, 100, 1e-5)};
    std::vector<uint8_t> G((size_t) (n - m) * n), cw((size_t) n);
    if (acg_ldpc_code_generator(code, G.data())) return 4;
    acg_ldpc_gen_codewords(G.data(), n - m, n, 239239239u, 1, cw.data());
    TFVector y((size_t) n);
    acg_ldpc_transmit_host(cw.data(), 1, n, 0, 1, 0.0, y.data());
    for (auto &d : decoders) {
        auto p = d->decode(H, y, 0.0);
        int diff = 0;
        if (p.second)
            for (int i = 0; i < n; i++) diff += (p.first[i] != (bool) cw[i]);
        std::printf("%s ok=%d size=%zu diff=%d\n", d->name().c_str(), (int) p.second, p.first.size(), diff);
        if (!p.second || diff) return 5;
    }
    TFVector bad((size_t) n, -0.05);  // hopeless word: BP must return the EMPTY vector + false (bp.h:198)
    for (int i = 0; i < n; i += 3) bad[i] = 0.07;
    auto p = decoders[0]->decode(H, bad, -5.0);
    std::printf("BP hopeless ok=%d size=%zu\n", (int) p.second, p.first.size());
    if (p.second || !p.first.empty()) return 6;
    acg_ldpc_code_destroy(code);
    return 0;
}
