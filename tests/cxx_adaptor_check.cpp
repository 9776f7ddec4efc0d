// Compiled (and, on a GPU box, run) by tests/test_cxx_adaptor.py: the C++ mirror of the reference's
// Decoder interface must compile with plain g++ against include/ and behave like algo/algo.h:6-11.
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>

#include "acg_ldpc_decoder.hpp"

using namespace acg_ldpc;

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
