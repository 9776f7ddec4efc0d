// Shared helpers of the "next row" drivers: thin C++ callers of the C ABI that replace the
// reference's main.cpp / qpadmm_params.cpp / optimize_H.cpp loops.  No decoding logic lives here.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "acg_ldpc.h"

namespace drv {

inline void die(const char *what) {
    std::fprintf(stderr, "%s: %s\n", what, acg_ldpc_last_error());
    std::exit(1);
}

struct Args {
    int argc;
    char **argv;
    const char *get(const char *key, const char *def = nullptr) const {
        for (int i = 1; i + 1 < argc; i++)
            if (!std::strcmp(argv[i], key)) return argv[i + 1];
        return def;
    }
    bool has(const char *key) const {
        for (int i = 1; i < argc; i++)
            if (!std::strcmp(argv[i], key)) return true;
        return false;
    }
    double num(const char *key, double def) const {
        const char *v = get(key);
        return v ? std::atof(v) : def;
    }
    long integer(const char *key, long def) const {
        const char *v = get(key);
        return v ? std::atol(v) : def;
    }
};

inline std::vector<double> parse_list(const char *s) {
    std::vector<double> out;
    std::string cur;
    for (const char *p = s;; ++p) {
        if (*p == ',' || *p == 0) {
            if (!cur.empty()) out.push_back(std::atof(cur.c_str()));
            cur.clear();
            if (*p == 0) break;
        } else
            cur.push_back(*p);
    }
    return out;
}

struct McOut {
    acg_ldpc_mc_result r;
    double fer() const { return (double) (r.total - r.correct) / (double) r.total; }            // experiment.h:59
    double avg_time() const { return r.time_sec / (double) r.total; }                           // :61
    double mean_hamming() const { return (double) r.sum_hamming / (double) r.total; }           // :63
    double mean_hamming_ok() const { return (double) r.sum_hamming_ok / (double) (r.correct > 1 ? r.correct : 1); }
    double mean_hamming_wrong() const {
        const int64_t w = r.total - r.correct;
        return (double) r.sum_hamming_wrong / (double) (w > 1 ? w : 1);
    }
};

// multithread_experiment (experiment.h:125-139) on the device
inline McOut run_mc(acg_ldpc_decoder *dec, const std::vector<uint8_t> &codewords, int n, double snr, int64_t frames,
                    int noise, uint64_t seed) {
    acg_ldpc_mc_cfg cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.frames = frames;
    cfg.snr = snr;
    cfg.seed = seed;
    cfg.noise = noise;
    cfg.codewords = codewords.empty() ? nullptr : codewords.data();
    cfg.n_codewords = codewords.empty() ? 0 : (int64_t) (codewords.size() / (size_t) n);
    McOut o;
    if (acg_ldpc_mc_run(dec, &cfg, &o.r)) die("acg_ldpc_mc_run");
    return o;
}

// codewords the way every reference driver makes them: G = GetOrtogonal(H) (or a G file), gen_random_codewords(G, mt19937(seed))
inline std::vector<uint8_t> make_codewords(const acg_ldpc_code *code, const char *gpath, uint32_t seed, int64_t count, bool *ok) {
    int m, n, E;
    acg_ldpc_code_dims(code, &m, &n, &E);
    std::vector<uint8_t> G;
    int k = n - m;
    *ok = true;
    if (gpath) {
        acg_ldpc_code *g = nullptr;
        if (acg_ldpc_code_load_txt(gpath, &g)) die("load G");
        int gm, gn, ge;
        acg_ldpc_code_dims(g, &gm, &gn, &ge);
        if (gn != n) {
            std::fprintf(stderr, "G has %d columns, H has %d\n", gn, n);
            std::exit(1);
        }
        k = gm;
        G.resize((size_t) gm * gn);
        acg_ldpc_code_dense(g, G.data());
        acg_ldpc_code_destroy(g);
    } else {
        G.resize((size_t) k * n);
        const int rc = acg_ldpc_code_generator(code, G.data());
        if (rc == 1) {
            *ok = false;  // GetOrtogonal's {TMatrix(), false}
            return {};
        }
        if (rc) die("generator");
    }
    std::vector<uint8_t> cw((size_t) count * n);
    if (acg_ldpc_gen_codewords(G.data(), k, n, seed, count, cw.data())) die("gen_codewords");
    return cw;
}

// One decoder handle per GPU; a Monte-Carlo run is cut into contiguous global frame ranges, one per handle, run from
// one host thread per device and merged with acg_ldpc_mc_merge (experiment.h:70-78) — frames are independent, so
// nothing but seven counters crosses devices.  Device k of `gpus` maps to ordinal k % device_count, so
// the sharding logic can be exercised on a one-GPU box.
struct MultiGpu {
    std::vector<acg_ldpc_decoder *> dec;

    void create(const acg_ldpc_code *code, acg_ldpc_params p, int gpus, int device_count) {
        for (int k = 0; k < gpus; k++) {
            p.device = device_count > 0 ? k % device_count : -1;
            acg_ldpc_decoder *d = nullptr;
            if (acg_ldpc_decoder_create(code, &p, &d)) die("acg_ldpc_decoder_create");
            dec.push_back(d);
        }
    }
    void destroy() {
        for (auto *d : dec) acg_ldpc_decoder_destroy(d);
        dec.clear();
    }
    const char *name() const { return acg_ldpc_decoder_name(dec[0]); }

    McOut run(const std::vector<uint8_t> &codewords, int n, double snr, int64_t frames, int noise, uint64_t seed) const {
        const int W = (int) dec.size();
        std::vector<McOut> part((size_t) W);
        std::vector<std::string> err((size_t) W);
        std::vector<std::thread> th;
        for (int r = 0; r < W; r++)
            th.emplace_back([&, r] {
                const int64_t lo = frames * r / W, hi = frames * (r + 1) / W;
                acg_ldpc_mc_cfg cfg;
                std::memset(&cfg, 0, sizeof cfg);
                cfg.frames = hi - lo;
                cfg.first_frame = lo;  // seeds derive from the GLOBAL frame index: same frames for any W
                cfg.snr = snr;
                cfg.seed = seed;
                cfg.noise = noise;
                cfg.codewords = codewords.empty() ? nullptr : codewords.data();
                cfg.n_codewords = codewords.empty() ? 0 : (int64_t) (codewords.size() / (size_t) n);
                if (acg_ldpc_mc_run(dec[(size_t) r], &cfg, &part[(size_t) r].r)) err[(size_t) r] = acg_ldpc_last_error();
            });
        for (auto &t : th) t.join();
        McOut tot;
        std::memset(&tot.r, 0, sizeof tot.r);
        double wall = 0;
        for (int r = 0; r < W; r++) {
            if (!err[(size_t) r].empty()) {
                std::fprintf(stderr, "acg_ldpc_mc_run (shard %d): %s\n", r, err[(size_t) r].c_str());
                std::exit(1);
            }
            wall = wall > part[(size_t) r].r.time_sec ? wall : part[(size_t) r].r.time_sec;
            acg_ldpc_mc_merge(&tot.r, &part[(size_t) r].r);
        }
        tot.r.time_sec = wall;  // shards run concurrently: the slowest one is the wall time
        return tot;
    }
};

}  // namespace drv
