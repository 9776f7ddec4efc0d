// acg_ldpc_decoder.hpp — C++ mirror of the reference's Decoder operator interface over the C ABI.
//
// Same class names, constructor arguments, decode() signature and return convention as the reference:
//   class Decoder                          algo/algo.h:6-11
//   class BeliefPropagationDecoder(int)    algo/bp.h:208-222      name() == "BP"
//   class QPADMMDecoder(a, mu, it, eps)    algo/qp_admm.h:180-194 name() == "QP-ADMM"
// so experiment.h / main.cpp style callers compile unchanged against these types.
// Header-only; link with -lacg_ldpc_hip.  The analysed graph is cached on the CONTENT of H, because the reference passes H
// to every call (SURVEY §8b "Inputs"); the cache is a small LRU (kMaxHandles), because the reference's optimize_H loop
// hands a NEW H per proposal to one shared decoder (optimize_H.cpp:16-25,89-104).
#pragma once

#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "acg_ldpc.h"

namespace acg_ldpc {

typedef std::vector<bool> TCodeword;       // utils/codeword.h:17
typedef std::vector<TCodeword> TMatrix;    // utils/codeword.h:18
typedef std::vector<double> TFVector;      // utils/channel.h:8

class Decoder {  // algo/algo.h:6-11
public:
    virtual ~Decoder() {}
    virtual std::pair<TCodeword, bool> decode(const TMatrix &H, const TFVector &channel_word, double snr) = 0;
    virtual std::string name() const = 0;
};

class HipDecoderBase : public Decoder {
public:
    // live device handles per decoder object: least recently used ones beyond this are destroyed
    static constexpr size_t kMaxHandles = 8;

    ~HipDecoderBase() override {
        for (auto &e : cache_) drop(e);
    }

    std::pair<TCodeword, bool> decode(const TMatrix &H, const TFVector &channel_word, double snr) override {
        // the handle stays checked out for the duration of the call, so a concurrent decode() with another H (the
        // reference calls one decoder object from THREADS_NUM pthreads, experiment.h:101) cannot evict it under us
        Lease l(this, H);
        const int n = (int) H[0].size();
        assert((int) channel_word.size() == n);
        std::vector<uint8_t> bits((size_t) n);
        uint8_t ok = 0;
        int rc = acg_ldpc_decode_batch(l.dec(), channel_word.data(), 1, snr, bits.data(), &ok, nullptr);
        check(rc);
        return finish(bits, ok != 0);
    }

    // batched form: Y is frames*n doubles; returns per-frame (word, flag) exactly as decode() would
    std::vector<std::pair<TCodeword, bool>> decode_batch(const TMatrix &H, const std::vector<double> &Y, double snr) {
        Lease l(this, H);
        const int n = (int) H[0].size();
        const int64_t frames = (int64_t) (Y.size() / (size_t) n);
        std::vector<uint8_t> bits((size_t) frames * n), ok((size_t) frames);
        check(acg_ldpc_decode_batch(l.dec(), Y.data(), frames, snr, bits.data(), ok.data(), nullptr));
        std::vector<std::pair<TCodeword, bool>> out;
        out.reserve((size_t) frames);
        for (int64_t f = 0; f < frames; f++)
            out.push_back(finish(std::vector<uint8_t>(bits.begin() + f * n, bits.begin() + (f + 1) * n), ok[f] != 0));
        return out;
    }

    // the C handle for Monte-Carlo runs (acg_ldpc_mc_run).  The reference hands H to every decode() and re-analyses it
    // (bp.h:136-153, qp_admm.h:15-21); here the analysed graph is cached on the CONTENT of H (hash of the packed words of
    // vector<bool> + a full compare of the packed copy on a hit), so a matrix that is modified or whose storage is recycled
    // can never meet a stale handle.  The returned pointer stays valid until kMaxHandles OTHER matrices have been used
    // through this object; decode() / decode_batch() pin theirs for the duration of the call.
    acg_ldpc_decoder *handle(const TMatrix &H) {
        std::lock_guard<std::mutex> lk(mu_);
        return lookup(H)->dec;
    }

    size_t live_handles() const {
        std::lock_guard<std::mutex> lk(mu_);
        return cache_.size();
    }

protected:
    virtual void fill(acg_ldpc_params &p) const = 0;
    virtual std::pair<TCodeword, bool> finish(const std::vector<uint8_t> &bits, bool ok) const = 0;

    static void check(int rc) {
        if (rc != 0) {  // the reference aborts through assert(); so does the adaptor, with the library's message
            std::fprintf(stderr, "acg_ldpc: error %d: %s\n", rc, acg_ldpc_last_error());
            std::abort();
        }
    }

private:
    struct Entry {
        int m = 0, n = 0;
        uint64_t h = 0;
        std::vector<uint64_t> packed;  // row-major, every row padded to whole 64-bit words
        acg_ldpc_code *code = nullptr;
        acg_ldpc_decoder *dec = nullptr;
        int pins = 0;                  // decode() calls in flight on this handle
    };
    typedef std::list<Entry>::iterator It;

    struct Lease {  // keeps one entry out of the eviction's reach while a call uses its handle
        Lease(HipDecoderBase *o, const TMatrix &H) : o_(o) {
            std::lock_guard<std::mutex> lk(o_->mu_);
            it_ = o_->lookup(H);
            it_->pins++;
        }
        ~Lease() {
            std::lock_guard<std::mutex> lk(o_->mu_);
            it_->pins--;
            o_->trim();
        }
        acg_ldpc_decoder *dec() const { return it_->dec; }
        HipDecoderBase *o_;
        It it_;
    };

    static void pack(const TMatrix &H, int m, int n, std::vector<uint64_t> &out, uint64_t &h) {
        const int wpr = (n + 63) / 64;
        out.assign((size_t) m * wpr, 0);
        h = 1469598103934665603ull;
        for (int i = 0; i < m; i++) {
            uint64_t *row = out.data() + (size_t) i * wpr;
#if defined(__GLIBCXX__)
            // whole 64-bit words of the row, then the tail bits (bits beyond size() in the last word are unspecified)
            static_assert(sizeof(unsigned long) == 8, "packed copy assumes 64-bit words");
            const unsigned long *wp = H[i].begin()._M_p;
            const int full = n / 64;
            for (int w = 0; w < full; w++) row[w] = (uint64_t) wp[w];
            for (int j = full * 64; j < n; j++) row[j >> 6] |= (uint64_t) H[i][j] << (j & 63);
#else
            for (int j = 0; j < n; j++) row[j >> 6] |= (uint64_t) H[i][j] << (j & 63);
#endif
            for (int w = 0; w < wpr; w++) h = (h ^ row[w]) * 1099511628211ull;
            h = (h ^ 0x9E3779B97F4A7C15ull) * 1099511628211ull;  // row separator
        }
    }

    // caller holds mu_.  Hit: move to the front.  Miss: analyse H, create the device handle, evict beyond kMaxHandles.
    It lookup(const TMatrix &H) {
        const int m = (int) H.size(), n = (int) H[0].size();
        pack(H, m, n, scratch_, scratch_h_);
        for (It it = cache_.begin(); it != cache_.end(); ++it)
            if (it->m == m && it->n == n && it->h == scratch_h_ && it->packed == scratch_) {
                cache_.splice(cache_.begin(), cache_, it);
                return cache_.begin();
            }
        std::vector<uint8_t> dense((size_t) m * n);
        for (int i = 0; i < m; i++)
            for (int j = 0; j < n; j++) dense[(size_t) i * n + j] = H[i][j];
        Entry e;
        e.m = m;
        e.n = n;
        e.h = scratch_h_;
        e.packed = scratch_;
        check(acg_ldpc_code_from_dense(dense.data(), m, n, &e.code));
        acg_ldpc_params p;
        acg_ldpc_params_default(&p);
        fill(p);
        check(acg_ldpc_decoder_create(e.code, &p, &e.dec));
        cache_.push_front(e);
        trim();
        return cache_.begin();
    }

    void trim() {  // caller holds mu_: drop least recently used entries that no call is using
        for (It it = cache_.end(); cache_.size() > kMaxHandles && it != cache_.begin();) {
            --it;
            if (it->pins == 0 && it != cache_.begin()) {
                drop(*it);
                it = cache_.erase(it);
            }
        }
    }

    static void drop(Entry &e) {
        acg_ldpc_decoder_destroy(e.dec);
        acg_ldpc_code_destroy(e.code);
        e.dec = nullptr;
        e.code = nullptr;
    }

    std::list<Entry> cache_;
    std::vector<uint64_t> scratch_;
    uint64_t scratch_h_ = 0;
    mutable std::mutex mu_;
};

// algo/bp.h:208-222
class BeliefPropagationDecoder : public HipDecoderBase {
public:
    // layered (build-added, default off = the reference's flooding schedule, bit-exact hard decisions): the same check rule with the
    // posteriors updated per block row — the reference's FER at about half the iterations, FER-level parity only
    explicit BeliefPropagationDecoder(int max_iter, bool layered = false) : _max_iter(max_iter), _layered(layered) {}
    std::string name() const override { return "BP"; }  // bp.h:218

protected:
    void fill(acg_ldpc_params &p) const override {
        p.algo = ACG_LDPC_BP_SUMPRODUCT;
        p.max_iter = _max_iter;
        p.schedule = _layered ? ACG_LDPC_SCHEDULE_LAYERED : ACG_LDPC_SCHEDULE_FLOODING;
    }
    std::pair<TCodeword, bool> finish(const std::vector<uint8_t> &bits, bool ok) const override {
        if (!ok) return {TCodeword(), false};  // bp.h:198
        return {TCodeword(bits.begin(), bits.end()), true};
    }

private:
    int _max_iter;
    bool _layered;
};

// algo/qp_admm.h:180-194 (same defaults)
class QPADMMDecoder : public HipDecoderBase {
public:
    explicit QPADMMDecoder(double alpha, double mu, int max_iter = 2000, double eps_stop = 1e-5)
        : _alpha(alpha), _mu(mu), _eps_stop(eps_stop), _max_iter(max_iter) {}
    std::string name() const override { return "QP-ADMM"; }  // qp_admm.h:189

protected:
    void fill(acg_ldpc_params &p) const override {
        p.algo = ACG_LDPC_QPADMM;
        p.max_iter = _max_iter;
        p.alpha = _alpha;
        p.mu = _mu;
        p.eps_stop = _eps_stop;
    }
    std::pair<TCodeword, bool> finish(const std::vector<uint8_t> &bits, bool ok) const override {
        return {TCodeword(bits.begin(), bits.end()), ok};  // qp_admm.h:112-114 (zeros,false) / :177 (word,true)
    }

private:
    double _alpha, _mu, _eps_stop;
    int _max_iter;
};

// build-added (north_star); not in the reference: parity unpinned.  layered = true: ACG_LDPC_SCHEDULE_LAYERED (about half the
// iterations for the same FER; FER-level parity only)
class MinSumDecoder : public HipDecoderBase {
public:
    explicit MinSumDecoder(int max_iter, double scale = 1.0, bool layered = false) : _max_iter(max_iter), _scale(scale), _layered(layered) {}
    std::string name() const override { return "MS"; }

protected:
    void fill(acg_ldpc_params &p) const override {
        p.algo = ACG_LDPC_BP_MINSUM;
        p.max_iter = _max_iter;
        p.ms_scale = _scale;
        p.schedule = _layered ? ACG_LDPC_SCHEDULE_LAYERED : ACG_LDPC_SCHEDULE_FLOODING;
    }
    std::pair<TCodeword, bool> finish(const std::vector<uint8_t> &bits, bool ok) const override {
        if (!ok) return {TCodeword(), false};
        return {TCodeword(bits.begin(), bits.end()), true};
    }

private:
    int _max_iter;
    double _scale;
    bool _layered;
};

}  // namespace acg_ldpc
