// C ABI of libacg_ldpc_hip.so (include/acg_ldpc.h): handles, uploads, launches.
// There is deliberately NO CPU decode path in this library: without a HIP device every decoder
// entry point fails with an error code and message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/acg_ldpc.h"
#include "kernels.hpp"
#include "ldpc_internal.hpp"

namespace acg {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

hipError_t bp_launch(const void *kernel, const BpTables &t, const DecodeArgs &a, int grid, int block, size_t lds,
                     hipStream_t s);
const void *bp_kernel_ptr(int algo, int f64, int maxd, int L, bool mc, int variant);
hipError_t phi_debug_launch(const void *x, void *out, int n, int f64, hipStream_t s);
hipError_t awgn_launch(float *y, int64_t frames, int n, int nwords, int64_t first_frame, uint64_t seed,
                       const uint32_t *cw_packed, int64_t n_cw, float sigma, hipStream_t s);

const void *bp_block_kernel_ptr(int algo, int f64, int L, bool mc, bool idxlds, bool idxreg, bool regular);
const void *bp_kernel_ptr_dbg(int f64, int L);
const void *bp_block_kernel_ptr_dbg(int f64);
const void *bp_pair_kernel_ptr(int L, bool regular);
const void *bp_layered_kernel_ptr(int G, int waves, bool qc_arith, bool f16, int algo, bool mc);
hipError_t bp_layered_launch(const void *kernel, const LayerTables &t, const DecodeArgs &a, int grid, int block, size_t lds, hipStream_t s);
const void *bp_streamed_ptr(int algo, int f64);
const void *bp_streamed_ring_ptr(int algo, bool nt);
const void *bp_streamed_ring_ptr_dbg();
hipError_t bp_streamed_ring_launch(const void *kernel, const StreamTables &t, const DecodeArgs &a, uint32_t *ws, int grid, hipStream_t s);
hipError_t bp_streamed_launch(const void *kernel, const StreamTables &t, const DecodeArgs &a, uint32_t *ws, int grid,
                              int block, hipStream_t s);
hipError_t classify_launch(const float *y, const uint32_t *bits, const uint8_t *ok, const int32_t *iters, int64_t frames,
                           int n, int nwords, int64_t first_frame, const uint32_t *cw_packed, int64_t n_cw,
                           unsigned long long *counters, const int32_t *row_ptr, const int32_t *edge_var, int m,
                           hipStream_t s);

struct AdmmDevice;  // admm_kernels.hip
AdmmDevice *admm_device_create(const Code &c, const acg_ldpc_params &p, int cu_count, std::string &err);
void admm_device_destroy(AdmmDevice *d);
hipError_t admm_launch(AdmmDevice *d, const DecodeArgs &a, hipStream_t s, std::string &err);
void admm_device_layout(const AdmmDevice *d, int *lds_per_frame, int *lanes, int *frames_per_block, int *grid);
bool admm_device_unfused_mc(const AdmmDevice *d, const int32_t **row_ptr, const int32_t **edge_var);

#define HIP_OK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
            return 10;                                                                          \
        }                                                                                       \
    } while (0)

// No C++ exception may cross the extern "C" boundary (std::bad_alloc from a vector, std::system_error from std::thread, ...):
// every entry point that can throw runs its body through guarded() and reports an error code + message instead.
template <class F>
static int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_error("out of host memory");
        return 12;
    } catch (const std::exception &e) {
        set_error(std::string("internal exception: ") + e.what());
        return 13;
    } catch (...) {
        set_error("internal exception");
        return 13;
    }
}

template <typename T>
static int upload(const std::vector<T> &h, T **d, size_t min_elems = 1) {
    size_t n = std::max(h.size(), min_elems);
    HIP_OK(hipMalloc((void **) d, n * sizeof(T)));
    if (!h.empty()) HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// A few persistent host threads for the byte shuffling of the host-buffer entry points (pageable user memory -> pinned
// staging, packed words -> one byte per bit): at 25 M frames/s that is ~30-60 GB/s of memcpy, more than one core moves.
class HostPool {
public:
    explicit HostPool(int n) {
        for (int i = 0; i < n; i++) th_.emplace_back([this, i] { run(i); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int) th_.size(); }
    // fn(part, parts) on every worker thread; returns when all are done
    void run_all(const std::function<void(int, int)> &fn) {
        std::lock_guard<std::mutex> one(call_mu_);   // the pool is shared by every handle of the process: one job at a time
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        pending_ = (int) th_.size();
        gen_++;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run(int id) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)> *fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
            }
            (*fn)(id, (int) th_.size());
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *fn_ = nullptr;
    uint64_t gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

// ONE pool per process, created the first time a batch is large enough to use it (>= 4096 frames): a caller that hands a new
// H to every decode — the reference's optimize_H loop, one decoder handle per proposal — must not collect threads per handle.
static HostPool *host_pool() {
    static std::mutex mu;
    static HostPool *pool = nullptr;   // intentionally never destroyed (worker threads must not be joined from a static destructor)
    std::lock_guard<std::mutex> lk(mu);
    if (!pool) {
        const unsigned hc = std::thread::hardware_concurrency();
        pool = new HostPool((int) std::max(2u, std::min(16u, hc ? hc / 2 : 2u)));
    }
    return pool;
}

// Double-buffered staging of acg_ldpc_decode_batch / _f32: while the GPU works on chunk c (H2D, kernel, D2H on stream c % 2)
// the host threads fill the pinned buffer of chunk c + 1 and unpack chunk c - 1.
struct HostPipe {
    static constexpr int NBUF = 2;
    int64_t chunk = 0;       // frames per chunk the buffers are sized for
    size_t y_bytes = 0;      // bytes per frame of the symbol buffers
    void *pin_y[NBUF] = {};
    unsigned char *pin_out[NBUF] = {};  // [frames][nwords] words | [frames] sweep counts | [frames] flags of the chunk in flight:
    void *dev_y[NBUF] = {};             // one region, so the results come back in ONE device-to-host copy
    unsigned char *dev_out[NBUF] = {};
    hipStream_t stream[NBUF] = {};
    hipEvent_t done[NBUF] = {};
    void release() {
        for (int b = 0; b < NBUF; b++) {
            if (pin_y[b]) (void) hipHostFree(pin_y[b]);
            if (pin_out[b]) (void) hipHostFree(pin_out[b]);
            if (dev_y[b]) (void) hipFree(dev_y[b]);
            if (dev_out[b]) (void) hipFree(dev_out[b]);
            pin_y[b] = dev_y[b] = nullptr;
            pin_out[b] = dev_out[b] = nullptr;
        }
        chunk = 0;
    }
    ~HostPipe() {
        release();
        for (int b = 0; b < NBUF; b++) {
            if (done[b]) (void) hipEventDestroy(done[b]);
            if (stream[b]) (void) hipStreamDestroy(stream[b]);
        }
    }
};

}  // namespace acg

using namespace acg;

// Workspace of the streamed engine as separately created physical chunks mapped into one virtual range in a shuffled order
// (HIP virtual-memory-management API).  Why: see decoder_setup_streamed — a physically CONTIGUOUS backing of the slabs is the
// slow mode of bp_streamed_ring_kernel on slabs beyond the Infinity Cache (181 ms against 158 ms per launch on configs[4]),
// and plain hipMalloc hands out either kind depending on the allocation history of the process.
struct ScatteredAlloc {
    void *va = nullptr;
    size_t bytes = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<char> mapped;
    void release() {
        if (!va) return;
        for (size_t i = 0; i < handles.size(); i++) {
            if (mapped[i]) (void) hipMemUnmap((char *) va + i * chunk, chunk);
        }
        for (auto &h : handles) (void) hipMemRelease(h);
        (void) hipMemAddressFree(va, bytes);
        va = nullptr;
        handles.clear();
        mapped.clear();
    }
    // -> true on success (va usable, read/write from `dev`)
    // spread > 1: `spread` times as many physical chunks are created and only every spread-th is kept (the others are released
    // again at once), so the kept ones are spaced out over a `spread` times larger part of the device memory
    bool create(size_t want, size_t chunk_bytes, int dev, bool shuffle, int spread = 1) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return false;
        chunk = (std::max(chunk_bytes, gran) + gran - 1) / gran * gran;
        const size_t n = (want + chunk - 1) / chunk;
        bytes = n * chunk;
        if (hipMemAddressReserve(&va, bytes, 0, nullptr, 0) != hipSuccess) {
            va = nullptr;
            return false;
        }
        handles.reserve(n);
        mapped.assign(n, 0);
        {
            std::vector<hipMemGenericAllocationHandle_t> spare;
            bool okc = true;
            for (size_t i = 0; i < n * (size_t) std::max(spread, 1) && okc; i++) {
                hipMemGenericAllocationHandle_t h;
                if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
                    okc = i >= n && handles.size() == n;   // out of memory while over-allocating: keep what there is if it suffices
                    if (!okc && handles.size() < n && !spare.empty()) {  // not enough kept ones: take spares
                        while (handles.size() < n && !spare.empty()) {
                            handles.push_back(spare.back());
                            spare.pop_back();
                        }
                        okc = handles.size() == n;
                    }
                    break;
                }
                if (handles.size() < n && i % (size_t) std::max(spread, 1) == 0) handles.push_back(h);
                else spare.push_back(h);
            }
            while (handles.size() < n && !spare.empty()) {
                handles.push_back(spare.back());
                spare.pop_back();
            }
            for (auto &h : spare) (void) hipMemRelease(h);
            if (handles.size() != n) {
                (void) hipGetLastError();
                release();
                return false;
            }
        }
        // physical chunk i (creation order: neighbours in physical memory more often than not) -> virtual slot perm[i]
        std::vector<size_t> perm(n);
        for (size_t i = 0; i < n; i++) perm[i] = i;
        if (shuffle) {
            uint64_t x = 0x9E3779B97F4A7C15ull;  // fixed seed: the layout of a given size is the same in every process
            for (size_t i = n; i > 1; i--) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                std::swap(perm[i - 1], perm[(size_t) (x % i)]);
            }
        }
        std::vector<hipMemGenericAllocationHandle_t> by_slot(n);
        for (size_t i = 0; i < n; i++) by_slot[perm[i]] = handles[i];
        handles = by_slot;
        for (size_t s = 0; s < n; s++) {
            if (hipMemMap((char *) va + s * chunk, chunk, 0, handles[s], 0) != hipSuccess) {
                release();
                return false;
            }
            mapped[s] = 1;
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, bytes, &acc, 1) != hipSuccess) {
            release();
            return false;
        }
        return true;
    }
};

struct acg_ldpc_code {
    Code c;
};

struct acg_ldpc_decoder {
    Code c;  // private copy: the handle outlives the code object safely
    acg_ldpc_params p;
    int device = 0;
    int cu_count = 256;
    hipStream_t stream = nullptr;
    bool ev_valid = false;
    std::mutex mu;
    std::string name;
    // BP
    BpLayout lay;
    BpTables tab{};
    std::vector<void *> dev_allocs;
    int maxd = 0, f64 = 0, L = 64;
    int block = 256, frames_per_block = 0;
    int grid_cap[2] = {0, 0};          // [mc] resident blocks: occupancy x CUs
    const void *kernel[2] = {nullptr, nullptr};
    size_t lds_block = 0;
    int variant = -1;       // wave-group kernels: 0 / 1 / 2 (see bp_inst_*.hip); -1 = workgroup-per-frame
    bool pair = false;      // ACG_LDPC_PREC_F16: two frames per workgroup, packed half-precision messages (bp_pair.hip)
    bool blk_idxlds = false, blk_idxreg = false;
    // layered min-sum (bp_layered.hip)
    bool layered = false;
    LayeredLayout llay;
    LayerTables ltab{};
    // streamed BP engine
    bool streamed = false;
    StreamTables stab{};
    const void *skernel = nullptr;
    const void *sring = nullptr;  // LDS-DMA ring variant (fp32), null = not available for this code
    int sring_per_cu = 2;
    bool sring_nt = false;  // ring instance with non-temporal slab accesses (slabs beyond the Infinity Cache)
    uint32_t *sws = nullptr;
    ScatteredAlloc sws_scattered;  // backing of sws when it was made of shuffled physical chunks (else sws is a hipMalloc)
    std::vector<float> sws_probe_ms;  // probe time of every workspace candidate that was tried (the fastest was kept)
    int sws_spread = 1;               // the kept physical chunks are every sws_spread-th of those created
    int sgrid = 0;
    // ADMM
    AdmmDevice *admm = nullptr;
    // staging for the host API
    void *st_y = nullptr;
    uint32_t *st_bits = nullptr;
    uint8_t *st_ok = nullptr;
    int32_t *st_iters = nullptr;
    int64_t st_frames = 0;
    HostPipe *pipe = nullptr;  // pipelined staging of the host-buffer entry points (created on first use)
    // MC through engines without an in-kernel generator (streamed): chunk buffers
    float *mc_y = nullptr;
    int64_t mc_frames = 0;
    // MC
    uint32_t *cw_dev = nullptr;
    int64_t cw_count = 0;
    uint64_t cw_hash = 0;
    unsigned long long *counters = nullptr;
    // Per-launch work counters: every launch takes the next slot of a small ring of device words (the dynamic frame /
    // tile hand-out of the kernels), so launches of one handle that overlap on different streams never share one.
    // ring_ev[k] is recorded behind the launch that used slot k; the next user of the slot — and, for the streamed
    // engine, whose HBM slabs belong to the handle, every launch on a different stream — waits on it on the device.
    // Timing: every launch also owns the (start, stop) event pair of its slot, so two launches of one handle in flight on
    // two streams never pair each other's events; ring_ev[k] IS the stop event of slot k.
    static constexpr int WORK_RING = 32;
    unsigned long long *work_ring = nullptr;
    hipEvent_t ring_ev0[WORK_RING] = {};
    hipEvent_t ring_ev[WORK_RING] = {};
    bool ring_used[WORK_RING] = {};
    uint64_t launch_seq = 0;
    int last_slot = -1;
    hipStream_t last_stream = nullptr;
};

extern "C" {

void acg_ldpc_params_default(acg_ldpc_params *p) {
    std::memset(p, 0, sizeof(*p));
    p->algo = ACG_LDPC_BP_SUMPRODUCT;
    p->max_iter = 50;
    p->alpha = 1.95;   // main.cpp:33
    p->mu = 0.5;
    p->eps_stop = 1e-5;  // qp_admm.h:182
    p->ms_scale = 1.0;
    p->early_exit = 1;
    p->precision = ACG_LDPC_PREC_DEFAULT;
    p->device = -1;
    p->lanes_per_frame = 0;
}

const char *acg_ldpc_last_error(void) { return g_err.c_str(); }

int acg_ldpc_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

// ---------------------------------------------------------------- code
static int acg_ldpc_code_from_dense_impl(const uint8_t *H, int32_t m, int32_t n, acg_ldpc_code **out) {
    if (!H || !out) {
        set_error("null argument");
        return 1;
    }
    auto *c = new acg_ldpc_code();
    if (!code_build(c->c, H, m, n)) {
        delete c;
        return 2;
    }
    *out = c;
    return 0;
}

int acg_ldpc_code_from_dense(const uint8_t *H, int32_t m, int32_t n, acg_ldpc_code **out) {
    return guarded([&] { return acg_ldpc_code_from_dense_impl(H, m, n, out); });
}

static int acg_ldpc_code_load_txt_impl(const char *path, acg_ldpc_code **out) {
    if (!path || !out) {
        set_error("null argument");
        return 1;
    }
    std::vector<uint8_t> H;
    int m = 0, n = 0;
    if (!code_read_txt(path, H, m, n)) return 2;
    return acg_ldpc_code_from_dense(H.data(), m, n, out);
}

int acg_ldpc_code_load_txt(const char *path, acg_ldpc_code **out) {
    return guarded([&] { return acg_ldpc_code_load_txt_impl(path, out); });
}

static int acg_ldpc_code_save_txt_impl(const acg_ldpc_code *code, const char *path) {
    if (!code || !path) {
        set_error("null argument");
        return 1;
    }
    return code_write_txt(code->c, path) ? 0 : 2;
}

int acg_ldpc_code_save_txt(const acg_ldpc_code *code, const char *path) {
    return guarded([&] { return acg_ldpc_code_save_txt_impl(code, path); });
}

void acg_ldpc_code_destroy(acg_ldpc_code *code) { delete code; }

void acg_ldpc_code_dims(const acg_ldpc_code *code, int32_t *m, int32_t *n, int32_t *E) {
    if (m) *m = code->c.m;
    if (n) *n = code->c.n;
    if (E) *E = code->c.E;
}

void acg_ldpc_code_dense(const acg_ldpc_code *code, uint8_t *H) {
    std::memcpy(H, code->c.H.data(), code->c.H.size());
}

void acg_ldpc_code_admm_shape(const acg_ldpc_code *code, int32_t *n_var, int32_t *n_con, int32_t *nnz, double *e_min,
                              double *e_max) {
    const AdmmLayout &a = code->c.admm;
    if (n_var) *n_var = a.n_var;
    if (n_con) *n_con = a.n_con;
    if (nnz) *nnz = a.nnz;
    if (e_min) *e_min = a.e_min;
    if (e_max) *e_max = a.e_max;
}

static int acg_ldpc_code_generator_impl(const acg_ldpc_code *code, uint8_t *G) {
    if (!code || !G) {
        set_error("null argument");
        return 2;
    }
    if (code->c.n <= code->c.m) {
        set_error("generator needs n > m");
        return 2;
    }
    return code_generator(code->c, G) ? 0 : 1;
}

int acg_ldpc_code_generator(const acg_ldpc_code *code, uint8_t *G) {
    return guarded([&] { return acg_ldpc_code_generator_impl(code, G); });
}

int acg_ldpc_code_is_codeword(const acg_ldpc_code *code, const uint8_t *bits) {
    return code_is_codeword(code->c, bits) ? 1 : 0;
}

// ---------------------------------------------------------------- decoder
static int decoder_setup_streamed(acg_ldpc_decoder *d) {
    const Code &c = d->c;
    if (std::max(c.max_cdeg, c.max_vdeg) > 16) {
        set_error("node degree above 16 is not supported by the streamed BP engine");
        return 3;
    }
    d->streamed = true;
    d->f64 = (d->p.precision == ACG_LDPC_PREC_F64) ? 1 : 0;
    d->L = 1;
    StreamTables &t = d->stab;
    int32_t *p32 = nullptr;
#define UP32S(vec, field)                                      \
    if (upload<int32_t>(vec, &p32)) return 10;                 \
    d->dev_allocs.push_back(p32);                              \
    t.field = p32;
    UP32S(c.row_ptr, row_ptr)
    UP32S(c.col_ptr, col_ptr)
    UP32S(c.col_edge, col_edge)
    t.m = c.m;
    t.n = c.n;
    t.E = c.E;
    t.nwords = (c.n + 31) / 32;
    const size_t ts = d->f64 ? 8 : 4;
    // LDS-DMA ring engine (fp32, node degrees that fit a ring slot): cut the sweeps into tasks
    size_t hb_bytes = (size_t) t.nwords * 64 * 4;  // HB[nwords][64] (u32)
    if (!d->f64 && c.max_cdeg <= RING_MAX_CDEG && c.max_vdeg <= RING_MAX_VDEG && getenv("ACG_STREAM_NO_RING") == nullptr) {
        RingTasks rt;
        ring_tasks_build(c, rt);
        const std::vector<int32_t> &hct = rt.ctask, &hvt = rt.vtask, &hvw = rt.vtask_of_word;
        struct { size_t n; size_t size() const { return n; } } ct{(size_t) rt.n_ctask}, vt{(size_t) rt.n_vtask};
        std::vector<int32_t> ce = c.col_edge;
        ce.resize(ce.size() + 4, 0);  // the gather reads its edge ids four at a time
        UP32S(hct, ctask)
        UP32S(hvt, vtask)
        UP32S(hvw, vtask_of_word)
        UP32S(ce, col_edge)
        t.n_ctask = (int32_t) ct.size();
        t.n_vtask = (int32_t) vt.size();
        // slabs of all resident workgroups beyond the 256 MiB Infinity Cache: stream them with non-temporal accesses
        const size_t slab_bytes = ((size_t) (c.E + c.n) * 64 * ts + (size_t) vt.size() * 64);
        bool nt = slab_bytes * 3 * (size_t) d->cu_count > ((size_t) 256 << 20);
        if (getenv("ACG_STREAM_NT")) nt = atoi(getenv("ACG_STREAM_NT")) != 0;  // developer A/B only
        d->sring_nt = nt;
        d->sring = bp_streamed_ring_ptr((d->p.algo == ACG_LDPC_BP_MINSUM) ? 1 : 0, nt);
        HIP_OK(hipFuncSetAttribute(d->sring, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
        hb_bytes = std::max(hb_bytes, (size_t) vt.size() * 64);  // ring engine: one byte per (variable task, frame)
    }
#undef UP32S
    // per workgroup: M[E][64] + LLR[n][64] (T) + hard decisions
    t.ws_words_per_wave = (int64_t) (((size_t) (c.E + c.n) * 64 * ts + hb_bytes + 255) / 256 * 64);
    if (const char *pad = getenv("ACG_STREAM_SLAB_PAD")) t.ws_words_per_wave += (int64_t) (atol(pad) / 256 * 64);  // developer A/B: slab stride + pad bytes
    d->block = 256;
    d->frames_per_block = 64;  // one 64-frame tile per workgroup at a time
    const int algo = (d->p.algo == ACG_LDPC_BP_MINSUM) ? 1 : 0;
    d->skernel = bp_streamed_ptr(algo, d->f64);
    // 2 workgroups per CU (x 4 wavefronts = 8 waves/CU keep > 1 MB of 256-byte lines in flight per CU);
    // each resident workgroup owns one slab: M[E][64] + LLR[n][64] + HB[nwords][64]
    d->sgrid = 2 * d->cu_count;
    if (d->sring) {  // ring engine: as many workgroups per CU as their rings fit in LDS
        int per_cu = 2;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, d->sring, RING_WAVES * 64, RING_LDS_BYTES) != hipSuccess) per_cu = 2;
        d->sring_per_cu = std::max(1, std::min(per_cu, ACG_RING_MAX_PER_CU));
        d->sgrid = std::max(d->sgrid, d->sring_per_cu * d->cu_count);
    }
    const size_t ws_bytes = (size_t) d->sgrid * (size_t) t.ws_words_per_wave * 4;
    if (t.ws_words_per_wave < (int64_t) (c.E + c.n) * 64) {  // the kernels index the slab without further checks
        set_error("internal: streamed-engine slab size not set");
        return 11;
    }
    {
        // Where the slabs live decides the rate of the ring kernel once they exceed the Infinity Cache (configs[4]: 768 slabs of
        // 10.4 MB; tools/stream_bimodal.py, profiles/r03_stream_bimodal.txt: one process, same kernel instance, same virtual
        // address, a new allocation per measurement, three GPU boxes):
        //   hipDeviceMallocContiguous             181 ms per launch, every time, whatever the slab stride: a physically contiguous
        //                                         backing is the slowest one
        //   hipMalloc                             158 or 181 ms, fixed for the life of the allocation, decided anew by every
        //                                         hipMalloc (round 2's "158 or 179 ms depending on the box"); one box gave 180 only
        //   physical chunks mapped into one       149-153 ms on one box (20 of 20), 152-180 ms (mostly 157-169) on another: never
        //   virtual range (hipMemCreate/hipMemMap) worse than hipMalloc, but still a draw per allocation
        // What the draw is: how COMPACT the physical backing is.  Keeping only every K-th of K times as many chunks (the others
        // are released at once) spaces the kept ones out over a K times larger part of the device memory, and on a box whose
        // plain candidates all probed slow (20.1 ms, 180 ms per launch) K = 4 gave 158 ms and K = 16 gave 148 ms — the fast
        // mode, every time since (a compact region keeps few DRAM banks in play for 768 concurrent streams; that reading fits
        // every observation above, the physical addresses themselves are not visible).  So a workspace beyond the Infinity Cache
        // is built from 64 MiB chunks spaced out 16-fold (ACG_STREAM_WS_SPREAD; costs ~3 s of decoder creation and, for a moment,
        // 16 x the workspace in device memory — when that is not available the chunks that were obtained are used as they are).
        // ACG_STREAM_WS_TRIES > 1 additionally times several candidates with a three-sweep launch of the decoder's own kernel
        // and keeps the fastest (the probe is always taken and reported: 17 ms = fast, 20 ms = slow on configs[4]).
        // hipMalloc remains the fallback and the small-workspace path.  ACG_STREAM_WS_ALLOC = 0 / 1 / 2 / 3 forces hipMalloc /
        // contiguous / shuffled chunks / chunks in creation order (developer A/B only).
        const char *wa = getenv("ACG_STREAM_WS_ALLOC");
        const int mode = wa ? atoi(wa) : (ws_bytes >= ((size_t) 64 << 20) ? 2 : 0);
        hipError_t e = hipErrorUnknown;
        if (mode == 1) e = hipExtMallocWithFlags((void **) &d->sws, ws_bytes, hipDeviceMallocContiguous);
        if (mode >= 2) {
            const char *cm = getenv("ACG_STREAM_WS_CHUNK_MB");
            const size_t chunk = (size_t) (cm ? atol(cm) : 64) << 20;
            const char *tr = getenv("ACG_STREAM_WS_TRIES");
            int tries = tr ? atoi(tr) : 1;
            // the placement only matters beyond the Infinity Cache, and the probe needs the ring kernel and room for its symbols
            const int64_t probe_frames = (int64_t) d->sgrid * 64;
            const bool big = ws_bytes > ((size_t) 512 << 20);
            const bool can_probe = d->sring && big && (size_t) probe_frames * c.n * 4 <= ws_bytes;
            if (!can_probe) tries = 1;
            tries = std::max(1, std::min(tries, 6));
            const char *sp = getenv("ACG_STREAM_WS_SPREAD");
            int spread = big ? std::max(1, std::min(sp ? atoi(sp) : 16, 64)) : 1;
            if (spread > 1) {   // never ask for more than ~80 % of the memory that is free right now
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && ws_bytes > 0)
                    spread = (int) std::max<size_t>(1, std::min<size_t>((size_t) spread, (size_t) ((double) free_b * 0.8 / (double) ws_bytes)));
            }
            std::vector<ScatteredAlloc> cand((size_t) tries);
            int best = -1;
            float best_ms = 0;
            d->sws_probe_ms.clear();
            d->sws_spread = spread;
            for (int k = 0; k < tries; k++) {
                if (!cand[k].create(ws_bytes, chunk, d->device, mode == 2, spread)) break;
                float ms = 0;
                if (can_probe) {
                    DecodeArgs pa{};
                    pa.y = cand[k].va;           // any readable memory will do as symbols: the probe times traffic, not decoding
                    pa.y_is_f64 = 0;
                    pa.frames = probe_frames;
                    pa.inv_var2 = 1.0;
                    pa.var = 1.0;
                    pa.max_iter = 3;
                    pa.early_exit = 0;
                    pa.ms_scale = 0.75f;
                    pa.work_counter = d->work_ring;
                    bool okp = true;
                    for (int rep = 0; rep < 2 && okp; rep++) {   // the second launch is the one timed
                        okp = hipMemsetAsync(d->work_ring, 0, sizeof(unsigned long long), d->stream) == hipSuccess &&
                              hipEventRecord(d->ring_ev0[0], d->stream) == hipSuccess &&
                              bp_streamed_ring_launch(d->sring, d->stab, pa, (uint32_t *) cand[k].va, d->sgrid, d->stream) == hipSuccess &&
                              hipEventRecord(d->ring_ev[0], d->stream) == hipSuccess && hipStreamSynchronize(d->stream) == hipSuccess;
                    }
                    if (!okp || hipEventElapsedTime(&ms, d->ring_ev0[0], d->ring_ev[0]) != hipSuccess) ms = 1e30f;
                    d->sws_probe_ms.push_back(ms);
                }
                if (best < 0 || ms < best_ms) {
                    best = k;
                    best_ms = ms;
                }
            }
            (void) hipGetLastError();
            for (int k = 0; k < tries; k++)
                if (k != best) cand[k].release();
            if (best >= 0) {
                d->sws_scattered = std::move(cand[best]);
                cand[best].va = nullptr;
                d->sws = (uint32_t *) d->sws_scattered.va;
                e = hipSuccess;
            }
        }
        if (e != hipSuccess) {
            (void) hipGetLastError();
            HIP_OK(hipMalloc((void **) &d->sws, ws_bytes));
        }
    }
    d->grid_cap[0] = d->grid_cap[1] = d->sgrid;
    return 0;
}

// schedule = LAYERED: min-sum over conflict-free layers of checks with in-place posteriors (bp_layered.hip)
static int decoder_setup_layered(acg_ldpc_decoder *d) {
    const Code &c = d->c;
    // (sum-product with the layered schedule is the reference's check rule, bp.h:49-57, in another message order: a different
    // algorithm from BeliefPropagationDecoder — FER-level parity only — that the caller has to ask for explicitly)
    if (d->p.engine == ACG_LDPC_ENGINE_STREAMED || d->p.precision == ACG_LDPC_PREC_F64) {
        set_error("the layered schedule runs on the LDS-resident engine with fp32 posteriors (messages fp32, or fp16 with ACG_LDPC_PREC_F16)");
        return 3;
    }
    const bool lay_f16 = d->p.precision == ACG_LDPC_PREC_F16;
    if (!bp_layered_build(c, d->llay)) return 3;
    const LayeredLayout &ll = d->llay;
    if (d->p.lanes_per_frame != 0 && d->p.lanes_per_frame != ll.G) {
        set_error("layered schedule: lanes_per_frame is chosen by the layering (pass 0)");
        return 3;
    }
    LayerTables &t = d->ltab;
    int32_t *p32 = nullptr;
    uint16_t *p16 = nullptr;
    if (upload<int32_t>(ll.layer, &p32)) return 10;
    d->dev_allocs.push_back(p32);
    t.layer = p32;
    if (ll.qc) {
        if (upload<int32_t>(ll.proto, &p32)) return 10;
        d->dev_allocs.push_back(p32);
        t.proto = p32;
        t.pos = nullptr;
        std::vector<int32_t> packed(ll.proto.size() / 2 + 8, 0);   // (+8: the kernel's scalar loads may run a few words ahead)
        for (size_t k = 0; k + 1 < ll.proto.size(); k += 2) packed[k / 2] = (int32_t) (((uint32_t) (ll.proto[k] * ll.Z * 4) << 16) | (uint32_t) (ll.proto[k + 1] * 4));
        if (upload<int32_t>(packed, &p32)) return 10;
        d->dev_allocs.push_back(p32);
        t.proto_packed = p32;
    } else {
        if (upload<uint16_t>(ll.pos, &p16)) return 10;
        d->dev_allocs.push_back(p16);
        t.pos = p16;
        t.proto = nullptr;
    }
    t.n_layers = ll.n_layers;
    t.Z = ll.Z;
    t.n = c.n;
    t.nwords = (c.n + 31) / 32;
    t.e_pad = ll.e_pad;
    t.p_words = (c.n + 1 + 3) & ~3;
    t.tab_lds_bytes = (int) (((size_t) ll.e_pad * 2 + 15) & ~(size_t) 15);
    // frame stride = G (mod 32) words: the lanes of the frames sharing a wavefront then fall into disjoint LDS banks
    t.r_words = lay_f16 ? (ll.e_pad + 1) / 2 : ll.e_pad;
    int words = t.p_words + t.r_words + t.nwords + 1;   // (+1: the Monte-Carlo instance's raw-channel error count)
    while (words % 32 != ll.G % 32) words++;
    t.lds_bytes_per_frame = words * 4;
    const int fpw = 64 / ll.G;
    const size_t per_wave = (size_t) t.lds_bytes_per_frame * fpw;
    // wavefronts per workgroup: the one that wastes the least LDS on the shared table while leaving >= 2 workgroups per CU
    int waves = 4;
    while (waves > 1 && (per_wave * waves + t.tab_lds_bytes) * 2 > 160 * 1024) waves >>= 1;
    {   // prefer the workgroup size that fits the most wavefronts per CU
        int best_w = waves, best_n = 0;
        for (int w : {4, 2, 1}) {
            const size_t blk = per_wave * w + t.tab_lds_bytes;
            if (blk > 160 * 1024) continue;
            const int nw = (int) ((160 * 1024) / blk) * w;
            if (nw > best_n) { best_n = nw; best_w = w; }
        }
        waves = best_w;
    }
    if (per_wave * waves + t.tab_lds_bytes > 160 * 1024) {
        set_error("layered schedule: a frame (posteriors + messages) does not fit in LDS");
        return 3;
    }
    d->layered = true;
    d->L = ll.G;
    d->f64 = 0;
    d->block = waves * 64;
    d->frames_per_block = waves * fpw;
    d->lds_block = per_wave * waves + t.tab_lds_bytes;
    // positions by arithmetic in the hot loop when the block columns' byte offsets fit the packed word (n * 4 < 65536 holds: n < 16000)
    // ACG_LAY_ARITH=1 (developer A/B): compute the positions of a quasi-cyclic H in the hot loop instead of reading the table
    const int lay_algo = d->p.algo == ACG_LDPC_BP_MINSUM ? 1 : 0;
    const bool qc_arith = ll.qc && ll.G == 20 && !lay_f16 && lay_algo == 1 && getenv("ACG_LAY_ARITH") != nullptr;
    for (int mc = 0; mc < 2; mc++) {
        const void *kp = bp_layered_kernel_ptr(ll.G, waves, qc_arith && !mc, lay_f16, lay_algo, mc != 0);
        if (!kp) {
            set_error("no layered kernel instance for this group width");
            return 3;
        }
        if (d->lds_block > 64 * 1024) HIP_OK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block));
        int occ = 0;
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, d->block, d->lds_block));
        if (occ < 1) occ = 1;
        d->kernel[mc] = kp;
        d->grid_cap[mc] = occ * d->cu_count;
    }
    d->tab.lds_bytes_per_frame = t.lds_bytes_per_frame;
    return 0;
}

static int decoder_setup_bp(acg_ldpc_decoder *d) {
    const Code &c = d->c;
    if (d->p.schedule == ACG_LDPC_SCHEDULE_LAYERED) return decoder_setup_layered(d);
    if (d->p.schedule != ACG_LDPC_SCHEDULE_FLOODING) {
        set_error("unknown schedule");
        return 1;
    }
    if (d->p.engine != ACG_LDPC_ENGINE_AUTO && d->p.engine != ACG_LDPC_ENGINE_FUSED &&
        d->p.engine != ACG_LDPC_ENGINE_STREAMED) {
        set_error("unknown engine");
        return 1;
    }
    const bool pair = (d->p.precision == ACG_LDPC_PREC_F16);
    if (pair) {
        if (d->p.algo != ACG_LDPC_BP_MINSUM || d->p.engine == ACG_LDPC_ENGINE_STREAMED) {
            set_error("ACG_LDPC_PREC_F16 exists for the fused min-sum decoder only (the reference's sum-product needs fp32/fp64 messages)");
            return 3;
        }
        if (c.max_cdeg > 8 || c.max_vdeg > 4 || c.n > 12 * 1024) {
            set_error("ACG_LDPC_PREC_F16 needs check degree <= 8, variable degree <= 4 and n <= 12288");
            return 3;
        }
    }
    if (d->p.engine == ACG_LDPC_ENGINE_STREAMED) return decoder_setup_streamed(d);
    d->maxd = std::max(c.max_cdeg, c.max_vdeg);
    if (!pair) {
        // does one frame fit in LDS?  (message words incl. padding at the smallest group size + LLRs)
        const size_t ts0 = (d->p.precision == ACG_LDPC_PREC_F64) ? 8 : 4;
        // (LLRs sit in registers for up to 12 passes of the group size, i.e. n <= 12288 in workgroup mode)
        const size_t approx = ((size_t) c.E + 64 + ((size_t) c.n > 12 * 1024 ? (size_t) c.n : 0)) * ts0;
        const bool fits = d->maxd <= 32 && approx <= 150 * 1024 && (size_t) c.E + 16 * (size_t) d->maxd < 60000;
        if (!fits) {
            if (d->p.engine == ACG_LDPC_ENGINE_FUSED) {
                set_error("code too large (or node degree > 32) for the fused LDS engine");
                return 3;
            }
            return decoder_setup_streamed(d);
        }
    }
    d->f64 = (d->p.precision == ACG_LDPC_PREC_F64) ? 1 : 0;
    int L = d->p.lanes_per_frame;
    if (L != 0 && L != 16 && L != 32 && L != 64 && L != 256 && L != 1024) {
        set_error("lanes_per_frame must be 0, 16, 32, 64 (wavefront groups) or 256, 1024 (one workgroup per frame)");
        return 3;
    }
    // Workgroup-per-frame mode (bp_block.hip) for codes whose message array leaves room for only a few
    // wavefront-sized frames per CU: the same LDS then feeds 4-16x as many wavefronts.
    bool blockmode = (L == 256 || L == 1024);
    if (pair) {  // always one workgroup per frame pair: 256 threads when the variables fit in 12 passes of them, else 1024
        if (L != 0 && L != 256 && L != 1024) {
            set_error("ACG_LDPC_PREC_F16: lanes_per_frame must be 0, 256 or 1024");
            return 3;
        }
        if (L == 0) L = (c.n <= 12 * 256) ? 256 : 1024;
        blockmode = true;
    }
    if (!pair && L == 0 && d->maxd <= 8) {
        const size_t ts0 = d->f64 ? 8 : 4;
        const size_t wave_frame = ((size_t) c.E + 64 + ((size_t) c.n > 12 * 64 ? (size_t) c.n : 0)) * ts0;  // rough, L = 64
        const int waves_cu = (int) std::min<size_t>(32, (160 * 1024) / std::max<size_t>(wave_frame, 1));
        if (waves_cu < 12) {
            // smallest workgroup that reaches >= 12 wavefronts per CU, else the largest
            const size_t blk_frame = ((size_t) c.E * 5 / 4 + 256) * ts0;
            L = (((160 * 1024) / std::max<size_t>(blk_frame, 1)) * 4 >= 12 && c.n <= 12 * 256) ? 256 : 1024;
            blockmode = (c.n <= 12 * L);
            if (!blockmode) L = 0;
        }
    }
    if (blockmode && (d->maxd > 8 || c.n > 12 * L)) {
        set_error("workgroup-per-frame BP needs node degree <= 8 and n <= 12 * lanes_per_frame");
        return 3;
    }
    if (L == 0) {
        // Auto: a pass costs its largest degree for all L lanes, so finer groups waste fewer padded
        // message slots (H05: 79% useful at L=64, 94% at L=32).  Measured on MI355X (H05, 50 it): fixed work
        // 19.0 M frames/s at L=32 vs 17.2 M at L=64; early exit 68.6 M vs 65.3 M at -2 dB, 296 M vs 285 M at +2 dB.
        BpLayout l64, l32;
        if (!bp_layout_build(c, 64, l64) || !bp_layout_build(c, 32, l32)) return 3;
        auto slots = [](const BpLayout &y) {
            long s = 0;
            for (int v : y.c_maxdeg) s += (long) v * y.L;
            for (int v : y.v_maxdeg) s += (long) v * y.L;
            return (double) s;
        };
        const double gain = slots(l64) / std::max(1.0, slots(l32));
        L = (gain > 1.05) ? 32 : 64;
    }
    d->L = L;
    if (!bp_layout_build(c, L, d->lay)) return 3;
    BpLayout &lay = d->lay;
    const int nwords = (c.n + 31) / 32;
    // the MC path stages n symbols in the message array before clearing it
    if (lay.a_words < ((c.n + 3) & ~3)) lay.a_words = (c.n + 3) & ~3;
    const size_t ts = d->f64 ? 8 : 4;
    // channel LLRs: in registers when there are at most 12 variable passes (degree <= 8 kernels), else in LDS
    // the fused kernels keep their LDS copy of the variable-side index table in BYTE offsets (16 bits): the message
    // array of a frame has to stay below 64 KiB for that copy (and the register-LLR instances, which require it)
    const bool a_fits16 = blockmode || (size_t) lay.a_words * ts <= 65535;
    const bool llr_regs = blockmode || ((d->maxd <= 8) && (lay.n_vpass <= 12) && a_fits16);
    const int llr_words = llr_regs ? 0 : lay.n_vpass * L;
    size_t per_frame = (size_t) (lay.a_words + llr_words) * ts + (size_t) nwords * 4;
    if (pair) per_frame = (size_t) lay.a_words * 4 + 2 * (size_t) nwords * 4;  // one 32-bit word per edge for TWO frames
    per_frame = (per_frame + 15) & ~(size_t) 15;

    BpTables &t = d->tab;
    int32_t *p32 = nullptr;
    uint16_t *p16 = nullptr;
    std::vector<int32_t> c_pass(2 * (size_t) lay.n_cpass), v_pass(2 * (size_t) lay.n_vpass);
    for (int p = 0; p < lay.n_cpass; p++) {
        c_pass[2 * p] = lay.c_maxdeg[p];
        c_pass[2 * p + 1] = lay.c_off[p];
    }
    for (int p = 0; p < lay.n_vpass; p++) {
        v_pass[2 * p] = lay.v_maxdeg[p];
        v_pass[2 * p + 1] = lay.v_idx_off[p];
    }
    std::vector<int32_t> c_cnt(34, 0), v_cnt(34, 0);
    for (size_t i = 0; i < lay.c_cnt_ge.size() && i < 34; i++) c_cnt[i] = lay.c_cnt_ge[i];
    for (size_t i = 0; i < lay.v_cnt_ge.size() && i < 34; i++) v_cnt[i] = lay.v_cnt_ge[i];
#define UP32(vec, field)                                       \
    if (upload<int32_t>(vec, &p32)) return 10;                 \
    d->dev_allocs.push_back(p32);                              \
    t.field = p32;
    UP32(c_pass, c_pass)
    UP32(c_cnt, c_cnt_ge)
    UP32(v_pass, v_pass)
    UP32(v_cnt, v_cnt_ge)
    UP32(lay.v_var, v_var)
#undef UP32
    if (upload<uint16_t>(lay.v_apos, &p16)) return 10;
    d->dev_allocs.push_back(p16);
    t.v_apos = p16;
    t.v_apos_len = lay.v_apos_len;
    // the variable-side index table is read by every wave in every iteration: keep a block-shared
    // copy in LDS unless it is large (then it is read through L1/L2)
    bool idxlds = a_fits16 && ((size_t) lay.v_apos_len * 2 <= 16 * 1024 || (llr_regs && !blockmode));
    if (blockmode) idxlds = (size_t) lay.v_apos_len * 2 <= 32 * 1024 &&
                            (((size_t) lay.v_apos_len * 2 + 15) & ~(size_t) 15) + per_frame <= 158 * 1024;
    t.idx_lds_bytes = idxlds ? (int) (((size_t) lay.v_apos_len * 2 + 15) & ~(size_t) 15) : 0;
    t.n_cpass = lay.n_cpass;
    t.n_vpass = lay.n_vpass;
    t.a_words = lay.a_words;
    t.zero_pos = lay.zero_pos;
    t.m = c.m;
    t.n = c.n;
    t.nwords = nwords;
    t.llr_words = llr_words;
    t.lds_bytes_per_frame = (int) per_frame;

    if (blockmode) {
        if (per_frame + t.idx_lds_bytes > 160 * 1024) {
            if (d->p.engine == ACG_LDPC_ENGINE_AUTO && d->p.lanes_per_frame == 0) {
                for (void *q : d->dev_allocs) (void) hipFree(q);
                d->dev_allocs.clear();
                return decoder_setup_streamed(d);
            }
            set_error("frame state does not fit in LDS (160 KiB per CU)");
            return 3;
        }
        d->block = L;
        d->frames_per_block = 1;
        d->lds_block = per_frame + t.idx_lds_bytes;
        if (pair) {
            d->pair = true;
            d->frames_per_block = 2;
            d->lds_block = per_frame;
            t.idx_lds_bytes = 0;
            // regular code (one check degree, one variable degree): the instance with fully unrolled passes
            bool regular = c.max_cdeg >= 1 && c.max_vdeg >= 1;
            for (int i = 0; i < c.m && regular; i++) regular = (c.row_ptr[i + 1] - c.row_ptr[i] == c.max_cdeg);
            for (int j = 0; j < c.n && regular; j++) regular = (c.col_ptr[j + 1] - c.col_ptr[j] == c.max_vdeg);
            const void *kp = bp_pair_kernel_ptr(L, regular);
            if (!kp) {
                set_error("no paired-frame kernel instance for this configuration");
                return 3;
            }
            if (d->lds_block > 64 * 1024)
                HIP_OK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block));
            int occ = 0;
            HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, d->block, d->lds_block));
            d->kernel[0] = kp;
            d->kernel[1] = nullptr;  // Monte-Carlo runs go AWGN kernel -> decode -> classify kernel
            d->grid_cap[0] = d->grid_cap[1] = std::max(occ, 1) * d->cu_count;
            return 0;
        }
        const int algo_b = (d->p.algo == ACG_LDPC_BP_MINSUM) ? 1 : 0;
        for (int mc = 0; mc < 2; mc++) {
            // index table too large for LDS and variable degree <= 4: keep it in registers (decode kernel only)
            const bool idxreg = !idxlds && c.max_vdeg <= 4 && lay.n_vpass <= 12 && getenv("ACG_BP_NO_IDXREG") == nullptr;
            // regular code (one check degree <= 8, one variable degree <= 4): the instance without the paths for anything else
            bool regular_b = idxreg && c.max_cdeg >= 1 && c.max_cdeg <= 8 && c.max_vdeg >= 1 && c.max_vdeg <= 4 && getenv("ACG_BP_NO_REGULAR") == nullptr;
            for (int i = 0; i < c.m && regular_b; i++) regular_b = (c.row_ptr[i + 1] - c.row_ptr[i] == c.max_cdeg);
            for (int j = 0; j < c.n && regular_b; j++) regular_b = (c.col_ptr[j + 1] - c.col_ptr[j] == c.max_vdeg);
            const void *kp = bp_block_kernel_ptr(algo_b, d->f64, L, mc != 0, idxlds, idxreg, regular_b);
            if (mc == 0) {
                d->blk_idxlds = idxlds;
                d->blk_idxreg = idxreg;
            }
            if (!kp) {
                set_error("no workgroup-per-frame kernel instance for this configuration");
                return 3;
            }
            if (d->lds_block > 64 * 1024)
                HIP_OK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block));
            int occ = 0;
            HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, d->block, d->lds_block));
            if (occ < 1) occ = 1;
            d->kernel[mc] = kp;
            d->grid_cap[mc] = occ * d->cu_count;
        }
        return 0;
    }
    const int fpw = 64 / L;
    // waves per block: as many as fit in half the LDS (so at least two blocks share a CU), at most 4
    int waves = 4;
    while (waves > 1 && per_frame * fpw * waves + t.idx_lds_bytes > 160 * 1024 / 2) waves >>= 1;
    if (per_frame * fpw * waves + t.idx_lds_bytes > 160 * 1024) {
        if (d->p.engine == ACG_LDPC_ENGINE_AUTO) {
            for (void *q : d->dev_allocs) (void) hipFree(q);
            d->dev_allocs.clear();
            return decoder_setup_streamed(d);
        }
        set_error("frame state does not fit in LDS (160 KiB per CU)");
        return 3;
    }
    d->block = waves * 64;
    d->frames_per_block = waves * fpw;
    d->lds_block = per_frame * fpw * waves + t.idx_lds_bytes;
    const int algo = (d->p.algo == ACG_LDPC_BP_MINSUM) ? 1 : 0;
    for (int mc = 0; mc < 2; mc++) {
        const int variant = idxlds ? ((llr_regs) ? 2 : 1) : 0;
        d->variant = variant;
        const void *kp = bp_kernel_ptr(algo, d->f64, d->maxd, L, mc != 0, variant);
        if (!kp) {
            set_error("no kernel instance for this configuration");
            return 3;
        }
        if (d->lds_block > 64 * 1024)
            HIP_OK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block));
        int occ = 0;
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, d->block, d->lds_block));
        if (occ < 1) occ = 1;
        d->kernel[mc] = kp;
        d->grid_cap[mc] = occ * d->cu_count;
    }
    return 0;
}

static int acg_ldpc_decoder_create_impl(const acg_ldpc_code *code, const acg_ldpc_params *params, acg_ldpc_decoder **out) {
    if (!code || !params || !out) {
        set_error("null argument");
        return 1;
    }
    if (params->max_iter < 0) {
        set_error("max_iter must be >= 0");
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: libacg_ldpc_hip has no CPU fallback");
        return 20;
    }
    // owned until handed out: an exception or an error below releases the streams, events and device memory made so far
    struct Drop { void operator()(acg_ldpc_decoder *x) const { acg_ldpc_decoder_destroy(x); } };
    std::unique_ptr<acg_ldpc_decoder, Drop> own(new acg_ldpc_decoder());
    acg_ldpc_decoder *d = own.get();
    d->c = code->c;
    d->p = *params;
    int dev = params->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        set_error("device ordinal out of range");
        return 1;
    }
    d->device = dev;
    int rc = 0;
    do {
        if (hipSetDevice(dev) != hipSuccess) { set_error("hipSetDevice failed"); rc = 10; break; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error("hipGetDeviceProperties failed"); rc = 10; break; }
        d->cu_count = prop.multiProcessorCount;
        if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); rc = 10; break; }
        if (hipMalloc((void **) &d->counters, sizeof(unsigned long long) * MC_NCOUNTERS) != hipSuccess) { set_error("hipMalloc failed"); rc = 10; break; }
        if (hipMalloc((void **) &d->work_ring, sizeof(unsigned long long) * acg_ldpc_decoder::WORK_RING) != hipSuccess) { set_error("hipMalloc failed"); rc = 10; break; }
        {
            bool evok = true;
            for (int k = 0; k < acg_ldpc_decoder::WORK_RING; k++)
                evok = evok && hipEventCreate(&d->ring_ev0[k]) == hipSuccess && hipEventCreate(&d->ring_ev[k]) == hipSuccess;
            if (!evok) { set_error("hipEventCreate failed"); rc = 10; break; }
        }
        if (params->algo == ACG_LDPC_QPADMM) {
            d->name = "QP-ADMM";  // qp_admm.h:189
            std::string err;
            d->admm = admm_device_create(d->c, d->p, d->cu_count, err);
            if (!d->admm) { set_error(err); rc = 3; break; }
        } else if (params->algo == ACG_LDPC_BP_SUMPRODUCT || params->algo == ACG_LDPC_BP_MINSUM) {
            d->name = params->algo == ACG_LDPC_BP_SUMPRODUCT ? "BP" : "MS";  // bp.h:218
            rc = decoder_setup_bp(d);
        } else {
            set_error("unknown algo");
            rc = 1;
        }
    } while (0);
    if (rc) return rc;
    *out = own.release();
    return 0;
}

int acg_ldpc_decoder_create(const acg_ldpc_code *code, const acg_ldpc_params *params, acg_ldpc_decoder **out) {
    return guarded([&] { return acg_ldpc_decoder_create_impl(code, params, out); });
}

void acg_ldpc_decoder_destroy(acg_ldpc_decoder *d) {
    if (!d) return;
    (void) hipSetDevice(d->device);
    if (d->stream) (void) hipStreamSynchronize(d->stream);
    // launches of this handle that are still in flight on CALLER streams (acg_ldpc_decode_batch_dev is asynchronous): every
    // launch recorded the stop event of its ring slot on its own stream — wait for them before the tables go (a handle can be
    // destroyed by the LRU of a host mirror while the caller's stream is still busy)
    for (int k = 0; k < acg_ldpc_decoder::WORK_RING; k++)
        if (d->ring_used[k] && d->ring_ev[k]) (void) hipEventSynchronize(d->ring_ev[k]);
    for (void *p : d->dev_allocs) (void) hipFree(p);
    if (d->admm) admm_device_destroy(d->admm);
    if (d->sws_scattered.va) d->sws_scattered.release();
    else if (d->sws) (void) hipFree(d->sws);
    if (d->mc_y) (void) hipFree(d->mc_y);
    if (d->st_y) (void) hipFree(d->st_y);
    if (d->st_bits) (void) hipFree(d->st_bits);
    if (d->st_ok) (void) hipFree(d->st_ok);
    if (d->st_iters) (void) hipFree(d->st_iters);
    delete d->pipe;
    if (d->cw_dev) (void) hipFree(d->cw_dev);
    if (d->counters) (void) hipFree(d->counters);
    if (d->work_ring) (void) hipFree(d->work_ring);
    for (int k = 0; k < acg_ldpc_decoder::WORK_RING; k++) {
        if (d->ring_ev0[k]) (void) hipEventDestroy(d->ring_ev0[k]);
        if (d->ring_ev[k]) (void) hipEventDestroy(d->ring_ev[k]);
    }
    if (d->stream) (void) hipStreamDestroy(d->stream);
    delete d;
}

const char *acg_ldpc_decoder_name(const acg_ldpc_decoder *d) { return d->name.c_str(); }

void acg_ldpc_decoder_layout(const acg_ldpc_decoder *d, int32_t *lds_bytes_per_frame, int32_t *lanes_per_frame,
                             int32_t *frames_per_block, int32_t *grid_blocks) {
    if (d->admm) {
        admm_device_layout(d->admm, lds_bytes_per_frame, lanes_per_frame, frames_per_block, grid_blocks);
        return;
    }
    if (lds_bytes_per_frame) *lds_bytes_per_frame = d->streamed ? 0 : d->tab.lds_bytes_per_frame;
    if (lanes_per_frame) *lanes_per_frame = d->L;
    if (frames_per_block) *frames_per_block = d->frames_per_block;
    if (grid_blocks) *grid_blocks = d->grid_cap[0];
}

// one line naming what this handle launches (diagnostics: bench.py records it next to every timed leg)
static std::string describe(const acg_ldpc_decoder *d) {
    char b[512];
    const char *algo = d->p.algo == ACG_LDPC_QPADMM ? "qpadmm" : (d->p.algo == ACG_LDPC_BP_MINSUM ? "minsum" : "sum-product");
    if (d->admm) {
        int lds = 0, L = 0, fpb = 0, grid = 0;
        admm_device_layout(d->admm, &lds, &L, &fpb, &grid);
        snprintf(b, sizeof b, "%s engine=lds lanes_per_frame=%d frames_per_block=%d lds_bytes_per_frame=%d grid_cap=%d", algo, L, fpb, lds, grid);
    } else if (d->streamed) {
        const size_t slab = (size_t) d->stab.ws_words_per_wave * 4;
        snprintf(b, sizeof b, "%s engine=streamed kernel=%s%s f64=%d slab_bytes=%zu slabs=%d workspace_bytes=%zu workspace_base=%p "
                               "workgroups_per_cu=%d schedule=%s",
                 algo, d->sring ? "bp_streamed_ring_kernel" : "bp_streamed_kernel", d->sring ? (d->sring_nt ? "<NT>" : "<default-policy>") : "",
                 d->f64, slab, d->sgrid, slab * (size_t) d->sgrid, (void *) d->sws, d->sring ? d->sring_per_cu : 2,
                 d->sws_scattered.va ? "flooding workspace=mapped-chunks" : "flooding workspace=hipMalloc");
        if (!d->sws_probe_ms.empty()) {
            std::string t = b;
            t += " workspace_spread=" + std::to_string(d->sws_spread) + " workspace_probe_ms=";
            for (size_t i = 0; i < d->sws_probe_ms.size(); i++) {
                char q[32];
                snprintf(q, sizeof q, "%s%.2f", i ? "/" : "", d->sws_probe_ms[i]);
                t += q;
            }
            return t;
        }
    } else {
        snprintf(b, sizeof b, "%s engine=fused kernel=%s lanes_per_frame=%d f64=%d block=%d frames_per_block=%d lds_block=%zu grid_cap=%d "
                               "idx_lds=%d idx_reg=%d schedule=%s",
                 algo, d->layered ? "bp_layered_kernel" : (d->pair ? "bp_pair_kernel" : (d->variant < 0 ? "bp_block_kernel" : "bp_fused_kernel")), d->L, d->f64, d->block,
                 d->frames_per_block, d->lds_block, d->grid_cap[0], d->variant < 0 ? (int) d->blk_idxlds : (d->variant > 0), (int) d->blk_idxreg,
                 d->p.schedule == ACG_LDPC_SCHEDULE_LAYERED ? (d->p.precision == ACG_LDPC_PREC_F16 ? "layered messages=fp16" : "layered messages=fp32")
                                                            : "flooding");
    }
    return b;
}

int32_t acg_ldpc_decoder_describe(const acg_ldpc_decoder *d, char *buf, int32_t cap) {
    if (!d) return 0;
    const std::string s = describe(d);
    if (buf && cap > 0) {
        const size_t k = std::min<size_t>(s.size(), (size_t) cap - 1);
        std::memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return (int32_t) s.size() + 1;
}

static void fill_channel(DecodeArgs &a, double snr) {
    const double var = std::pow(10, -(snr / 10)) / 2;  // llr_variance, channel.h:12
    a.var = var;
    a.inv_var2 = 2.0 / var;
    a.sigma = (float) std::sqrt(var);
}

// launch on stream s (events recorded around the kernel on that stream).  Caller holds d->mu.
static int launch_decode(acg_ldpc_decoder *d, DecodeArgs &a, hipStream_t s) {
    a.max_iter = d->p.max_iter;
    a.early_exit = d->p.early_exit;
    a.ms_scale = (float) d->p.ms_scale;
    if (a.frames <= 0) return 0;
    // this launch's own work counter (see acg_ldpc_decoder::work_ring)
    const int slot = (int) (d->launch_seq++ % acg_ldpc_decoder::WORK_RING);
    if (d->ring_used[slot]) HIP_OK(hipStreamWaitEvent(s, d->ring_ev[slot], 0));
    if (d->streamed && d->last_slot >= 0 && d->last_stream != s) HIP_OK(hipStreamWaitEvent(s, d->ring_ev[d->last_slot], 0));
    a.work_counter = d->work_ring + slot;
    HIP_OK(hipMemsetAsync(a.work_counter, 0, sizeof(unsigned long long), s));
#ifdef ACG_BLOCK_STAMPS
    static unsigned long long *stamp_buf = nullptr;  // developer build only (tools/ab_variant.sh ... -DACG_BLOCK_STAMPS)
    if (!stamp_buf) HIP_OK(hipMalloc((void **) &stamp_buf, 16 * 5 * sizeof(unsigned long long)));
    HIP_OK(hipMemsetAsync(stamp_buf, 0, 16 * 5 * sizeof(unsigned long long), s));
    a.dbg_post = stamp_buf;
#endif
    HIP_OK(hipEventRecord(d->ring_ev0[slot], s));
    if (d->admm) {
        std::string err;
        hipError_t e = admm_launch(d->admm, a, s, err);
        if (e != hipSuccess) {
            set_error(err.empty() ? std::string("admm launch: ") + hipGetErrorString(e) : err);
            return 10;
        }
    } else if (d->streamed) {
        if (a.mc) {
            set_error("internal: streamed engine has no in-kernel generator");
            return 11;
        }
        // W wavefronts cooperate on a tile: 4 when there are enough tiles to fill the chip, more for small batches
        const int64_t tiles = (a.frames + 63) / 64;
        int W = 4;
        while (W < 8 && tiles * W < 8 * (int64_t) d->cu_count) W <<= 1;
        const int per_cu = (W <= 4) ? 2 : 1;
        int grid = (int) std::min<int64_t>(tiles, (int64_t) per_cu * d->cu_count);
        if (d->sring) {
            // traces (acg_ldpc_debug_bp_trace) run the debug instance of the SAME kernel: its sweeps, its counted waits
            const void *kp = d->sring;
            if (a.dbg_c2v || a.dbg_v2c) {
                kp = bp_streamed_ring_ptr_dbg();
                HIP_OK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
            }
            grid = (int) std::min<int64_t>(tiles, (int64_t) d->sring_per_cu * d->cu_count);
            HIP_OK(bp_streamed_ring_launch(kp, d->stab, a, d->sws, grid, s));
        } else {
            HIP_OK(bp_streamed_launch(d->skernel, d->stab, a, d->sws, grid, W * 64, s));
        }
    } else if (d->layered) {
        const int mc = a.mc ? 1 : 0;
        const int64_t blocks = (a.frames + d->frames_per_block - 1) / d->frames_per_block;
        const int grid = (int) std::min<int64_t>(blocks, d->grid_cap[mc]);
        HIP_OK(bp_layered_launch(d->kernel[mc], d->ltab, a, grid, d->block, d->lds_block, s));
    } else {
        int64_t blocks = (a.frames + d->frames_per_block - 1) / d->frames_per_block;
        const int mc = a.mc ? 1 : 0;
        int grid = (int) std::min<int64_t>(blocks, d->grid_cap[mc]);
        HIP_OK(bp_launch(d->kernel[mc], d->tab, a, grid, d->block, d->lds_block, s));
    }
    HIP_OK(hipEventRecord(d->ring_ev[slot], s));   // stop event of this launch = the event later users of the slot wait on
#ifdef ACG_BLOCK_STAMPS
    if (!d->admm && !d->streamed && getenv("ACG_STAMPS")) {
        unsigned long long h[16 * 5];
        HIP_OK(hipStreamSynchronize(s));
        HIP_OK(hipMemcpy(h, stamp_buf, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 16; w++)
            if (h[w * 5 + 4])
                fprintf(stderr, "[stamps] wave %2d: per sweep: check %6.0f  barrier %6.0f  var %6.0f  barrier %6.0f cycles (%llu sweeps)\n", w,
                        (double) h[w * 5] / h[w * 5 + 4], (double) h[w * 5 + 1] / h[w * 5 + 4], (double) h[w * 5 + 2] / h[w * 5 + 4],
                        (double) h[w * 5 + 3] / h[w * 5 + 4], h[w * 5 + 4]);
    }
#endif
    d->ring_used[slot] = true;
    d->last_slot = slot;
    d->last_stream = s;
    d->ev_valid = true;
    return 0;
}

static int acg_ldpc_decode_batch_dev_impl(acg_ldpc_decoder *d, const void *y_dev, int32_t y_is_f64, int64_t frames, double snr,
                              uint32_t *bits_dev, uint8_t *ok_dev, int32_t *iters_dev, void *stream) {
    if (!d) {
        set_error("null decoder");
        return 1;
    }
    if (frames < 0 || (frames > 0 && !y_dev)) {
        set_error("bad frames / y");
        return 1;
    }
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_OK(hipSetDevice(d->device));
    DecodeArgs a{};
    a.y = y_dev;
    a.y_is_f64 = y_is_f64;
    a.frames = frames;
    fill_channel(a, snr);
    a.out_bits = bits_dev;
    a.out_ok = ok_dev;
    a.out_iters = iters_dev;
    a.mc = 0;
    return launch_decode(d, a, stream ? (hipStream_t) stream : d->stream);
}

int acg_ldpc_decode_batch_dev(acg_ldpc_decoder *d, const void *y_dev, int32_t y_is_f64, int64_t frames, double snr,
                              uint32_t *bits_dev, uint8_t *ok_dev, int32_t *iters_dev, void *stream) {
    return guarded([&] { return acg_ldpc_decode_batch_dev_impl(d, y_dev, y_is_f64, frames, snr, bits_dev, ok_dev, iters_dev, stream); });
}

static int ensure_staging(acg_ldpc_decoder *d, int64_t frames) {
    if (frames <= d->st_frames) return 0;
    if (d->st_y) (void) hipFree(d->st_y);
    if (d->st_bits) (void) hipFree(d->st_bits);
    if (d->st_ok) (void) hipFree(d->st_ok);
    if (d->st_iters) (void) hipFree(d->st_iters);
    d->st_y = nullptr;
    d->st_bits = nullptr;
    d->st_ok = nullptr;
    d->st_iters = nullptr;
    d->st_frames = 0;
    const int nwords = (d->c.n + 31) / 32;
    HIP_OK(hipMalloc(&d->st_y, (size_t) frames * d->c.n * sizeof(double)));
    HIP_OK(hipMalloc((void **) &d->st_bits, (size_t) frames * nwords * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **) &d->st_ok, (size_t) frames));
    HIP_OK(hipMalloc((void **) &d->st_iters, (size_t) frames * sizeof(int32_t)));
    d->st_frames = frames;
    return 0;
}

static int ensure_pipe(acg_ldpc_decoder *d, int64_t chunk, size_t y_bytes) {
    if (!d->pipe) {
        // built completely before it is published in the handle: a half-made pipe must never be seen by a later call
        std::unique_ptr<HostPipe> np(new HostPipe());
        for (int b = 0; b < HostPipe::NBUF; b++) {
            HIP_OK(hipStreamCreateWithFlags(&np->stream[b], hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&np->done[b], hipEventDisableTiming));
        }
        d->pipe = np.release();
    }
    HostPipe &P = *d->pipe;
    // keep the buffers while they fit and are not grossly oversized for what is asked now (a 1M-frame batch followed by
    // single-frame decode() calls must not pin hundreds of MB for good)
    const size_t want = (size_t) chunk * y_bytes, have = (size_t) P.chunk * P.y_bytes;
    if (chunk <= P.chunk && y_bytes <= P.y_bytes && (have <= ((size_t) 32 << 20) || have <= 16 * want)) return 0;
    P.release();
    const int nwords = (d->c.n + 31) / 32;
    for (int b = 0; b < HostPipe::NBUF; b++) {
        HIP_OK(hipHostMalloc(&P.pin_y[b], (size_t) chunk * y_bytes, hipHostMallocDefault));
        HIP_OK(hipHostMalloc((void **) &P.pin_out[b], (size_t) chunk * (nwords * 4 + 5), hipHostMallocDefault));
        HIP_OK(hipMalloc(&P.dev_y[b], (size_t) chunk * y_bytes));
        HIP_OK(hipMalloc((void **) &P.dev_out[b], (size_t) chunk * (nwords * 4 + 5)));
    }
    P.chunk = chunk;
    P.y_bytes = y_bytes;
    return 0;
}

// frames per chunk of the pipelined host path: bounded by BYTES (256 MiB of symbols per staging buffer), not by a frame
// count — 65536 frames of the 10 000-symbol code in doubles would pin 2 x 5.2 GB of host memory and as much HBM per handle
static int64_t host_chunk_frames(int64_t frames, size_t y_bytes) {
    const int64_t by_bytes = (int64_t) (((size_t) 256 << 20) / std::max<size_t>(y_bytes, 1));
    const int64_t cap = std::max<int64_t>(1024, std::min<int64_t>(1 << 16, by_bytes));
    return std::min<int64_t>(frames, cap);
}

// packed words -> one byte per bit, 8 bits at a time through a 256-entry table
static void unpack_bits(const uint32_t *words, int nwords, int n, int64_t frames, uint8_t *bits) {
    static const std::vector<uint64_t> lut = [] {
        std::vector<uint64_t> t(256);
        for (int x = 0; x < 256; x++) {
            uint64_t v = 0;
            for (int k = 0; k < 8; k++) v |= (uint64_t) ((x >> k) & 1) << (8 * k);
            t[x] = v;
        }
        return t;
    }();
    for (int64_t f = 0; f < frames; f++) {
        uint8_t *b = bits + (size_t) f * n;
        const uint8_t *w = reinterpret_cast<const uint8_t *>(words + (size_t) f * nwords);
        int v = 0;
        for (; v + 8 <= n; v += 8) std::memcpy(b + v, &lut[w[v >> 3]], 8);
        for (; v < n; v++) b[v] = (w[v >> 3] >> (v & 7)) & 1u;
    }
}

// Host buffers in, host buffers out: chunks of the batch travel through two pinned staging sets.  Per chunk c (set c % 2):
// host threads copy the symbols into pinned memory -> H2D, decode, D2H of words / flags / sweep counts on the set's
// stream -> host threads expand the words into one byte per bit.  Chunk c + 1 is packed and chunk c - 1 unpacked while
// the GPU works on chunk c.  elem = 8 (double symbols: exact LLRs, channel.h:14-16) or 4 (float symbols).
static int decode_batch_host(acg_ldpc_decoder *d, const void *y, int elem, int64_t frames, double snr, uint8_t *bits, uint8_t *ok,
                             int32_t *iters) {
    const int n = d->c.n, nwords = (n + 31) / 32;
    const size_t y_bytes = (size_t) n * elem;
    // small batches (single frames: the reference's decode()) take one chunk; large ones <= 64k frames / 256 MiB per chunk
    const int64_t chunk = host_chunk_frames(frames, y_bytes);
    if (int rc = ensure_pipe(d, chunk, y_bytes)) return rc;
    HostPipe &P = *d->pipe;
    const int64_t nchunks = (frames + chunk - 1) / chunk;
    const bool threads = frames >= 4096;  // tiny batches: the hand-off to the pool costs more than the copy
    HostPool *pool = threads ? host_pool() : nullptr;   // process-wide, created on first use
    auto chunk_frames = [&](int64_t c) { return std::min(chunk, frames - c * chunk); };
    auto pack = [&](int64_t c) {
        const int b = (int) (c % HostPipe::NBUF);
        const int64_t fc = chunk_frames(c);
        const unsigned char *src = reinterpret_cast<const unsigned char *>(y) + (size_t) c * chunk * y_bytes;
        unsigned char *dst = reinterpret_cast<unsigned char *>(P.pin_y[b]);
        const size_t total = (size_t) fc * y_bytes;
        if (!threads) {
            std::memcpy(dst, src, total);
            return;
        }
        pool->run_all([&](int part, int parts) {
            const size_t lo = total * part / parts / 64 * 64, hi = (part + 1 == parts) ? total : total * (part + 1) / parts / 64 * 64;
            std::memcpy(dst + lo, src + lo, hi - lo);
        });
    };
    auto submit = [&](int64_t c) -> int {
        const int b = (int) (c % HostPipe::NBUF);
        const int64_t fc = chunk_frames(c);
        hipStream_t s = P.stream[b];
        HIP_OK(hipMemcpyAsync(P.dev_y[b], P.pin_y[b], (size_t) fc * y_bytes, hipMemcpyHostToDevice, s));
        DecodeArgs a{};
        a.y = P.dev_y[b];
        a.y_is_f64 = (elem == 8) ? 1 : 0;
        a.frames = fc;
        fill_channel(a, snr);
        a.out_bits = reinterpret_cast<uint32_t *>(P.dev_out[b]);
        a.out_iters = reinterpret_cast<int32_t *>(P.dev_out[b] + (size_t) fc * nwords * 4);
        a.out_ok = P.dev_out[b] + (size_t) fc * (nwords * 4 + 4);
        if (int rc = launch_decode(d, a, s)) return rc;
        HIP_OK(hipMemcpyAsync(P.pin_out[b], P.dev_out[b], (size_t) fc * (nwords * 4 + 5), hipMemcpyDeviceToHost, s));
        HIP_OK(hipEventRecord(P.done[b], s));
        return 0;
    };
    auto collect = [&](int64_t c) -> int {
        const int b = (int) (c % HostPipe::NBUF);
        const int64_t fc = chunk_frames(c), f0 = c * chunk;
        HIP_OK(hipEventSynchronize(P.done[b]));
        const uint32_t *pbits = reinterpret_cast<const uint32_t *>(P.pin_out[b]);
        std::memcpy(ok + f0, P.pin_out[b] + (size_t) fc * (nwords * 4 + 4), (size_t) fc);
        if (iters) std::memcpy(iters + f0, P.pin_out[b] + (size_t) fc * nwords * 4, (size_t) fc * 4);
        if (!threads) {
            unpack_bits(pbits, nwords, n, fc, bits + (size_t) f0 * n);
            return 0;
        }
        pool->run_all([&](int part, int parts) {
            const int64_t lo = fc * part / parts, hi = fc * (part + 1) / parts;
            unpack_bits(pbits + (size_t) lo * nwords, nwords, n, hi - lo, bits + (size_t) (f0 + lo) * n);
        });
        return 0;
    };
    pack(0);
    for (int64_t c = 0; c < nchunks; c++) {
        if (int rc = submit(c)) return rc;
        if (c + 1 < nchunks) {
            // set (c + 1) % 2 was last used by chunk c - 1: its results must be out before its buffers are refilled
            if (c >= 1)
                if (int rc = collect(c - 1)) return rc;
            pack(c + 1);
        } else if (c >= 1) {
            if (int rc = collect(c - 1)) return rc;
        }
    }
    return collect(nchunks - 1);
}

static int acg_ldpc_decode_batch_impl(acg_ldpc_decoder *d, const double *y, int64_t frames, double snr, uint8_t *bits, uint8_t *ok,
                          int32_t *iters) {
    if (!d) {
        set_error("null decoder");
        return 1;
    }
    if (frames < 0 || (frames > 0 && (!y || !bits || !ok))) {
        set_error("null buffer");
        return 1;
    }
    if (frames == 0) return 0;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_OK(hipSetDevice(d->device));
    return decode_batch_host(d, y, 8, frames, snr, bits, ok, iters);
}

int acg_ldpc_decode_batch(acg_ldpc_decoder *d, const double *y, int64_t frames, double snr, uint8_t *bits, uint8_t *ok,
                          int32_t *iters) {
    return guarded([&] { return acg_ldpc_decode_batch_impl(d, y, frames, snr, bits, ok, iters); });
}

static int acg_ldpc_decode_batch_f32_impl(acg_ldpc_decoder *d, const float *y, int64_t frames, double snr, uint8_t *bits, uint8_t *ok,
                              int32_t *iters) {
    if (!d) {
        set_error("null decoder");
        return 1;
    }
    if (frames < 0 || (frames > 0 && (!y || !bits || !ok))) {
        set_error("null buffer");
        return 1;
    }
    if (frames == 0) return 0;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_OK(hipSetDevice(d->device));
    return decode_batch_host(d, y, 4, frames, snr, bits, ok, iters);
}

int acg_ldpc_decode_batch_f32(acg_ldpc_decoder *d, const float *y, int64_t frames, double snr, uint8_t *bits, uint8_t *ok,
                              int32_t *iters) {
    return guarded([&] { return acg_ldpc_decode_batch_f32_impl(d, y, frames, snr, bits, ok, iters); });
}

int acg_ldpc_decoder_sync(acg_ldpc_decoder *d) {
    if (!d) return 1;
    HIP_OK(hipSetDevice(d->device));
    HIP_OK(hipStreamSynchronize(d->stream));
    return 0;
}

float acg_ldpc_decoder_last_kernel_ms(acg_ldpc_decoder *d) {
    if (!d) return -1.0f;
    int slot;
    {
        std::lock_guard<std::mutex> lk(d->mu);
        if (!d->ev_valid || d->last_slot < 0) return -1.0f;
        slot = d->last_slot;
    }
    (void) hipSetDevice(d->device);
    if (hipEventSynchronize(d->ring_ev[slot]) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, d->ring_ev0[slot], d->ring_ev[slot]) != hipSuccess) return -1.0f;
    return ms;
}

// ---------------------------------------------------------------- Monte-Carlo
static int ensure_codewords(acg_ldpc_decoder *d, const acg_ldpc_mc_cfg *cfg) {
    if (!cfg->codewords || cfg->n_codewords <= 0) return 0;
    // the device copy is keyed on the CONTENT of the host array (a pointer can be recycled for different words)
    // (hashed 8 bytes at a time in four independent lanes: the byte-wise loop took 2.5 ms for 8192 x 280 codewords — more than
    // an early-exit launch over a million frames — on every Monte-Carlo call)
    uint64_t h = 1469598103934665603ull;
    {
        const uint8_t *pb = cfg->codewords;
        const size_t nb = (size_t) cfg->n_codewords * (size_t) d->c.n;
        uint64_t hl[4] = {h, h ^ 0x9E3779B97F4A7C15ull, h ^ 0xC2B2AE3D27D4EB4Full, h ^ 0x165667B19E3779F9ull};
        size_t i = 0;
        for (; i + 32 <= nb; i += 32)
            for (int k = 0; k < 4; k++) {
                uint64_t w;
                std::memcpy(&w, pb + i + 8 * k, 8);
                // any non-zero byte means bit 1 (the reference reads '1' cells, others are 0): normalise every byte to 0 / 1
                w |= w >> 4;
                w |= w >> 2;
                w |= w >> 1;
                w &= 0x0101010101010101ull;
                hl[k] = (hl[k] ^ w) * 1099511628211ull;
                hl[k] ^= hl[k] >> 29;
            }
        for (; i < nb; i++) hl[0] = (hl[0] ^ (uint64_t) (pb[i] != 0)) * 1099511628211ull;
        h = ((hl[0] * 31 + hl[1]) * 31 + hl[2]) * 31 + hl[3];
    }
    if (d->cw_dev && d->cw_hash == h && d->cw_count == cfg->n_codewords) return 0;
    if (d->cw_dev) (void) hipFree(d->cw_dev);
    d->cw_dev = nullptr;
    const int n = d->c.n, nwords = (n + 31) / 32;
    std::vector<uint32_t> packed((size_t) cfg->n_codewords * nwords, 0u);
    for (int64_t f = 0; f < cfg->n_codewords; f++)
        for (int v = 0; v < n; v++)
            if (cfg->codewords[(size_t) f * n + v]) packed[(size_t) f * nwords + (v >> 5)] |= 1u << (v & 31);
    HIP_OK(hipMalloc((void **) &d->cw_dev, packed.size() * sizeof(uint32_t)));
    HIP_OK(hipMemcpy(d->cw_dev, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    d->cw_hash = h;
    d->cw_count = cfg->n_codewords;
    return 0;
}

void acg_ldpc_mc_merge(acg_ldpc_mc_result *a, const acg_ldpc_mc_result *b) {
    // merge_exp_results, experiment.h:70-78
    a->correct += b->correct;
    a->pseudo += b->pseudo;
    a->total += b->total;
    a->sum_hamming += b->sum_hamming;
    a->sum_hamming_ok += b->sum_hamming_ok;
    a->sum_hamming_wrong += b->sum_hamming_wrong;
    a->sum_iters += b->sum_iters;
    a->time_sec += b->time_sec;
    a->kernel_ms += b->kernel_ms;
}

static int mc_run_host_noise(acg_ldpc_decoder *d, const acg_ldpc_mc_cfg *cfg, acg_ldpc_mc_result *res) {
    // Bit-exact experiment.h:80-123 (single-threaded order): frame g (0-based global index) is
    // seeded mt19937(g+1) (:90-97), transmitted with libstdc++ normal_distribution (channel.h:18-26),
    // decoded on the device, classified on the host.
    const int n = d->c.n;
    const int64_t chunk_max = 1 << 16;
    std::vector<double> y;
    std::vector<uint8_t> bits, ok;
    std::vector<int32_t> iters;
    const double sigma = std::sqrt(std::pow(10, -(cfg->snr / 10)) / 2);
    for (int64_t f0 = 0; f0 < cfg->frames; f0 += chunk_max) {
        const int64_t fc = std::min(chunk_max, cfg->frames - f0);
        y.resize((size_t) fc * n);
        bits.resize((size_t) fc * n);
        ok.resize((size_t) fc);
        iters.resize((size_t) fc);
        for (int64_t f = 0; f < fc; f++) {
            const int64_t gidx = cfg->first_frame + f0 + f;
            const uint8_t *cw = cfg->codewords ? cfg->codewords + (size_t) (gidx % cfg->n_codewords) * n : nullptr;
            std::mt19937 rnd((uint32_t) (gidx + 1));
            std::normal_distribution<double> dst(0, sigma);
            for (int i = 0; i < n; i++) y[(size_t) f * n + i] = ((cw && cw[i]) ? -1.0 : 1.0) + dst(rnd);
        }
        if (int rc = acg_ldpc_decode_batch(d, y.data(), fc, cfg->snr, bits.data(), ok.data(), iters.data())) return rc;
        res->kernel_ms += acg_ldpc_decoder_last_kernel_ms(d);
        for (int64_t f = 0; f < fc; f++) {
            const int64_t gidx = cfg->first_frame + f0 + f;
            const uint8_t *cw = cfg->codewords ? cfg->codewords + (size_t) (gidx % cfg->n_codewords) * n : nullptr;
            const uint8_t *b = &bits[(size_t) f * n];
            bool is_correct = false;
            if (ok[f] && code_is_codeword(d->c, b)) {  // experiment.h:110-111
                bool eq = true;
                for (int i = 0; i < n; i++) eq &= (b[i] == (cw ? cw[i] : 0));
                if (eq) {
                    res->correct++;
                    is_correct = true;
                } else
                    res->pseudo++;
            }
            res->total++;
            int h = 0;
            for (int i = 0; i < n; i++) {
                const bool c1 = cw && cw[i];
                const double yv = y[(size_t) f * n + i];
                if (!c1 && yv <= 0) h++;
                if (c1 && yv > 0) h++;
            }
            res->sum_hamming += h;
            if (is_correct) res->sum_hamming_ok += h;
            else res->sum_hamming_wrong += h;
            res->sum_iters += iters[f];
        }
    }
    return 0;
}

static int acg_ldpc_mc_run_impl(acg_ldpc_decoder *d, const acg_ldpc_mc_cfg *cfg, acg_ldpc_mc_result *res) {
    if (!d || !cfg || !res) {
        set_error("null argument");
        return 1;
    }
    if (cfg->frames < 0 || (cfg->codewords && cfg->n_codewords <= 0)) {
        set_error("bad mc cfg");
        return 1;
    }
    std::memset(res, 0, sizeof(*res));
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    if (cfg->noise == ACG_LDPC_NOISE_HOST_MT19937) {
        rc = mc_run_host_noise(d, cfg, res);
    } else if (d->streamed || d->pair || (d->layered && getenv("ACG_LAY_UNFUSED_MC")) || (d->admm && admm_device_unfused_mc(d->admm, nullptr, nullptr))) {
        // AWGN kernel -> decode -> classify kernel, in bounded chunks, all on the device.  Used by the streamed BP
        // engine (no in-kernel generator) and by the workgroup-per-frame QP-ADMM kernel, whose fused Monte-Carlo
        // variant needs 156 VGPRs (3 waves/SIMD) against 117 (4) for the plain decode: 1.6 M vs 2.7 M frames/s.
        const int32_t *csr_row = nullptr, *csr_col = nullptr;
        if (d->admm) (void) admm_device_unfused_mc(d->admm, &csr_row, &csr_col);
        std::lock_guard<std::mutex> lk(d->mu);
        HIP_OK(hipSetDevice(d->device));
        if ((rc = ensure_codewords(d, cfg))) return rc;
        const int n = d->c.n, nwords = (n + 31) / 32;
        int64_t chunk = std::max<int64_t>(256, std::min<int64_t>(cfg->frames, (int64_t) (1ull << 31) / ((int64_t) n * 4)));
        if (chunk > d->mc_frames) {
            if (d->mc_y) (void) hipFree(d->mc_y);
            d->mc_y = nullptr;
            HIP_OK(hipMalloc((void **) &d->mc_y, (size_t) chunk * n * sizeof(float)));
            d->mc_frames = chunk;
        }
        if (int rc2 = ensure_staging(d, std::min<int64_t>(chunk, std::max<int64_t>(cfg->frames, 1)))) return rc2;
        HIP_OK(hipMemsetAsync(d->counters, 0, sizeof(unsigned long long) * MC_NCOUNTERS, d->stream));
        const double var = std::pow(10, -(cfg->snr / 10)) / 2;
        float kms = 0;
        for (int64_t f0 = 0; f0 < cfg->frames; f0 += chunk) {
            const int64_t fc = std::min(chunk, cfg->frames - f0);
            HIP_OK(awgn_launch(d->mc_y, fc, n, nwords, cfg->first_frame + f0, cfg->seed, cfg->codewords ? d->cw_dev : nullptr,
                               cfg->codewords ? cfg->n_codewords : 1, (float) std::sqrt(var), d->stream));
            DecodeArgs a{};
            a.y = d->mc_y;
            a.y_is_f64 = 0;
            a.frames = fc;
            fill_channel(a, cfg->snr);
            a.out_bits = d->st_bits;
            a.out_ok = d->st_ok;
            a.out_iters = d->st_iters;
            if ((rc = launch_decode(d, a, d->stream))) return rc;
            const int slot = d->last_slot;   // this launch's own event pair (d->mu is held)
            HIP_OK(classify_launch(d->mc_y, d->st_bits, d->st_ok, d->st_iters, fc, n, nwords, cfg->first_frame + f0,
                                   cfg->codewords ? d->cw_dev : nullptr, cfg->codewords ? cfg->n_codewords : 1,
                                   d->counters, csr_row, csr_col, d->c.m, d->stream));
            HIP_OK(hipStreamSynchronize(d->stream));
            float ms = 0;
            if (hipEventElapsedTime(&ms, d->ring_ev0[slot], d->ring_ev[slot]) == hipSuccess) kms += ms;
        }
        unsigned long long h[MC_NCOUNTERS];
        HIP_OK(hipMemcpy(h, d->counters, sizeof(h), hipMemcpyDeviceToHost));
        res->correct = (int64_t) h[MC_CORRECT];
        res->pseudo = (int64_t) h[MC_PSEUDO];
        res->total = (int64_t) h[MC_TOTAL];
        res->sum_hamming = (int64_t) h[MC_HAM];
        res->sum_hamming_ok = (int64_t) h[MC_HAM_OK];
        res->sum_hamming_wrong = (int64_t) h[MC_HAM_WRONG];
        res->sum_iters = (int64_t) h[MC_ITERS];
        res->kernel_ms = kms;
    } else {
        std::lock_guard<std::mutex> lk(d->mu);
        HIP_OK(hipSetDevice(d->device));
        if ((rc = ensure_codewords(d, cfg))) return rc;
        HIP_OK(hipMemsetAsync(d->counters, 0, sizeof(unsigned long long) * MC_NCOUNTERS, d->stream));
        DecodeArgs a{};
        a.frames = cfg->frames;
        fill_channel(a, cfg->snr);
        a.mc = 1;
        a.seed = cfg->seed;
        a.first_frame = cfg->first_frame;
        a.cw_packed = cfg->codewords ? d->cw_dev : nullptr;
        a.n_cw = cfg->codewords ? cfg->n_codewords : 1;
        a.counters = d->counters;
        if ((rc = launch_decode(d, a, d->stream))) return rc;
        const int slot = d->last_slot;   // this launch's own event pair (d->mu is held)
        unsigned long long h[MC_NCOUNTERS];
        HIP_OK(hipMemcpyAsync(h, d->counters, sizeof(h), hipMemcpyDeviceToHost, d->stream));
        HIP_OK(hipStreamSynchronize(d->stream));
        res->correct = (int64_t) h[MC_CORRECT];
        res->pseudo = (int64_t) h[MC_PSEUDO];
        res->total = (int64_t) h[MC_TOTAL];
        res->sum_hamming = (int64_t) h[MC_HAM];
        res->sum_hamming_ok = (int64_t) h[MC_HAM_OK];
        res->sum_hamming_wrong = (int64_t) h[MC_HAM_WRONG];
        res->sum_iters = (int64_t) h[MC_ITERS];
        float ms = 0;
        if (cfg->frames > 0 && slot >= 0 && hipEventElapsedTime(&ms, d->ring_ev0[slot], d->ring_ev[slot]) == hipSuccess) res->kernel_ms = ms;
    }
    res->time_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int acg_ldpc_mc_run(acg_ldpc_decoder *d, const acg_ldpc_mc_cfg *cfg, acg_ldpc_mc_result *res) {
    return guarded([&] { return acg_ldpc_mc_run_impl(d, cfg, res); });
}

static int acg_ldpc_awgn_dev_impl(acg_ldpc_decoder *d, const acg_ldpc_mc_cfg *cfg, float *y_dev, void *stream) {
    if (!d || !cfg || !y_dev) {
        set_error("null argument");
        return 1;
    }
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_OK(hipSetDevice(d->device));
    if (int rc = ensure_codewords(d, cfg)) return rc;
    const double var = std::pow(10, -(cfg->snr / 10)) / 2;
    HIP_OK(awgn_launch(y_dev, cfg->frames, d->c.n, (d->c.n + 31) / 32, cfg->first_frame, cfg->seed,
                       cfg->codewords ? d->cw_dev : nullptr, cfg->codewords ? cfg->n_codewords : 1,
                       (float) std::sqrt(var), stream ? (hipStream_t) stream : d->stream));
    return 0;
}

int acg_ldpc_awgn_dev(acg_ldpc_decoder *d, const acg_ldpc_mc_cfg *cfg, float *y_dev, void *stream) {
    return guarded([&] { return acg_ldpc_awgn_dev_impl(d, cfg, y_dev, stream); });
}

// ---------------------------------------------------------------- host generators
static int acg_ldpc_gen_codewords_impl(const uint8_t *G, int32_t k, int32_t n, uint32_t seed, int64_t count, uint8_t *out) {
    if (!G || !out || k <= 0 || n <= 0 || count < 0) {
        set_error("bad argument");
        return 1;
    }
    std::mt19937 rnd(seed);  // main.cpp:63
    for (int64_t f = 0; f < count; f++) {
        uint8_t *res = out + (size_t) f * n;
        std::memset(res, 0, (size_t) n);
        for (int i = 0; i < k; i++)
            if (rnd() % 2 == 0) {  // channel.h:33
                const uint8_t *row = G + (size_t) i * n;
                for (int j = 0; j < n; j++) res[j] ^= (row[j] ? 1 : 0);
            }
    }
    return 0;
}

int acg_ldpc_gen_codewords(const uint8_t *G, int32_t k, int32_t n, uint32_t seed, int64_t count, uint8_t *out) {
    return guarded([&] { return acg_ldpc_gen_codewords_impl(G, k, n, seed, count, out); });
}

double acg_ldpc_llr_variance(double snr) { return std::pow(10, -(snr / 10)) / 2; }

static int acg_ldpc_transmit_host_impl(const uint8_t *codewords, int64_t n_codewords, int32_t n, int64_t first_frame,
                           int64_t frames, double snr, double *y) {
    if (!y || n <= 0 || frames < 0 || (codewords && n_codewords <= 0)) {
        set_error("bad argument");
        return 1;
    }
    const double sigma = std::sqrt(acg_ldpc_llr_variance(snr));  // channel.h:20
    for (int64_t f = 0; f < frames; f++) {
        const int64_t gidx = first_frame + f;
        const uint8_t *cw = codewords ? codewords + (size_t) (gidx % n_codewords) * n : nullptr;
        std::mt19937 rnd((uint32_t) (gidx + 1));              // experiment.h:90-97, single-threaded order
        std::normal_distribution<double> dst(0, sigma);       // channel.h:22
        for (int i = 0; i < n; i++) y[(size_t) f * n + i] = ((cw && cw[i]) ? -1.0 : 1.0) + dst(rnd);
    }
    return 0;
}

int acg_ldpc_transmit_host(const uint8_t *codewords, int64_t n_codewords, int32_t n, int64_t first_frame,
                           int64_t frames, double snr, double *y) {
    return guarded([&] { return acg_ldpc_transmit_host_impl(codewords, n_codewords, n, first_frame, frames, snr, y); });
}

// ---------------------------------------------------------------- debug helpers (tests only)
static int acg_ldpc_debug_bp_trace_impl(const acg_ldpc_code *code, const double *y, int32_t frames, double snr, int32_t iters,
                            int32_t f64, int32_t engine, int32_t lanes_per_frame, double *c2v, double *v2c_mag,
                            double *v2c_sgn, double *post) {
    if (!code || !y || frames < 1 || frames > 64 || iters < 1) {
        set_error("bad argument (1..64 frames, iters >= 1)");
        return 1;
    }
    const bool fused = (engine == ACG_LDPC_ENGINE_FUSED);
    acg_ldpc_params p;
    acg_ldpc_params_default(&p);
    p.algo = ACG_LDPC_BP_SUMPRODUCT;
    p.max_iter = iters;
    p.early_exit = 0;
    p.engine = fused ? ACG_LDPC_ENGINE_FUSED : ACG_LDPC_ENGINE_STREAMED;
    p.lanes_per_frame = fused ? lanes_per_frame : 0;
    p.precision = f64 ? ACG_LDPC_PREC_F64 : ACG_LDPC_PREC_DEFAULT;
    acg_ldpc_decoder *d = nullptr;
    if (int rc = acg_ldpc_decoder_create(code, &p, &d)) return rc;
    const int n = d->c.n, E = d->c.E;
    const size_t ts = f64 ? 8 : 4;
    // words per frame of the three dumps: streamed [E][64] / [E][64] / [n][64]; fused [frame][a_words] x2 / [frame][n_vpass*L]
    size_t wc = (size_t) E * 64, wp = (size_t) n * 64;
    if (fused) {
        const void *kp = nullptr;
        if (d->variant == 2 && d->maxd <= 8) kp = bp_kernel_ptr_dbg(f64, d->L);
        else if (d->variant == -1 && d->L == 256 && d->blk_idxlds && !d->blk_idxreg) kp = bp_block_kernel_ptr_dbg(f64);
        if (!kp) {
            set_error("no debug instance of the fused kernel for this code / lanes_per_frame");
            acg_ldpc_decoder_destroy(d);
            return 3;
        }
        if (d->lds_block > 64 * 1024) (void) hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block);
        d->kernel[0] = kp;
        wc = (size_t) frames * d->tab.a_words;
        wp = (size_t) frames * d->lay.n_vpass * d->L;
    }
    void *dc = nullptr, *dv = nullptr, *dp = nullptr;
    int rc = 0;
    do {
        if (hipMalloc(&dc, wc * ts) != hipSuccess || hipMalloc(&dv, wc * ts) != hipSuccess ||
            hipMalloc(&dp, wp * ts) != hipSuccess) { set_error("hipMalloc failed"); rc = 10; break; }
        if ((rc = ensure_staging(d, frames))) break;
        if (hipMemcpy(d->st_y, y, (size_t) frames * n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = 10; break; }
        DecodeArgs a{};
        a.y = d->st_y;
        a.y_is_f64 = 1;
        a.frames = frames;
        fill_channel(a, snr);
        a.out_bits = d->st_bits;
        a.out_ok = d->st_ok;
        a.out_iters = d->st_iters;
        a.dbg_c2v = dc;
        a.dbg_v2c = dv;
        a.dbg_post = dp;
        if ((rc = launch_decode(d, a, d->stream))) break;
        if (hipStreamSynchronize(d->stream) != hipSuccess) { set_error("sync failed"); rc = 10; break; }
        std::vector<unsigned char> hc(wc * ts), hv(wc * ts), hp(wp * ts);
        (void) hipMemcpy(hc.data(), dc, hc.size(), hipMemcpyDeviceToHost);
        (void) hipMemcpy(hv.data(), dv, hv.size(), hipMemcpyDeviceToHost);
        (void) hipMemcpy(hp.data(), dp, hp.size(), hipMemcpyDeviceToHost);
        // the fp32 kernels work in the log2(e)-scaled message domain (bp_core.inc: Dom<float>): undo it here
        const double unscale = f64 ? 1.0 : 0.693147180559945309;
        auto get = [&](const std::vector<unsigned char> &b, size_t idx) -> double {
            if (f64) return reinterpret_cast<const double *>(b.data())[idx];
            return unscale * (double) reinterpret_cast<const float *>(b.data())[idx];
        };
        // a v->c word = magnitude | hard-decision bit in the LSB | sign: strip the LSB before reading it
        auto get_v2c = [&](size_t idx) -> double {
            if (f64) {
                uint64_t u = reinterpret_cast<const uint64_t *>(hv.data())[idx] & ~1ull;
                double w;
                std::memcpy(&w, &u, 8);
                return w;
            }
            uint32_t u = reinterpret_cast<const uint32_t *>(hv.data())[idx] & ~1u;
            float wf;
            std::memcpy(&wf, &u, 4);
            return unscale * (double) wf;
        };
        // where edge e (check-major, variables ascending — the oracle's trace order) and variable v live in the dumps
        std::vector<size_t> epos((size_t) E), vslot((size_t) n, (size_t) -1);
        if (fused) {
            const BpLayout &lay = d->lay;
            for (int sl = 0; sl < lay.n_cpass * lay.L; sl++) {
                const int chk = lay.c_chk[sl];
                if (chk < 0) continue;
                const int pss = sl / lay.L, l = sl % lay.L;
                for (int j = 0; j < d->c.row_ptr[chk + 1] - d->c.row_ptr[chk]; j++)
                    epos[(size_t) d->c.row_ptr[chk] + j] = (size_t) lay.c_off[pss] + (size_t) j * lay.L + l;
            }
            for (int sl = 0; sl < lay.n_vpass * lay.L; sl++)
                if (lay.v_var[sl] >= 0) vslot[lay.v_var[sl]] = (size_t) sl;
        }
        for (int f = 0; f < frames; f++) {
            auto eidx = [&](int e) { return fused ? (size_t) f * d->tab.a_words + epos[e] : (size_t) e * 64 + f; };
            for (int e = 0; e < E; e++) {
                c2v[(size_t) f * E + e] = get(hc, eidx(e));
                const double w = get_v2c(eidx(e));
                v2c_mag[(size_t) f * E + e] = std::fabs(w);
                v2c_sgn[(size_t) f * E + e] = std::signbit(w) ? -1.0 : 1.0;
            }
            for (int v = 0; v < n; v++) {
                if (!fused) {
                    post[(size_t) f * n + v] = get(hp, (size_t) v * 64 + f);
                    continue;
                }
                // estimate() = llr + sum of the c->v mailbox (bp.h:85-90), summed here from the kernel's own c->v words
                // and channel LLR (slot order dump), checks ascending
                double sum = 0;
                for (int k = d->c.col_ptr[v]; k < d->c.col_ptr[v + 1]; k++) sum += c2v[(size_t) f * E + d->c.col_edge[k]];
                post[(size_t) f * n + v] = get(hp, (size_t) f * d->lay.n_vpass * d->L + vslot[v]) + sum;
            }
        }
    } while (0);
    if (dc) (void) hipFree(dc);
    if (dv) (void) hipFree(dv);
    if (dp) (void) hipFree(dp);
    acg_ldpc_decoder_destroy(d);
    return rc;
}

int acg_ldpc_debug_bp_trace(const acg_ldpc_code *code, const double *y, int32_t frames, double snr, int32_t iters,
                            int32_t f64, int32_t engine, int32_t lanes_per_frame, double *c2v, double *v2c_mag,
                            double *v2c_sgn, double *post) {
    return guarded([&] { return acg_ldpc_debug_bp_trace_impl(code, y, frames, snr, iters, f64, engine, lanes_per_frame, c2v, v2c_mag, v2c_sgn, post); });
}

int acg_ldpc_debug_ring_tasks(const acg_ldpc_code *code, int32_t *n_ctask, int32_t *n_vtask, int32_t *ctask, int32_t *vtask, int64_t cap,
                              int32_t *consts) {
    return guarded([&]() -> int {
        if (!code) {
            set_error("null argument");
            return 1;
        }
        RingTasks rt;
        ring_tasks_build(code->c, rt);
        if (n_ctask) *n_ctask = rt.n_ctask;
        if (n_vtask) *n_vtask = rt.n_vtask;
        if (ctask)
            for (int64_t i = 0; i < std::min<int64_t>(cap, (int64_t) rt.ctask.size()); i++) ctask[i] = rt.ctask[(size_t) i];
        if (vtask)
            for (int64_t i = 0; i < std::min<int64_t>(cap, (int64_t) rt.vtask.size()); i++) vtask[i] = rt.vtask[(size_t) i];
        if (consts) {
            consts[0] = RING_WAVES;
            consts[1] = RING_SLOTS;
            consts[2] = RING_SLOT_LINES;
            consts[3] = RING_VAR_EDGE_LINES;
        }
        return 0;
    });
}

int acg_ldpc_debug_layers(const acg_ldpc_code *code, int32_t *lanes, int32_t *n_layers, int32_t *qc_Z, int32_t *chk, int64_t cap) {
    return guarded([&]() -> int {
        if (!code) {
            set_error("null argument");
            return 1;
        }
        LayeredLayout ll;
        if (!bp_layered_build(code->c, ll)) return 3;
        if (lanes) *lanes = ll.G;
        if (n_layers) *n_layers = ll.n_layers;
        if (qc_Z) *qc_Z = ll.qc ? ll.Z : 0;
        if (chk)
            for (int64_t i = 0; i < std::min<int64_t>(cap, (int64_t) ll.chk.size()); i++) chk[i] = ll.chk[(size_t) i];
        return 0;
    });
}

static int acg_ldpc_debug_phi_impl(const void *x_host, void *out_host, int32_t n, int32_t f64) {
    const size_t es = f64 ? 8 : 4;
    void *dx = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc(&dx, es * n));
    HIP_OK(hipMalloc(&dout, es * n));
    HIP_OK(hipMemcpy(dx, x_host, es * n, hipMemcpyHostToDevice));
    HIP_OK(phi_debug_launch(dx, dout, n, f64, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out_host, dout, es * n, hipMemcpyDeviceToHost));
    (void) hipFree(dx);
    (void) hipFree(dout);
    return 0;
}

int acg_ldpc_debug_phi(const void *x_host, void *out_host, int32_t n, int32_t f64) {
    return guarded([&] { return acg_ldpc_debug_phi_impl(x_host, out_host, n, f64); });
}

}  // extern "C"
