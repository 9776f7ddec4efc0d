/*
 * acg_ldpc.h — C ABI of the MI355X-native batched LDPC decoder (libacg_ldpc_hip.so).
 *
 * Drop-in boundary for ONE path of GreatDrake/acg-alp-ldpc: the per-frame call
 *     pair<TCodeword,bool> Decoder::decode(const TMatrix &H, const TFVector &y, double snr)
 * (reference algo/algo.h:8) as implemented by BeliefPropagationDecoder (algo/bp.h:208-222) and
 * QPADMMDecoder (algo/qp_admm.h:180-194), plus the Monte-Carlo loop that drives it
 * (experiment.h:80-139) with its AWGN generator (utils/channel.h:18-26).
 *
 * Plain C: opaque handles, caller-owned buffers, int return codes (0 = ok), no exceptions,
 * no torch / STL types.  Every entry point cites the reference interface it replaces.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Threading: a decoder handle owns one HIP stream + workspace on one device; host calls on the same
 * handle are serialised by an internal mutex (the reference calls one decoder object from
 * THREADS_NUM pthreads, experiment.h:101,127-130 — that keeps working, one handle per thread
 * is faster).  Code handles are immutable after creation and may be shared.
 * Streams: acg_ldpc_decode_batch_dev is asynchronous on the caller's stream.  Launches of ONE handle on
 * DIFFERENT streams may overlap on the device: each launch owns its work counter (a ring of 32 per handle,
 * a slot is reused only behind the launch that held it).  The streamed BP engine keeps its message slabs
 * in the handle, so its launches are ordered on the device (a launch on another stream waits for the
 * previous one through an event) — correct, not concurrent; use one handle per stream to overlap those.
 */
#ifndef ACG_LDPC_H
#define ACG_LDPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct acg_ldpc_code acg_ldpc_code;       /* parity-check matrix + analysed Tanner graph (host) */
typedef struct acg_ldpc_decoder acg_ldpc_decoder; /* device-resident graph, workspace, stream, params */

/* algorithm selector */
enum {
    ACG_LDPC_BP_SUMPRODUCT = 0, /* algo/bp.h — the reference's BP (phi-domain sum-product) */
    ACG_LDPC_BP_MINSUM = 1,     /* build-added (north_star); NOT in the reference: parity unpinned */
    ACG_LDPC_QPADMM = 2         /* algo/qp_admm.h */
};

/* arithmetic selector */
enum {
    ACG_LDPC_PREC_DEFAULT = 0, /* BP: fp32 messages; QP-ADMM: fp64 */
    ACG_LDPC_PREC_F64 = 1,     /* everything fp64 */
    ACG_LDPC_PREC_F32 = 2,     /* everything fp32 (QP-ADMM then matches FER only) */
    ACG_LDPC_PREC_F16 = 3      /* min-sum only (build-added variant, parity unpinned): half-precision messages, two frames
                                  per workgroup sharing every LDS word, index and barrier — for codes whose messages fill
                                  the LDS of a CU (check degree <= 8, variable degree <= 4, n <= 12288) */
};

/* BP engine selector */
enum {
    ACG_LDPC_ENGINE_AUTO = 0,     /* fused when a frame's messages fit in LDS, else streamed */
    ACG_LDPC_ENGINE_FUSED = 1,    /* messages resident in LDS for the whole decode (HBM: symbols in, bits out) */
    ACG_LDPC_ENGINE_STREAMED = 2  /* messages [edge][frame] in HBM, one lane per frame, coalesced sweeps (any code size) */
};

/* BP message schedule */
enum {
    ACG_LDPC_SCHEDULE_FLOODING = 0, /* the reference's schedule (bp.h:183-199): all checks, then all variables */
    ACG_LDPC_SCHEDULE_LAYERED = 1   /* build-added: the block rows of a quasi-cyclic H (or, for any other H, groups
                                       of checks that share no variable) are processed in sequence and the posteriors are
                                       updated in place after every layer, so one iteration does the work of about two
                                       flooding sweeps.  A DIFFERENT algorithm from the reference's BeliefPropagationDecoder:
                                       parity is FER-level only.  With ACG_LDPC_BP_MINSUM: normalised min-sum (parity unpinned
                                       anyway); with ACG_LDPC_BP_SUMPRODUCT: the reference's check rule (bp.h:49-57) in the
                                       layered message order — the reference's FER at about half its iterations.  fp32
                                       posteriors; messages fp32, or fp16 with ACG_LDPC_PREC_F16. */
};

/* noise source for acg_ldpc_mc_run */
enum {
    ACG_LDPC_NOISE_DEVICE_PHILOX = 0, /* counter-based, keyed on (seed, global frame, symbol): same
                                         frames for any GPU count; statistically validated only */
    ACG_LDPC_NOISE_HOST_MT19937 = 1   /* bit-exact experiment.h:97-99: frame i <- mt19937(i+1) +
                                         libstdc++ normal_distribution, generated on the host */
};

typedef struct acg_ldpc_params {
    int32_t algo;       /* ACG_LDPC_* */
    int32_t max_iter;   /* BeliefPropagationDecoder(max_iter) bp.h:210 / QPADMMDecoder max_iter qp_admm.h:182 */
    double alpha;       /* QP-ADMM (qp_admm.h:182) */
    double mu;          /* QP-ADMM */
    double eps_stop;    /* QP-ADMM residual threshold (qp_admm.h:161) */
    double ms_scale;    /* min-sum normalisation factor (1.0 = plain) */
    int32_t early_exit; /* 1 = reference semantics: stop a frame at its first zero syndrome (bp.h:195-196)
                           / residual < eps (qp_admm.h:161).  0 = fixed work: run max_iter sweeps for every
                           frame, output LATCHED at the first zero syndrome (identical results). */
    int32_t precision;  /* ACG_LDPC_PREC_* */
    int32_t device;     /* HIP device ordinal; -1 = current device */
    int32_t lanes_per_frame; /* 0 = auto; 16/32/64: that many lanes of a wavefront cooperate on one frame; 256 (BP also
                                1024): one workgroup per frame (QP-ADMM then picks 128, 192 or 256 threads itself) */
    int32_t engine;     /* ACG_LDPC_ENGINE_* (BP only) */
    int32_t fast_setup; /* 0 = default: spend up to ~1 s per decoder on the static LDS placement of the QP-ADMM kernel
                           (bank-conflict search; cached per parity-check matrix inside the process);
                           1 = skip that search (throw-away decoders, e.g. one per proposal of the check-matrix local
                           search, optimize_H.cpp:89-104).  Results are identical either way. */
    int32_t schedule;   /* ACG_LDPC_SCHEDULE_* (BP only; default flooding) */
} acg_ldpc_params;

void acg_ldpc_params_default(acg_ldpc_params *p);

/* last error message of the calling thread ("" if none).  The reference aborts via assert();
 * here every failure is an error code + message and nothing is silently computed on the CPU. */
const char *acg_ldpc_last_error(void);

/* 1 if a HIP device is usable from this process, 0 otherwise (never falls back to a CPU decode). */
int acg_ldpc_device_available(void);

/* ---- parity-check matrix --------------------------------------------------------------- */

/* replaces: TMatrix (utils/codeword.h:18) handed to decode() on every call; analysed once here
 * (the reference re-scans H per frame: bp.h:136-153, qp_admm.h:15-21,60-66). H: m*n bytes, !=0 -> 1. */
int acg_ldpc_code_from_dense(const uint8_t *H, int32_t m, int32_t n, acg_ldpc_code **out);
/* replaces read_pcm (utils/parse_data.h:6-25), same quirks */
int acg_ldpc_code_load_txt(const char *path, acg_ldpc_code **out);
/* replaces save_matrix (utils/parse_data.h:44-54) */
int acg_ldpc_code_save_txt(const acg_ldpc_code *code, const char *path);
void acg_ldpc_code_destroy(acg_ldpc_code *code);
/* m checks, n variables, E edges (ones of H) */
void acg_ldpc_code_dims(const acg_ldpc_code *code, int32_t *m, int32_t *n, int32_t *E);
/* dense copy back (m*n bytes) */
void acg_ldpc_code_dense(const acg_ldpc_code *code, uint8_t *H);
/* QP-ADMM problem shape of ConstructADMMProblem (qp_admm.h:13-102) */
void acg_ldpc_code_admm_shape(const acg_ldpc_code *code, int32_t *n_var, int32_t *n_con, int32_t *nnz,
                              double *e_min, double *e_max);
/* replaces GetOrtogonal (utils/codeword.h:97-128): G must hold (n-m)*n bytes; returns 0 ok,
 * 1 if a row of H vanishes during elimination (the reference's {TMatrix(), false}). */
int acg_ldpc_code_generator(const acg_ldpc_code *code, uint8_t *G);
/* replaces IsCodeword (utils/codeword.h:90-95): 1 / 0 */
int acg_ldpc_code_is_codeword(const acg_ldpc_code *code, const uint8_t *bits);

/* ---- decoder --------------------------------------------------------------------------- */

/* replaces make_shared<BeliefPropagationDecoder>(it) / make_shared<QPADMMDecoder>(a,mu,it,eps)
 * (main.cpp:28-40).  Fails (non-zero) when no HIP device is present. */
int acg_ldpc_decoder_create(const acg_ldpc_code *code, const acg_ldpc_params *params, acg_ldpc_decoder **out);
void acg_ldpc_decoder_destroy(acg_ldpc_decoder *dec);
/* replaces Decoder::name() (algo/algo.h:10): "BP" (bp.h:218), "QP-ADMM" (qp_admm.h:189), "MS" */
const char *acg_ldpc_decoder_name(const acg_ldpc_decoder *dec);

/* replaces Decoder::decode(H, y, snr) (algo/algo.h:8) for `frames` frames at once.
 *   y      host, frames*n doubles, raw channel symbols (NOT LLRs; llr = 2y/sigma^2 is formed inside,
 *          channel.h:12-16, bp.h:66, qp_admm.h:27)
 *   bits   host, frames*n bytes (0/1).  BP failure -> zeros (the reference returns an empty vector, bp.h:198)
 *   ok     host, frames bytes: the reference's bool (BP: zero syndrome reached; QP-ADMM: always 1 unless the
 *          e_min*mu<=alpha guard fires, qp_admm.h:112-114,177)
 *   iters  host, frames int32 or NULL: sweeps executed until exit (BP: iteration of the first zero syndrome)
 * Large batches are pipelined in chunks of at most 65536 frames / 256 MiB of symbols through two pinned staging sets
 * (a process-wide pool of host threads, created on the first batch of >= 4096 frames, copies / unpacks while the GPU
 * decodes the previous chunk); the rate is PCIe-bound: 8 bytes in + 1 byte out per symbol. */
int acg_ldpc_decode_batch(acg_ldpc_decoder *dec, const double *y, int64_t frames, double snr, uint8_t *bits,
                          uint8_t *ok, int32_t *iters);

/* Same with single-precision symbols (half the PCIe bytes).  The LLR is formed as (double) y * (2 / sigma^2), rounded
 * to the kernel's message type — the path of acg_ldpc_decode_batch_dev with y_is_f64 = 0.  A caller holding doubles
 * who wants the reference's exact llr = 2y / sigma^2 (channel.h:14-16) uses acg_ldpc_decode_batch. */
int acg_ldpc_decode_batch_f32(acg_ldpc_decoder *dec, const float *y, int64_t frames, double snr, uint8_t *bits,
                              uint8_t *ok, int32_t *iters);

/* Same, buffers already resident in HBM (this is what bench.py times).
 *   y_dev        device, frames*n of float (y_is_f64=0) or double (y_is_f64=1)
 *   bits_dev     device, frames*words uint32, words = (n+31)/32; bit v of a frame = word v>>5, bit v&31
 *   ok_dev       device, frames bytes;  iters_dev device, frames int32 (may be NULL)
 *   stream       hipStream_t to launch on (NULL = the decoder's own stream); asynchronous.  The output buffers of
 *                two launches in flight must not overlap; see "Streams" at the top of this file. */
int acg_ldpc_decode_batch_dev(acg_ldpc_decoder *dec, const void *y_dev, int32_t y_is_f64, int64_t frames,
                              double snr, uint32_t *bits_dev, uint8_t *ok_dev, int32_t *iters_dev, void *stream);
/* block until the decoder's own stream is idle */
int acg_ldpc_decoder_sync(acg_ldpc_decoder *dec);
/* duration in ms of the most recent decode/mc kernel launch on this handle, measured with the HIP event pair that
 * launch recorded on its own stream (every launch owns a pair: launches of one handle overlapping on two streams never
 * pair each other's events); synchronises on the stop event */
float acg_ldpc_decoder_last_kernel_ms(acg_ldpc_decoder *dec);
/* bytes of LDS per frame, frames resident per CU, lanes per frame chosen for this code (diagnostics) */
void acg_ldpc_decoder_layout(const acg_ldpc_decoder *dec, int32_t *lds_bytes_per_frame, int32_t *lanes_per_frame,
                             int32_t *frames_per_block, int32_t *grid_blocks);

/* one line naming the engine / kernel instance / launch shape this handle uses (diagnostics; bench.py records it with every
 * timed leg).  Writes at most cap bytes incl. the terminating 0; returns the size the full text needs. */
int32_t acg_ldpc_decoder_describe(const acg_ldpc_decoder *dec, char *buf, int32_t cap);

/* ---- Monte-Carlo loop (experiment.h) --------------------------------------------------- */

typedef struct acg_ldpc_mc_cfg {
    int64_t frames;       /* frames to simulate in THIS call */
    int64_t first_frame;  /* global index of the first frame (shard offset; seeds derive from the global index) */
    double snr;           /* Es/N0 dB, sigma^2 = 10^(-snr/10)/2 (channel.h:12) */
    uint64_t seed;        /* Philox key (device noise).  Host mt19937 mode ignores it: frame i uses mt19937(i+1) */
    int32_t noise;        /* ACG_LDPC_NOISE_* */
    const uint8_t *codewords; /* host, n_codewords*n bytes, frame g transmits codewords[g % n_codewords];
                                 NULL = all-zero codeword */
    int64_t n_codewords;
} acg_ldpc_mc_cfg;

/* mirrors ExperimentResult + HammingDistanceTracker (experiment.h:25-68) */
typedef struct acg_ldpc_mc_result {
    int64_t correct, pseudo, total;
    int64_t sum_hamming, sum_hamming_ok, sum_hamming_wrong;
    int64_t sum_iters;   /* sweeps executed, for the mean-iterations figure */
    double time_sec;     /* wall time of the call */
    double kernel_ms;    /* device time of the decode kernel(s) */
} acg_ldpc_mc_result;

/* replaces multithread_experiment (experiment.h:125-139): transmit + decode + classify
 * (correct / pseudo-codeword / fail) + raw-channel Hamming statistics, all on the device. */
int acg_ldpc_mc_run(acg_ldpc_decoder *dec, const acg_ldpc_mc_cfg *cfg, acg_ldpc_mc_result *res);
/* merge_exp_results (experiment.h:70-78): a += b (used to combine per-GPU shards on the host) */
void acg_ldpc_mc_merge(acg_ldpc_mc_result *a, const acg_ldpc_mc_result *b);

/* ---- host-side generators used by the reference's drivers (bit-exact, libstdc++) ----------- */

/* replaces gen_random_codewords (utils/channel.h:28-44) with std::mt19937(seed): row i of G (k x n bytes)
 * is XORed in when rnd() % 2 == 0.  out: count*n bytes. */
int acg_ldpc_gen_codewords(const uint8_t *G, int32_t k, int32_t n, uint32_t seed, int64_t count, uint8_t *out);
/* replaces transmit (utils/channel.h:18-26) as driven by exp() (experiment.h:97-99): global frame g uses
 * std::mt19937(g+1) and std::normal_distribution<double>(0, sigma); transmits codewords[g % n_codewords]
 * (NULL = all-zero word).  y: frames*n doubles. */
int acg_ldpc_transmit_host(const uint8_t *codewords, int64_t n_codewords, int32_t n, int64_t first_frame,
                           int64_t frames, double snr, double *y);
/* llr_variance (utils/channel.h:12) */
double acg_ldpc_llr_variance(double snr);

/* device-side AWGN only (utils/channel.h:18-26 with the Philox generator): fills y_dev (frames*n floats)
 * for global frames [first_frame, first_frame+frames). codewords as in acg_ldpc_mc_cfg (host pointer). */
int acg_ldpc_awgn_dev(acg_ldpc_decoder *dec, const acg_ldpc_mc_cfg *cfg, float *y_dev, void *stream);

/* diagnostics (used by tests/): evaluates the device phi(x) = -log(tanh(x/2)) (bp.h:34) of the BP kernels
 * on n host values; f64 selects the double variant. */
int acg_ldpc_debug_phi(const void *x_host, void *out_host, int32_t n, int32_t f64);
/* diagnostics: soft state of the device sum-product decoder after `iters` full iterations of bp.h:183-199 without
 * the exit test, for 1..64 frames (y: frames*n doubles).  Outputs are frames*E (edge order: check-major, variables
 * ascending) / frames*n doubles: c2v = messages check->variable, (v2c_mag, v2c_sgn) = the (phi(|x|), sign) pairs
 * variable->check, post = VNode::estimate() (bp.h:85-90).
 * engine: ACG_LDPC_ENGINE_STREAMED (or AUTO) = the HBM engine (fp32: a debug instance of the LDS-DMA ring kernel that ships, with
 * its own sweeps and counted waits; fp64 and node degrees above 12: the register-staged kernel); ACG_LDPC_ENGINE_FUSED = the LDS-resident kernels, read out
 * of LDS by a debug instance of the same kernel: lanes_per_frame 0/32/64 = wavefront groups (node degree <= 8, n <= 12
 * passes), 256 = one workgroup per frame (index table in LDS).  For the fused kernels `post` is the channel LLR plus the
 * sum of the dumped c2v words, added on the host. */
int acg_ldpc_debug_bp_trace(const acg_ldpc_code *code, const double *y, int32_t frames, double snr, int32_t iters,
                            int32_t f64, int32_t engine, int32_t lanes_per_frame, double *c2v, double *v2c_mag,
                            double *v2c_sgn, double *post);

/* diagnostics (host only, no device needed): the task tables of the streamed engine's LDS-DMA ring kernel, 4 int32 per task
 * {first node, nodes, first line / col_ptr entry, lines | wait << 8 | wait_without_stores << 16}; consts (4 int32) receives
 * {wavefronts per workgroup, ring slots, lines per slot, edge lines per variable task}.  tests/test_ring_waits.py replays
 * the kernel's issue order against the counted waits. */
int acg_ldpc_debug_ring_tasks(const acg_ldpc_code *code, int32_t *n_ctask, int32_t *n_vtask, int32_t *ctask, int32_t *vtask,
                              int64_t cap, int32_t *consts);

/* diagnostics (host only, no device needed): the layers of ACG_LDPC_SCHEDULE_LAYERED for this matrix.  lanes = lanes per
 * frame G, n_layers, qc_Z = circulant size if the block rows of a quasi-cyclic H were used (0: greedy colouring);
 * chk (may be NULL) receives n_layers * G check ids in processing order (-1 = empty lane), at most cap entries.
 * Returns 0, or non-zero if the matrix cannot be layered (message in acg_ldpc_last_error). */
int acg_ldpc_debug_layers(const acg_ldpc_code *code, int32_t *lanes, int32_t *n_layers, int32_t *qc_Z, int32_t *chk, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ACG_LDPC_H */
