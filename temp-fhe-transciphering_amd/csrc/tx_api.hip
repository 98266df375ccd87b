// tx_api.hip -- C ABI (include/tfhe_mi355x.h) over the gfx950 kernels: context, key residency,
// the AES round schedules and the max tournament.  Host logic mirrors the reference's
// `main` / `aes_to_lwe_trasnciphering` (server_encrypted_aes_decryption.rs:28-191, 599-707) and
// server_encrypted_compute.rs:203-356; all arithmetic on ciphertexts runs in the kernels of
// tx_kernels.hpp.  There is no CPU fallback: every entry point fails if HIP does.
#include "../../include/tfhe_mi355x.h"
#include "tx_kernels.hpp"
#include "tx_br2.hpp"
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace txd;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return -1; }
extern "C" const char *tx_last_error(void) { return g_err.c_str(); }

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

static const size_t CHUNK_BLOCKS = 128;     // blocks per workspace chunk (16 384 bit ciphertexts: 8.4 GB of Fourier GGSWs)

enum Family { F_KS = 0, F_BR, F_TRACE, F_SS, F_LUT, F_LIN, F_COUNT };

struct tx_ctx {
    tx_params p;
    int device;
    hipStream_t stream;
    // host-output pipeline of tx_transcipher (round 5): result buffers res / res2 alternate by chunk; chunk k's LWEs go to the
    // caller's memory on copy_stream while chunk k + 1 computes on `stream`
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_done[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    uint64_t *res2 = nullptr;
    // resident evaluation keys (Fourier, private layout)
    cplx *bsk_f = nullptr, *ksk_f = nullptr, *auto_f = nullptr, *ss_f = nullptr;
    size_t auto_split_stride = 0;
    cplx *tw = nullptr, *tw256 = nullptr;
    uint32_t *br8_fault = nullptr;          // host-mapped word: a wave of k_blind_rotate8 that gave up waiting at a software barrier stores 1 (checked per call)
    uint32_t *fair_tab = nullptr;           // k_blind_rotate6: step counters of the workgroups of every CU (tx_br2.hpp)
    uint32_t br6_epoch = 0;                 // launch number of k_blind_rotate6: a table word of another launch is not a competitor
    int num_cus = 0;
    int trace_variant = 4;                  // k_extract_trace4 (generated trace loop); TX_TRACE_VARIANT=0: the hipcc kernel
    int ss_variant = 4;                     // k_scheme_switch4 (generated body); TX_SS_VARIANT=0: the hipcc kernel
    int lut_variant = 0;                    // k_lut_eval (hipcc); TX_LUT_VARIANT=4: k_lut_eval4, the generated CMux tree (bit-identical, 10 % slower:
                                            // 2.30 against 2.09 ms per launch of the medium instance, profiles/r4t_*)
    int br_variant = -1;                    // -1: by launch size (run_br); TX_BR_VARIANT=0|3|4|5 forces a kernel
    int br5_wgs = 0;                        // 0: one workgroup per CU; TX_BR5_WGS caps the persistent grid (experiments)
    int trss_wgs_per_cu = 1;                // TX_TRSS_WGS_PER_CU (experiments)
    int tie_on = 1;                         // 1: balanced top-level decomposition tie (default); 0: tfhe-rs' SignedDecomposer rule (tx_set_decomposer_tie_rule)
    int ctr_share = 1;                      // TX_CTR_SHARE=0: the CTR schedule bootstraps every column of its second round, shared or not
    const char *last_br_kernel = "";        // blind-rotation kernel that took most units of the last call (tx_last_blind_rotate_kernel)
    struct BrUse { const char *kernel; uint64_t units; uint32_t launches; };
    std::vector<BrUse> br_use;              // launches of the current call by kernel (launch_br; does not depend on the events)
    const char *ev_error = nullptr;         // first HIP event call that failed during the current call (ev_begin / ev_end)
    uint64_t *share_tmp = nullptr; size_t share_tmp_bits = 0;      // compact LUT output of the CTR schedule's shared round
    int *d_idx = nullptr; uint8_t *d_pos = nullptr; size_t idx_cap = 0;
    uint64_t *luts = nullptr, *ctr_luts = nullptr;
    bool have_keys = false;
    // workspace (grown on demand)
    size_t ws_blocks = 0, glev_bits = 0;
    size_t reserve_blocks = 0;              // tx_reserve: blocks per chunk the next calls will need (allocated with the keys, or at once)
    uint64_t *st = nullptr, *sm = nullptr, *small = nullptr, *acc = nullptr, *glev = nullptr, *res = nullptr;
    cplx *ggsw_f = nullptr;
    uint8_t *d_bytes = nullptr, *d_ct = nullptr;
    // timing
    struct Ev { int fam; const char *kernel; size_t units; hipEvent_t a, b; };
    std::vector<Ev> evs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    tx_timing timing;
#ifdef TXD_NO_EVENTS   // experiment: what the per-kernel HIP events cost (tools/wall_time.py)
    bool time_enabled = false;
#else
    bool time_enabled = true;
#endif
};

extern "C" void tx_params_aes_tight(tx_params *p)
{   // cbs_lib/src/aes_instances.rs:76-97
    if (!p) return;
    p->n = 768; p->N = 1024; p->k = 2;
    p->pbs_base_log = 23; p->pbs_level = 1;
    p->ks_base_log = 4; p->ks_level = 3; p->ks_N = 256;
    p->auto_base_log = 13; p->auto_level = 3; p->auto_split = 41;
    p->ss_base_log = 17; p->ss_level = 2;
    p->cbs_base_log = 2; p->cbs_level = 7;
    p->log_lut_count = 3;
    p->lwe_std = 0.00000702047462940120;
    p->glwe_std = 0.00000000000000029403601535432533;
}

static cplx unit_root(long double num, long double den)
{   // exp(i pi num / den), rounded once from long double
    const long double PI = 3.14159265358979323846264338327950288L;
    long double a = PI * num / den;
    return make_double2((double)cosl(a), (double)sinl(a));
}

// TX_* experiment knobs: read ONCE (tx_create), validated; a value that is not an integer in [lo, hi] is an error
static int env_knob(const char *name, int lo, int hi, int &out)
{
    const char *e = getenv(name);
    if (!e || !*e) return 0;
    char *end = nullptr;
    long v = strtol(e, &end, 10);
    if (*end != 0 || v < lo || v > hi)
        return fail(std::string("tx_create: ") + name + "=" + e + " is not an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
    out = (int)v;
    return 0;
}

static int upload_tables(tx_ctx *c)
{
    std::vector<cplx> tw(TW_ROWS * 64), t2(192);
    for (int r = 0; r < 8; r++)         // pass A with the lane part of the twist: exp(i pi t (1 + 4 r) / 1024)
        for (int t = 0; t < 64; t++) tw[r * 64 + t] = unit_root(t * (1 + 4 * r), 1024);
    for (int r = 1; r < 8; r++)
        for (int t = 0; t < 64; t++) tw[(7 + r) * 64 + t] = unit_root(2 * ((t & 7) * r), 64);
    for (int j = 0; j < 128; j++) t2[j] = unit_root(j, 256);
    for (int j = 0; j < 64; j++) t2[128 + j] = unit_root(2 * j, 128);
    HIP_OK(hipMalloc(&c->tw, tw.size() * sizeof(cplx)));
    HIP_OK(hipMalloc(&c->tw256, t2.size() * sizeof(cplx)));
    HIP_OK(hipMemcpy(c->tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(c->tw256, t2.data(), t2.size() * sizeof(cplx), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->fair_tab, (size_t)FAIR_TAB_ENTRIES * FAIR_TAB_WORDS * sizeof(uint32_t)));
    HIP_OK(hipMemset(c->fair_tab, 0, (size_t)FAIR_TAB_ENTRIES * FAIR_TAB_WORDS * sizeof(uint32_t)));
    HIP_OK(hipHostMalloc((void **)&c->br8_fault, sizeof(uint32_t), hipHostMallocMapped));
    *c->br8_fault = 0;
    return 0;
}

extern "C" tx_ctx *tx_create(const tx_params *p, int device)
{
    if (!p) { fail("tx_create: null parameter set"); return nullptr; }
    if (p->N != (uint32_t)N || p->k != 2 || p->pbs_base_log != PBS_B || p->pbs_level != PBS_L ||
        p->ks_base_log != KS_B || p->ks_level != KS_L || p->ks_N != KS_N || p->auto_base_log != AUTO_B ||
        p->auto_level != AUTO_L || p->auto_split != AUTO_SPLIT || p->ss_base_log != SS_B || p->ss_level != SS_L ||
        p->cbs_base_log != CBS_B || p->cbs_level != CBS_L || p->log_lut_count != LOG_LUT) {
        fail("tx_create: kernels are specialised for the AES_TIGHT constants (only n may differ)");
        return nullptr;
    }
    if (p->n == 0 || p->n % KS_N != 0 || p->n > 1024) { fail("tx_create: n must be a multiple of 256, at most 1024"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) { fail("tx_create: no such HIP device (the HIP path is mandatory, there is no CPU fallback)"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail("tx_create: hipSetDevice failed"); return nullptr; }
    tx_ctx *c = new tx_ctx();
    c->p = *p; c->device = device; c->stream = nullptr;
    memset(&c->timing, 0, sizeof c->timing);
    // every failure below releases what was created so far (stream, tables): tx_destroy tolerates a partial context
    if (hipStreamCreate(&c->stream) != hipSuccess) { c->stream = nullptr; fail("tx_create: hipStreamCreate failed"); tx_destroy(c); return nullptr; }
    if (upload_tables(c) != 0) { tx_destroy(c); return nullptr; }
#ifdef TXD_EXPERIMENTS
    // experiment knobs (the -DTXD_EXPERIMENTS build only: libtfhe_mi355x_exp.so), parsed once and validated here (behind the
    // stream and the tables, so that a bad value also exercises the release of a half-built context)
    if (env_knob("TX_BR_VARIANT", 0, 8, c->br_variant) || env_knob("TX_TRACE_VARIANT", 0, 4, c->trace_variant) ||
        env_knob("TX_SS_VARIANT", 0, 4, c->ss_variant) || env_knob("TX_LUT_VARIANT", 0, 4, c->lut_variant) || env_knob("TX_BR5_WGS", 1, 1 << 20, c->br5_wgs) ||
        env_knob("TX_TRSS_WGS_PER_CU", 1, 64, c->trss_wgs_per_cu) || env_knob("TX_CTR_SHARE", 0, 1, c->ctr_share)) { tx_destroy(c); return nullptr; }
    {
        bool ok = (c->br_variant == -1 || c->br_variant == 0 || c->br_variant == 4 || c->br_variant == 5 || c->br_variant == 6 || c->br_variant == 7 ||
                   c->br_variant == 8);
#ifdef TXD_WITH_BR3
        ok = ok || c->br_variant == 3;
#endif
        if (!ok || (c->trace_variant != 0 && c->trace_variant != 4) || (c->ss_variant != 0 && c->ss_variant != 4) ||
            (c->lut_variant != 0 && c->lut_variant != 4)) {
            fail("tx_create: TX_BR_VARIANT must be 0, 4, 5, 6, 7 or 8 (3 only in a TXD_WITH_BR3 build), TX_TRACE_VARIANT, TX_SS_VARIANT and TX_LUT_VARIANT 0 or 4");
            tx_destroy(c); return nullptr;
        }
    }
#endif
    // function attributes are per device and process-wide: contexts created from several threads at once set them in turn
    static std::mutex attr_mutex;
    std::lock_guard<std::mutex> attr_lock(attr_mutex);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail("tx_create: hipGetDeviceProperties failed"); tx_destroy(c); return nullptr; }
    c->num_cus = prop.multiProcessorCount;
    // every kernel that keeps ciphertext state in LDS needs more than the 64 KiB default
    auto lds_attr = [](const void *k, size_t bytes) {
        return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    };
    bool ok = lds_attr(reinterpret_cast<const void *>(k_blind_rotate5), LDS_BYTES_BR5) &&
              lds_attr(reinterpret_cast<const void *>(k_blind_rotate6), LDS_BYTES_BR6) &&
              lds_attr(reinterpret_cast<const void *>(k_blind_rotate8), LDS_BYTES_BR7) &&
              lds_attr(reinterpret_cast<const void *>(k_extract_trace4), LDS_BYTES_TR4) &&
              lds_attr(reinterpret_cast<const void *>(k_scheme_switch4), LDS_BYTES_TR4) &&
              lds_attr(reinterpret_cast<const void *>(k_trace_scheme_switch4), LDS_BYTES_TR4);
#ifdef TXD_EXPERIMENTS
    ok = ok && lds_attr(reinterpret_cast<const void *>(k_extract_trace), LDS_BYTES_TRACE) &&
         lds_attr(reinterpret_cast<const void *>(k_trace), LDS_BYTES_TRACE) &&
         lds_attr(reinterpret_cast<const void *>(k_blind_rotate), LDS_BYTES_BR) &&
         lds_attr(reinterpret_cast<const void *>(k_blind_rotate4), LDS_BYTES_BR3) &&
         lds_attr(reinterpret_cast<const void *>(k_blind_rotate7), LDS_BYTES_BR7) &&
         lds_attr(reinterpret_cast<const void *>(k_lut_eval4), LDS_BYTES_TR4);
#ifdef TXD_WITH_BR3
    ok = ok && lds_attr(reinterpret_cast<const void *>(k_blind_rotate3), LDS_BYTES_BR3);
#endif
#endif
    if (!ok) { fail("tx_create: cannot reserve LDS for the blind-rotation / trace kernels"); tx_destroy(c); return nullptr; }
    return c;
}

static void free_ws(tx_ctx *c)
{
    hipFree(c->st); hipFree(c->sm); hipFree(c->small); hipFree(c->acc); hipFree(c->glev); hipFree(c->res);
    hipFree(c->ggsw_f); hipFree(c->d_bytes); hipFree(c->d_ct);
    c->st = c->sm = c->small = c->acc = c->glev = c->res = nullptr; c->ggsw_f = nullptr; c->d_bytes = c->d_ct = nullptr;
    hipFree(c->share_tmp); hipFree(c->d_idx); hipFree(c->d_pos); hipFree(c->res2);
    c->share_tmp = nullptr; c->d_idx = nullptr; c->d_pos = nullptr; c->res2 = nullptr; c->share_tmp_bits = 0; c->idx_cap = 0;
    c->ws_blocks = 0; c->glev_bits = 0;
}

extern "C" void tx_destroy(tx_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 2; i++) { if (c->ev_done[i]) hipEventDestroy(c->ev_done[i]); if (c->ev_copied[i]) hipEventDestroy(c->ev_copied[i]); }
    free_ws(c);
    hipFree(c->bsk_f); hipFree(c->ksk_f); hipFree(c->auto_f); hipFree(c->ss_f);
    hipFree(c->tw); hipFree(c->tw256); hipFree(c->fair_tab); hipFree(c->luts); hipFree(c->ctr_luts);
    if (c->br8_fault) hipHostFree(c->br8_fault);
    for (auto &e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// ---- sizes ----
static size_t bsk_words(const tx_params &p) { return (size_t)p.n * PBS_L * KP * KP * N; }
static size_t ksk_words(const tx_params &p) { return (size_t)(2 * N / KS_N) * KS_L * (p.n / KS_N + 1) * KS_N; }
static size_t auto_words() { return (size_t)LOGN * 2 * AUTO_L * KP * N; }
static size_t ss_words() { return (size_t)2 * SS_L * KP * KP * N; }
static size_t lut_words() { return (size_t)(9 * 4 + 1) * 16 * 2 * GL; }
static size_t ctr_lut_words() { return (size_t)10 * 3 * 16 * 2 * GL; }

static inline int grid_for(size_t units) { return (int)((units + WAVES - 1) / WAVES); }

// ---- timing helpers: HIP events on the launch stream ----
// A failed event call never stops a launch sequence half way (the kernels between ev_begin and ev_end still run): it is
// remembered in c->ev_error, no further events are recorded for the call, and timing_collect reports the call's timing as
// failed (tx_timing.timing_failed, all times zero) -- the call itself, whose results are complete, still returns 0.
static void ev_begin(tx_ctx *c, int fam, const char *kernel = "", size_t units = 0)
{
    if (!c->time_enabled || c->ev_error) return;
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess) { c->ev_error = "hipEventCreate"; return; }
        if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); c->ev_error = "hipEventCreate"; return; }
        c->ev_pool.push_back({a, b});
    }
    auto &e = c->ev_pool[c->ev_used];
    if (hipEventRecord(e.first, c->stream) != hipSuccess) { c->ev_error = "hipEventRecord"; return; }
    c->evs.push_back({fam, kernel, units, e.first, e.second});
}
static void ev_end(tx_ctx *c)
{
    if (!c->time_enabled || c->ev_error) return;
    if (c->evs.size() != c->ev_used + 1) return;            // the matching ev_begin recorded nothing
    if (hipEventRecord(c->ev_pool[c->ev_used].second, c->stream) != hipSuccess) { c->ev_error = "hipEventRecord"; c->evs.pop_back(); return; }
    c->ev_used++;
}
static void timing_reset(tx_ctx *c)
{
    c->evs.clear(); c->ev_used = 0; memset(&c->timing, 0, sizeof c->timing);
    c->br_use.clear(); c->ev_error = nullptr;       // (last_br_kernel keeps naming the last blind rotation that ran)
}
static int timing_collect(tx_ctx *c)
{
    // every entry point ends here, behind its last synchronisation: a wave of k_blind_rotate8 that gave up waiting at a software
    // barrier (2^18 polls: a fault, never seen) has made the launch's results invalid -- the call fails, loudly
    if (c->br8_fault && *c->br8_fault) {
        *c->br8_fault = 0;
        return fail("k_blind_rotate8: a wave gave up waiting at a software barrier; the results of this call are invalid");
    }
    // what ran is taken from the launches themselves (launch_br), so it is right without events too
    tx_timing &t = c->timing;
    t.units_br_main = 0; t.n_br_main = 0; t.units_blind_rotate = 0;
    for (auto &u : c->br_use) {
        t.units_blind_rotate += u.units;
        if (u.units > t.units_br_main) { t.units_br_main = u.units; t.n_br_main = u.launches; c->last_br_kernel = u.kernel; }
    }
    if (c->ev_error) {       // the measurement failed, not the path: the results are computed and copied
        t.timing_failed = 1;
        g_err = std::string(c->ev_error) + " failed while timing the call (results are complete; tx_timing.timing_failed is set)";
        return 0;
    }
    if (c->evs.size() != c->ev_used) c->evs.resize(c->ev_used);
    double ms[F_COUNT] = {0}; uint32_t cnt[F_COUNT] = {0};
    // the blind rotation by kernel: a call may use more than one (launch-size dispatch, split tails)
    struct G { const char *k; double ms; uint32_t n; uint64_t units; };
    std::vector<G> br;
    for (auto &e : c->evs) {
        float t = 0;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) {
            c->timing.timing_failed = 1;
            g_err = "hipEventElapsedTime failed while timing the call (results are complete; tx_timing.timing_failed is set)";
            return 0;
        }
        ms[e.fam] += t; cnt[e.fam]++;
        if (e.fam == F_BR) {
            size_t g = 0;
            while (g < br.size() && strcmp(br[g].k, e.kernel) != 0) g++;
            if (g == br.size()) br.push_back({e.kernel, 0.0, 0, 0});
            br[g].ms += t; br[g].n++; br[g].units += e.units;
        }
    }
    t.ms_keyswitch = ms[F_KS]; t.ms_blind_rotate = ms[F_BR]; t.ms_trace = ms[F_TRACE];
    t.ms_scheme_switch = ms[F_SS]; t.ms_lut_eval = ms[F_LUT]; t.ms_linear = ms[F_LIN];
    t.n_keyswitch = cnt[F_KS]; t.n_blind_rotate = cnt[F_BR]; t.n_trace = cnt[F_TRACE];
    t.n_scheme_switch = cnt[F_SS]; t.n_lut_eval = cnt[F_LUT]; t.n_linear = cnt[F_LIN];
    t.ms_total = 0; for (int i = 0; i < F_COUNT; i++) t.ms_total += ms[i];
    t.ms_br_main = 0;
    for (auto &g : br)
        if (t.units_br_main && strcmp(g.k, c->last_br_kernel) == 0) t.ms_br_main = g.ms;
    return 0;
}
extern "C" size_t tx_timing_size(void) { return sizeof(tx_timing); }
extern "C" int tx_get_timing(tx_ctx *c, tx_timing *t) { if (!c || !t) return fail("tx_get_timing: null"); *t = c->timing; return 0; }
extern "C" const char *tx_last_blind_rotate_kernel(const tx_ctx *c) { return c ? c->last_br_kernel : ""; }
extern "C" int tx_get_twiddles(tx_ctx *c, double *out)
{
    if (!c || !out) return fail("tx_get_twiddles: null");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpy(out, c->tw, TW_ROWS * 64 * sizeof(cplx), hipMemcpyDeviceToHost));
    return 0;
}

#ifdef TXD_STAMPS
// diagnostic builds only: the fairness table of k_blind_rotate6 (word 4 of an entry counts the workgroups that CU has hosted)
extern "C" int tx_debug_fair_table(tx_ctx *c, uint32_t *out)
{
    if (!c || !out) return fail("tx_debug_fair_table: null");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpy(out, c->fair_tab, (size_t)FAIR_TAB_ENTRIES * FAIR_TAB_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int tx_debug_wave_times(tx_ctx *c, unsigned long long *out, int n)
{
    if (!c || !out || n > 2 * 4096) return fail("tx_debug_wave_times: bad argument");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(txd::txd_wave_rt), (size_t)n * sizeof(unsigned long long)));
    return 0;
}
// diagnostic builds only: per-phase cycle sums of one wave of the last k_blind_rotate launch (tools/br_only.py)
extern "C" int tx_debug_stamps(tx_ctx *c, unsigned long long *out16)
{
    if (!c || !out16) return fail("tx_debug_stamps: null");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(txd::txd_stamps), 16 * sizeof(unsigned long long)));
    return 0;
}
#endif

// ---- device memory for the *_is_device arguments (callers without a HIP binding of their own: the reference's Rust host) ----
extern "C" void *tx_device_alloc(tx_ctx *c, size_t bytes)
{
    if (!c) { fail("tx_device_alloc: null context"); return nullptr; }
    void *p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) {
        fail("tx_device_alloc: hipMalloc of " + std::to_string(bytes) + " bytes failed"); return nullptr;
    }
    return p;
}
extern "C" int tx_device_free(tx_ctx *c, void *p)
{
    if (!c) return fail("tx_device_free: null context");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipFree(p));
    return 0;
}
extern "C" int tx_device_upload(tx_ctx *c, void *dst_device, const void *src_host, size_t bytes)
{
    if (!c || (bytes && (!dst_device || !src_host))) return fail("tx_device_upload: null argument");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int tx_device_download(tx_ctx *c, void *dst_host, const void *src_device, size_t bytes)
{
    if (!c || (bytes && (!dst_host || !src_device))) return fail("tx_device_download: null argument");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---- pinned host memory (no context: usable by every device, from any thread) ----
extern "C" void *tx_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocPortable) != hipSuccess) {
        fail("tx_host_alloc: hipHostMalloc of " + std::to_string(bytes) + " bytes failed"); return nullptr;
    }
    return p;
}
extern "C" int tx_host_free(void *p)
{
    if (!p) return 0;
    HIP_OK(hipHostFree(p));
    return 0;
}
extern "C" int tx_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return fail("tx_host_register: null argument");
    HIP_OK(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return 0;
}
extern "C" int tx_host_unregister(void *p)
{
    if (!p) return 0;
    HIP_OK(hipHostUnregister(p));
    return 0;
}
extern "C" int tx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return fail("tx_device_count: hipGetDeviceCount failed (no HIP device, and no CPU fallback)");
    return n;
}

// device buffer released on every exit path
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : -1; }
};

// ---- key loading (SRV_AES:640-687) ----
// Uploads and transforms are enqueued on the context's stream and NOT waited for here: tx_load_eval_keys synchronises once at
// its end, and allocates in between (staging buffers are released after that synchronisation).
static int fwd_polys(tx_ctx *c, const uint64_t *h_src, size_t words, cplx *d_dst, int mode, DevBuf &buf)
{
    if (buf.alloc(words * 8)) return fail("fwd_polys: hipMalloc failed");
    uint64_t *tmp = (uint64_t *)buf.p;
    HIP_OK(hipMemcpyAsync(tmp, h_src, words * 8, hipMemcpyHostToDevice, c->stream));
    int npoly = (int)(words / N);
    hipLaunchKernelGGL(k_fwd_torus, dim3(grid_for(npoly)), dim3(256), LDS_BYTES, c->stream, tmp, d_dst, npoly, mode,
                       (int)AUTO_SPLIT, c->tw);
    if (mode == 1) {   // both halves of the split key from one upload
        hipLaunchKernelGGL(k_fwd_torus, dim3(grid_for(npoly)), dim3(256), LDS_BYTES, c->stream, tmp,
                           d_dst + c->auto_split_stride, npoly, 2, (int)AUTO_SPLIT, c->tw);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int tx_eval_key_words(const tx_ctx *c, size_t out4[4])
{
    if (!c || !out4) return fail("tx_eval_key_words: null argument");
    out4[0] = bsk_words(c->p); out4[1] = ksk_words(c->p); out4[2] = auto_words(); out4[3] = ss_words();
    return 0;
}

static int ensure_ws(tx_ctx *c, size_t nb);

extern "C" int tx_load_eval_keys(tx_ctx *c, const uint64_t *bsk, size_t n_bsk, const uint64_t *ksk, size_t n_ksk,
                                 const uint64_t *auto_keys, size_t n_auto, const uint64_t *ss_key, size_t n_ss)
{
    if (!c || !bsk || !ksk || !auto_keys || !ss_key) return fail("tx_load_eval_keys: null argument");
    const size_t nb = bsk_words(c->p), nk = ksk_words(c->p), na = auto_words(), ns = ss_words();
    if (n_bsk != nb || n_ksk != nk || n_auto != na || n_ss != ns)       // nothing is read from a buffer of the wrong length
        return fail("tx_load_eval_keys: key length mismatch (bsk " + std::to_string(n_bsk) + "/" + std::to_string(nb) +
                    ", ksk " + std::to_string(n_ksk) + "/" + std::to_string(nk) + ", auto " + std::to_string(n_auto) + "/" +
                    std::to_string(na) + ", ss " + std::to_string(n_ss) + "/" + std::to_string(ns) +
                    " words given/expected for n = " + std::to_string(c->p.n) + ")");
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    hipFree(c->bsk_f); hipFree(c->ksk_f); hipFree(c->auto_f); hipFree(c->ss_f);
    c->bsk_f = c->ksk_f = c->auto_f = c->ss_f = nullptr; c->have_keys = false;
    HIP_OK(hipMalloc(&c->bsk_f, nb * 8));
    HIP_OK(hipMalloc(&c->ksk_f, nk * 8));
    HIP_OK(hipMalloc(&c->auto_f, na * 8 * 2));
    HIP_OK(hipMalloc(&c->ss_f, ns * 8));
    c->auto_split_stride = na / 2;                                  // complex units per split
    DevBuf s_bsk, s_auto, s_ss, s_ksk;                              // staging, alive until the synchronisation below
    int rc = fwd_polys(c, bsk, nb, c->bsk_f, 0, s_bsk);
    if (!rc) rc = fwd_polys(c, auto_keys, na, c->auto_f, 1, s_auto);
    if (!rc) rc = fwd_polys(c, ss_key, ns, c->ss_f, 0, s_ss);
    if (!rc) {
        if (s_ksk.alloc(nk * 8)) rc = fail("tx_load_eval_keys: hipMalloc failed");
        else {
            uint64_t *tmp = (uint64_t *)s_ksk.p;
            if (hipMemcpyAsync(tmp, ksk, nk * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail("tx_load_eval_keys: upload failed");
            else {
                int npoly = (int)(nk / KS_N);
                hipLaunchKernelGGL(k_fwd_torus256, dim3(grid_for(npoly)), dim3(256), 0, c->stream, tmp, c->ksk_f, npoly, c->tw256);
                if (hipGetLastError() != hipSuccess) rc = fail("tx_load_eval_keys: launch failed");
            }
        }
    }
    // a reservation made before the keys arrived (tx_reserve) is filled while the transforms run
    if (!rc && c->reserve_blocks > c->ws_blocks) rc = ensure_ws(c, c->reserve_blocks);
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail("tx_load_eval_keys: hipStreamSynchronize failed");
    if (rc) return -1;
    c->have_keys = true;
    return 0;
}

static int load_blob(tx_ctx *c, uint64_t **dst, const uint64_t *src, size_t words, size_t expect, const char *what)
{
    if (!c || !src) return fail(std::string(what) + ": null argument");
    if (words != expect) return fail(std::string(what) + ": wrong size");
    HIP_OK(hipSetDevice(c->device));
    hipFree(*dst); *dst = nullptr;
    HIP_OK(hipMalloc(dst, words * 8));
    HIP_OK(hipMemcpy(*dst, src, words * 8, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int tx_load_round_luts(tx_ctx *c, const uint64_t *luts, size_t words)
{ return load_blob(c, c ? &c->luts : nullptr, luts, words, lut_words(), "tx_load_round_luts"); }
extern "C" int tx_load_ctr_luts(tx_ctx *c, const uint64_t *luts, size_t words)
{ return load_blob(c, c ? &c->ctr_luts : nullptr, luts, words, ctr_lut_words(), "tx_load_ctr_luts"); }

// ---- workspace ----
extern "C" int tx_reserve(tx_ctx *c, size_t n_blocks)
{
    if (!c) return fail("tx_reserve: null context");
    c->reserve_blocks = std::min(n_blocks, CHUNK_BLOCKS);
    if (!c->have_keys || c->reserve_blocks <= c->ws_blocks) return 0;      // without keys: allocated under their transforms
    HIP_OK(hipSetDevice(c->device));
    return ensure_ws(c, c->reserve_blocks);
}

static int ensure_ws(tx_ctx *c, size_t nb)
{
    if (nb <= c->ws_blocks) return 0;
    free_ws(c);
    const size_t bits = nb * 128;
    HIP_OK(hipMalloc(&c->st, bits * BIG * 8));
    HIP_OK(hipMalloc(&c->res, bits * BIG * 8));
    HIP_OK(hipMalloc(&c->sm, 4 * bits * BIG * 8));
    HIP_OK(hipMalloc(&c->small, bits * (c->p.n + 1) * 8));
    HIP_OK(hipMalloc(&c->acc, bits * GL * 8));
    // (no GLev buffer: the fused kernel keeps the GLev element in LDS; run_cbs allocates one for the separate kernels)
    HIP_OK(hipMalloc(&c->ggsw_f, bits * CBS_L * KP * KP * H * sizeof(cplx)));
    HIP_OK(hipMalloc(&c->d_bytes, nb * 16));
    HIP_OK(hipMalloc(&c->d_ct, nb * 16));
    c->ws_blocks = nb;
    return 0;
}

// ---- kernel launch wrappers (device pointers) ----
static void run_ks(tx_ctx *c, const uint64_t *in, size_t count, uint64_t *out)
{
    ev_begin(c, F_KS);
    hipLaunchKernelGGL(k_lwe_keyswitch, dim3(grid_for(count)), dim3(256), 0, c->stream, in, out, (int)count, (int)c->p.n,
                       c->ksk_f, c->tw256, c->tie_on);
    ev_end(c);
}
// which kernel(s) a blind rotation of `count` units runs on a chip of `cus` CUs (host logic, no GPU; the rule is explained
// in run_br): at most two launches, in unit order.  forced >= 0: one launch of that kernel.
static int br_plan(size_t count, int cus_, int forced, int variants[2], size_t counts[2])
{
    if (count == 0) return 0;
    const size_t cus = (size_t)(cus_ > 0 ? cus_ : 1);
    auto small = [&](size_t k) { return k <= 3 * cus ? 6 : 8; };
    if (forced >= 0) { variants[0] = forced; counts[0] = count; return 1; }
    if (count <= 4 * cus) { variants[0] = small(count); counts[0] = count; return 1; }
    const size_t pass = 8 * cus, rem = count % pass, full = count - rem;
    if (full == 0 || rem == 0 || rem > 4 * cus) { variants[0] = 5; counts[0] = count; return 1; }
    variants[0] = 5; counts[0] = full;
    variants[1] = small(rem); counts[1] = rem;
    return 2;
}
extern "C" int tx_plan_blind_rotate(size_t count, int num_cus, int *variants, size_t *counts)
{
    if (!variants || !counts || num_cus <= 0) return fail("tx_plan_blind_rotate: bad argument");
    return br_plan(count, num_cus, -1, variants, counts);
}

static void br_used(tx_ctx *c, const char *kernel, size_t units)
{
    for (auto &u : c->br_use) if (strcmp(u.kernel, kernel) == 0) { u.units += units; u.launches++; return; }
    c->br_use.push_back({kernel, (uint64_t)units, 1u});
}
static void launch_br(tx_ctx *c, int variant, const uint64_t *in, size_t count, uint64_t *acc)
{
    const int n = (int)c->p.n;
#ifdef TXD_EXPERIMENTS
    br_used(c, variant == 8 ? "k_blind_rotate8" : variant == 7 ? "k_blind_rotate7" : variant == 6 ? "k_blind_rotate6" : variant == 5 ? "k_blind_rotate5" :
               variant == 4 ? "k_blind_rotate4" : variant == 3 ? "k_blind_rotate3" : "k_blind_rotate", count);
#else
    br_used(c, variant == 8 ? "k_blind_rotate8" : variant == 6 ? "k_blind_rotate6" : "k_blind_rotate5", count);
#endif
    if (variant == 8) {
        ev_begin(c, F_BR, "k_blind_rotate8", count);
        hipLaunchKernelGGL(k_blind_rotate8, dim3((count + BR7_UNITS - 1) / BR7_UNITS), dim3(768), LDS_BYTES_BR7, c->stream, in, acc, (int)count, n,
                           c->bsk_f, c->tw, c->tie_on, c->br8_fault);
    } else if (variant == 6) {
        // launch number 0 = no arbitration: with at most one workgroup per CU there is nobody to be fair to
        uint32_t ep = count <= (size_t)c->num_cus ? 0u : (++c->br6_epoch & 0x7FFFu);          // 15 bits above the 16-bit step count of a table word
        if (ep == 0 && count > (size_t)c->num_cus) {                                   // wrapped: forget every word, so that none of 32 768 launches ago reads as current
            hipMemsetAsync(c->fair_tab, 0, (size_t)FAIR_TAB_ENTRIES * FAIR_TAB_WORDS * sizeof(uint32_t), c->stream);
            ep = ++c->br6_epoch & 0x7FFFu;
        }
        ev_begin(c, F_BR, "k_blind_rotate6", count);
        hipLaunchKernelGGL(k_blind_rotate6, dim3(count), dim3(192), LDS_BYTES_BR6, c->stream, in, acc, (int)count, n, c->bsk_f, c->tw,
                           c->fair_tab, ep << 16, c->tie_on);
    } else if (variant == 5) {
        size_t wgs = std::min<size_t>(count, (size_t)c->num_cus);     // persistent, one per CU; units are dealt wave-major
        if (c->br5_wgs > 0) wgs = std::min<size_t>(wgs, (size_t)c->br5_wgs);                                       // experiments only
        ev_begin(c, F_BR, "k_blind_rotate5", count);
        hipLaunchKernelGGL(k_blind_rotate5, dim3(wgs), dim3(512), LDS_BYTES_BR5, c->stream, in, acc, (int)count, n, c->bsk_f, c->tw, c->tie_on);
#ifdef TXD_EXPERIMENTS
    } else if (variant == 7) {
        ev_begin(c, F_BR, "k_blind_rotate7", count);
        hipLaunchKernelGGL(k_blind_rotate7, dim3((count + BR7_UNITS - 1) / BR7_UNITS), dim3(768), LDS_BYTES_BR7, c->stream, in, acc, (int)count, n,
                           c->bsk_f, c->tw, c->fair_tab, c->tie_on);
    } else if (variant == 4) {
        ev_begin(c, F_BR, "k_blind_rotate4", count);
        hipLaunchKernelGGL(k_blind_rotate4, dim3(grid_for(count)), dim3(256), LDS_BYTES_BR3, c->stream, in, acc, (int)count, n, c->bsk_f, c->tw, c->tie_on);
#ifdef TXD_WITH_BR3
    } else if (variant == 3) {
        ev_begin(c, F_BR, "k_blind_rotate3", count);
        hipLaunchKernelGGL(k_blind_rotate3, dim3(grid_for(count)), dim3(256), LDS_BYTES_BR3, c->stream, in, acc, (int)count, n, c->bsk_f, c->tw);
#endif
    } else {
        ev_begin(c, F_BR, "k_blind_rotate", count);
        hipLaunchKernelGGL(k_blind_rotate, dim3(grid_for(count)), dim3(256), LDS_BYTES_BR, c->stream, in, acc, (int)count, n, c->bsk_f, c->tw);
    }
#else
    }
#endif
    ev_end(c);
}
static void run_br(tx_ctx *c, const uint64_t *in, size_t count, uint64_t *acc)
{
    // Same results bit for bit from every kernel; the choice is by launch size, in units per CU (measured on MI355X, n = 768,
    // inside the round pipeline, i.e. under sustained load):
    //   > 4   every SIMD can hold two units: k_blind_rotate5 (generated assembly, two waves per SIMD in one persistent
    //         512-thread workgroup per CU, fair arbitration between the two waves of a SIMD); k_blind_rotate4 (the same loop
    //         in two independent 256-thread workgroups per CU) is 1 % behind it (TX_BR_VARIANT=4).
    //   <= 3  the launch time is the length of ONE CMux chain: k_blind_rotate6 puts three waves on every unit, one per
    //         polynomial (tx_br2.hpp), and (round 4) keeps the workgroups of a CU level through a table of step counters and
    //         s_setprio: 128 and 256 units 2.9 against 7.1 ms for the one-wave kernel, 512 units 4.0 (4.5 without the
    //         fairness), 768 units 4.9 (5.8 without).  One workgroup per CU has nobody to be fair to (launch number 0).
    //   3..4  (the 8-blocks-per-GPU share of the medium instance: 1024 units) k_blind_rotate8: the same loop with the four units
    //         of a CU in ONE twelve-wave workgroup -- four independent three-wave workgroups need a packing of 168-register
    //         waves the hardware's placement misses for a handful of workgroups per launch (9 ms instead of 6.5) -- and, since
    //         round 5, software barriers per unit instead of the workgroup's s_barrier (k_blind_rotate7: lockstep): 5.9 ms
    //         against 6.4 for k_blind_rotate7 and 7.5 for k_blind_rotate (one wave per unit, hipcc), inside the pipeline.
    // A large launch that does not fill its last pass of 8 units per CU is split: whole passes go to k_blind_rotate5, and a
    // remainder of at most 4 units per CU to the small-launch kernels behind it (a persistent wave would otherwise take one
    // more 10.7 ms unit for it: 2144 units = 21.4 ms unsplit, 10.7 + 2.9 ms split).
    size_t counts[2]; int variants[2];
    const int n_launch = br_plan(count, c->num_cus, c->br_variant, variants, counts);
#ifdef TXD_SYNC_BEFORE_BR      // experiment: does the blind rotation start on a chip that is really empty?
    hipStreamSynchronize(c->stream);
#endif
    size_t done = 0;
    for (int i = 0; i < n_launch; i++) {
        launch_br(c, variants[i], in + done * (c->p.n + 1), counts[i], acc + done * GL);
        done += counts[i];
    }
}
static void run_extract_trace(tx_ctx *c, const uint64_t *acc, size_t count, uint64_t *glev)
{
    ev_begin(c, F_TRACE);
#ifdef TXD_EXPERIMENTS
    if (c->trace_variant != 4)
        hipLaunchKernelGGL(k_extract_trace, dim3(grid_for(count * CBS_L)), dim3(256), LDS_BYTES_TRACE, c->stream, acc, glev,
                           (int)count, c->auto_f, c->auto_split_stride, c->tw);
    else
#endif
        hipLaunchKernelGGL(k_extract_trace4, dim3(grid_for(count * CBS_L)), dim3(256), LDS_BYTES_TR4, c->stream, acc, glev,
                           (int)count, c->auto_f, c->auto_split_stride, c->tw, c->tie_on);
    ev_end(c);
}
static void run_ss(tx_ctx *c, const uint64_t *glev, size_t count, cplx *ggsw_f)
{
    ev_begin(c, F_SS);
#ifdef TXD_EXPERIMENTS
    if (c->ss_variant != 4)
        hipLaunchKernelGGL(k_scheme_switch, dim3(grid_for(count * CBS_L)), dim3(256), LDS_BYTES, c->stream, glev, ggsw_f,
                           (int)(count * CBS_L), c->ss_f, c->tw);
    else
#endif
        hipLaunchKernelGGL(k_scheme_switch4, dim3(grid_for(count * CBS_L)), dim3(256), LDS_BYTES_TR4, c->stream, glev, ggsw_f,
                           (int)(count * CBS_L), c->ss_f, c->tw, c->tie_on);
    ev_end(c);
}
static void run_lut(tx_ctx *c, const cplx *ggsw_f, size_t nbytes, const uint64_t *lut_set, size_t lut_stride, int nlut,
                    uint64_t *out, const uint8_t *pos = nullptr)
{
    ev_begin(c, F_LUT);
    const dim3 grid((grid_for(nbytes * nlut * 2) + 7) / 8 * 8);      // a multiple of 8: the kernels deal the units out XCD by XCD
#ifdef TXD_EXPERIMENTS
    if (c->lut_variant == 4)
        hipLaunchKernelGGL(k_lut_eval4, grid, dim3(256), LDS_BYTES_TR4, c->stream, ggsw_f, lut_set,
                           lut_stride, (size_t)2 * GL, out, (int)nbytes, nlut, c->tw, pos);
    else
#endif
        hipLaunchKernelGGL(k_lut_eval, grid, dim3(256), LDS_BYTES, c->stream, ggsw_f, lut_set,
                           lut_stride, (size_t)2 * GL, out, (int)nbytes, nlut, c->tw, pos, c->tie_on);
    ev_end(c);
}
// KS + CBS of `count` big LWEs into Fourier GGSWs
static int run_cbs(tx_ctx *c, const uint64_t *lwe_big, size_t count, cplx *ggsw_f)
{
    run_ks(c, lwe_big, count, c->small);
    run_br(c, c->small, count, c->acc);
    if (c->trace_variant == 4 && c->ss_variant == 4) {      // fused: the GLev element never leaves the workgroup's LDS
        ev_begin(c, F_TRACE);                                // (reported under ms_trace; ms_scheme_switch stays 0)
        const int wgs = std::min(grid_for(count * CBS_L), c->num_cus * c->trss_wgs_per_cu);
        hipLaunchKernelGGL(k_trace_scheme_switch4, dim3(wgs), dim3(256), LDS_BYTES_TR4, c->stream, c->acc, ggsw_f,
                           (int)count, c->auto_f, c->auto_split_stride, c->ss_f, c->tw, c->tie_on);
        ev_end(c);
        return 0;
    }
    if (c->glev_bits < count) {                              // the separate kernels (TX_TRACE_VARIANT / TX_SS_VARIANT = 0) pass the
        HIP_OK(hipStreamSynchronize(c->stream));             // GLev through HBM: 172 KiB per bit, allocated only here
        hipFree(c->glev); c->glev = nullptr; c->glev_bits = 0;
        HIP_OK(hipMalloc(&c->glev, std::max(count, c->ws_blocks * 128) * CBS_L * GL * 8));
        c->glev_bits = std::max(count, c->ws_blocks * 128);
    }
    run_extract_trace(c, c->acc, count, c->glev);
    run_ss(c, c->glev, count, ggsw_f);
    return 0;
}
static inline int ew_grid(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 4096 ? 4096 : g); }

// ---- CTR: which state columns of the second round hold the same ciphertexts (host logic, no GPU) ----
// ctr: nb counter blocks of 16 bytes.  Column col of block b entering the second round is built (k_linear, forward:
// sh[r, col] = s[r, (r + col) % 4]) from the first-round table entries of counter bytes 4 ((r + col) % 4) + r, r = 0..3.
// rep_of[b * 4 + col] = index into col_of of the first (block, column) with the same column index and the same four bytes.
static void ctr_round2_groups(const uint8_t *ctr, size_t nb, std::vector<int> &rep_of, std::vector<int> &col_of)
{
    std::unordered_map<uint64_t, int> seen;                  // (column index, its four counter bytes) -> distinct column
    rep_of.assign(nb * 4, 0); col_of.clear();
    for (size_t b = 0; b < nb; b++)
        for (int col = 0; col < 4; col++) {
            uint64_t key = (uint64_t)col;
            for (int r = 0; r < 4; r++) key = (key << 8) | ctr[b * 16 + 4 * ((r + col) & 3) + r];
            auto it = seen.find(key);
            int idx;
            if (it == seen.end()) { idx = (int)col_of.size(); seen.emplace(key, idx); col_of.push_back((int)(b * 4 + col)); }
            else idx = it->second;
            rep_of[b * 4 + col] = idx;
        }
}
static void ctr_blocks(const uint8_t *iv, size_t blk0, size_t nb, std::vector<uint8_t> &ctr)
{   // iv + (blk0 + b), 128-bit big-endian (harness/aes_keygen_and_encrypt.py:52-54)
    ctr.resize(nb * 16);
    for (size_t b = 0; b < nb; b++) {
        uint8_t *p = &ctr[b * 16]; memcpy(p, iv, 16);
        uint64_t add = blk0 + b;
        for (int i = 15; i >= 0 && add; i--) { uint64_t v = (uint64_t)p[i] + (add & 0xFF); p[i] = (uint8_t)v; add = (add >> 8) + (v >> 8); }
    }
}
extern "C" int tx_ctr_shared_columns(const uint8_t *iv, size_t first_block, size_t n_blocks, int *rep_of, int *col_of)
{
    if (!iv || !rep_of || !col_of) return fail("tx_ctr_shared_columns: null argument");
    if (n_blocks > (size_t)INT_MAX / 4) return fail("tx_ctr_shared_columns: n_blocks * 4 does not fit the int indices of rep_of / col_of");
    std::vector<uint8_t> ctr; std::vector<int> rep, cols;
    ctr_blocks(iv, first_block, n_blocks, ctr);
    ctr_round2_groups(ctr.data(), n_blocks, rep, cols);
    std::copy(rep.begin(), rep.end(), rep_of);
    std::copy(cols.begin(), cols.end(), col_of);
    return (int)cols.size();
}

// ---- a1: round schedules ----
// `spare`: a buffer of nb x 128 big LWEs that is free until the chunk's last kernel (the result buffer the chunk will be
// finalised into qualifies: nothing reads it before); the CTR schedule parks the compact state of its shared round there
static int transcipher_chunk(tx_ctx *c, const uint8_t *ct, size_t nb, const uint8_t *iv, size_t blk0, int rounds_done,
                             uint64_t *d_out, bool reference_order, uint64_t *spare)
{
    const size_t bits = nb * 128, blkw = (size_t)128 * BIG;
    const size_t mult_sz = (size_t)16 * 2 * GL;
    if (!iv) {
        if (!c->luts) return fail("tx_transcipher: round LUTs not loaded");
        const size_t set_sz = 4 * mult_sz;
        // SRV_AES:89-91: clear inv_shift_rows on the public ciphertext bytes
        std::vector<uint8_t> t(nb * 16);
        for (size_t b = 0; b < nb; b++)
            for (int r = 0; r < 4; r++)
                for (int col = 0; col < 4; col++) t[b * 16 + 4 * col + r] = ct[b * 16 + 4 * ((4 + col - r) % 4) + r];
        HIP_OK(hipMemcpyAsync(c->d_bytes, t.data(), nb * 16, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));   // t goes out of scope
        ev_begin(c, F_LIN);
        for (int m = 0; m < 4; m++)                // SRV_AES:99-118
            hipLaunchKernelGGL(k_known_rotate, dim3(grid_for(bits)), dim3(256), 0, c->stream, c->d_bytes,
                               c->luts + (size_t)m * mult_sz, (size_t)2 * GL, c->sm + (size_t)m * nb * blkw, (int)bits);
        hipLaunchKernelGGL(k_linear, dim3(ew_grid(bits * BIG)), dim3(256), 0, c->stream, c->sm, c->st, (int)nb, 1);
        ev_end(c);
        for (int s = 1; s <= 8 && s <= rounds_done; s++) {      // SRV_AES:130-163
            if (run_cbs(c, c->st, bits, c->ggsw_f)) return -1;
            run_lut(c, c->ggsw_f, nb * 16, c->luts + (size_t)s * set_sz, mult_sz, 4, c->sm);
            ev_begin(c, F_LIN);
            hipLaunchKernelGGL(k_linear, dim3(ew_grid(bits * BIG)), dim3(256), 0, c->stream, c->sm, c->st, (int)nb, 1);
            ev_end(c);
        }
        const uint64_t *fin = c->st;
        if (rounds_done >= 9) {                                  // SRV_AES:166-180
            if (run_cbs(c, c->st, bits, c->ggsw_f)) return -1;
            run_lut(c, c->ggsw_f, nb * 16, c->luts + (size_t)9 * set_sz, 0, 1, c->sm);
            fin = c->sm;
        }
        if (reference_order) {
            HIP_OK(hipMemcpyAsync(d_out, fin, bits * BIG * 8, hipMemcpyDeviceToDevice, c->stream));
        } else {
            ev_begin(c, F_LIN);
            hipLaunchKernelGGL(k_finalize, dim3(ew_grid(bits * BIG)), dim3(256), 0, c->stream, fin, d_out, (int)nb, 0,
                               (const uint8_t *)nullptr);
            ev_end(c);
        }
    } else {
        if (!c->ctr_luts) return fail("tx_transcipher: CTR LUTs not loaded");
        const size_t set_sz = 3 * mult_sz;
        std::vector<uint8_t> ctr;
        ctr_blocks(iv, blk0, nb, ctr);
        HIP_OK(hipMemcpyAsync(c->d_bytes, ctr.data(), nb * 16, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(c->d_ct, ct, nb * 16, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        ev_begin(c, F_LIN);
        for (int m = 0; m < 3; m++)
            hipLaunchKernelGGL(k_known_rotate, dim3(grid_for(bits)), dim3(256), 0, c->stream, c->d_bytes,
                               c->ctr_luts + (size_t)m * mult_sz, (size_t)2 * GL, c->sm + (size_t)m * nb * blkw, (int)bits);
        ev_end(c);
        // Second round of a CTR batch: the state entering it is a public function of the counter blocks up to the first
        // linear layer (the first round is a sample extraction at a PUBLIC index), so two blocks whose counters agree in
        // the four bytes that feed a column hold the SAME 32 ciphertexts in that column -- bit for bit, the operations are
        // deterministic -- and the circuit bootstraps and table look-ups of that column give the same result.  Counters of
        // one batch differ in their last byte(s) only: of the 4 nb columns, 3 are shared by all blocks (until a carry into
        // byte 14), so the round bootstraps each DISTINCT column once and copies (medium instance: 268 instead of 1024
        // bytes).  From the third round on every column has mixed in a block's own byte.  Same ciphertexts as evaluating
        // every column: tests compare with the oracle, which does; TX_CTR_SHARE=0 evaluates every column here too.
        std::vector<int> col_of, rep_of(nb * 4);
        if (c->ctr_share) ctr_round2_groups(ctr.data(), nb, rep_of, col_of);
        const size_t ucols = col_of.size();
        const bool share = c->ctr_share && ucols < nb * 4;
        if (share) {
            if (c->share_tmp_bits < c->ws_blocks * 128) {
                hipFree(c->share_tmp); c->share_tmp = nullptr; c->share_tmp_bits = 0;
                HIP_OK(hipMalloc(&c->share_tmp, 3 * c->ws_blocks * 128 * BIG * 8));
                c->share_tmp_bits = c->ws_blocks * 128;
            }
            if (c->idx_cap < nb * 4) {
                hipFree(c->d_idx); hipFree(c->d_pos); c->d_idx = nullptr; c->d_pos = nullptr; c->idx_cap = 0;
                HIP_OK(hipMalloc(&c->d_idx, 2 * c->ws_blocks * 4 * sizeof(int)));
                HIP_OK(hipMalloc(&c->d_pos, c->ws_blocks * 16));
                c->idx_cap = c->ws_blocks * 4;
            }
            std::vector<uint8_t> pos(ucols * 4);
            for (size_t u = 0; u < ucols; u++)
                for (int r = 0; r < 4; r++) pos[u * 4 + r] = (uint8_t)(4 * (col_of[u] & 3) + r);
            HIP_OK(hipMemcpyAsync(c->d_idx, col_of.data(), ucols * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(c->d_idx + c->idx_cap, rep_of.data(), nb * 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(c->d_pos, pos.data(), pos.size(), hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));       // the host vectors go out of scope
        }
        for (int s = 1; s <= 9; s++) {
            ev_begin(c, F_LIN);
            hipLaunchKernelGGL(k_linear, dim3(ew_grid(bits * BIG)), dim3(256), 0, c->stream, c->sm, c->st, (int)nb, 0);
            ev_end(c);
            if (s == 1 && share) {
                uint64_t *cst = spare;                          // compact state: the distinct columns
                ev_begin(c, F_LIN);
                hipLaunchKernelGGL(k_gather_cols, dim3(ew_grid(ucols * 32 * BIG)), dim3(256), 0, c->stream, c->st, c->d_idx, cst, (int)ucols);
                ev_end(c);
                if (run_cbs(c, cst, ucols * 32, c->ggsw_f)) return -1;
                run_lut(c, c->ggsw_f, ucols * 4, c->ctr_luts + (size_t)s * set_sz, mult_sz, 3, c->share_tmp, c->d_pos);
                ev_begin(c, F_LIN);
                hipLaunchKernelGGL(k_scatter_cols, dim3(ew_grid(3 * bits * BIG)), dim3(256), 0, c->stream, c->share_tmp,
                                   c->d_idx + c->idx_cap, c->sm, (int)(nb * 4), (int)ucols, 3);
                ev_end(c);
                continue;
            }
            if (run_cbs(c, c->st, bits, c->ggsw_f)) return -1;
            run_lut(c, c->ggsw_f, nb * 16, c->ctr_luts + (size_t)s * set_sz, mult_sz, s == 9 ? 1 : 3, c->sm);
        }
        ev_begin(c, F_LIN);
        hipLaunchKernelGGL(k_finalize, dim3(ew_grid(bits * BIG)), dim3(256), 0, c->stream, c->sm, d_out, (int)nb, 1, c->d_ct);
        ev_end(c);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// Host output (round 5): the chunks' results alternate between two device buffers; chunk k's LWEs travel to the caller's
// memory on copy_stream while chunk k + 1 computes, and `cb` (may be null) is called on the calling thread for every chunk
// whose LWEs have ARRIVED in `out` -- for chunk k - 1 right after chunk k has been enqueued, so that the caller's work on
// it (writing a file) overlaps the GPU's.  A pinned `out` (tx_host_alloc) makes the copy a DMA at link speed.
static int pipeline_init(tx_ctx *c)
{
    if (c->copy_stream) return 0;
    HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        HIP_OK(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
    }
    return 0;
}

static int transcipher_impl(tx_ctx *c, const uint8_t *ct, size_t n_blocks, const uint8_t *iv, int rounds_done,
                            uint64_t *out, int out_is_device, bool reference_order, tx_chunk_fn cb = nullptr, void *user = nullptr)
{
    if (!c || !ct || !out) return fail("tx_transcipher: null argument");
    if (!c->have_keys) return fail("tx_transcipher: evaluation keys not loaded");
    if (n_blocks == 0) return 0;
    HIP_OK(hipSetDevice(c->device));
    timing_reset(c);
    const size_t chunk = n_blocks < CHUNK_BLOCKS ? n_blocks : CHUNK_BLOCKS;
    if (ensure_ws(c, chunk)) return -1;
    if (out_is_device) {
        for (size_t b0 = 0; b0 < n_blocks; b0 += chunk) {
            const size_t nb = std::min(chunk, n_blocks - b0);
            if (transcipher_chunk(c, ct + b0 * 16, nb, iv, b0, rounds_done, out + b0 * 128 * BIG, reference_order, c->res)) return -1;
            if (cb) { HIP_OK(hipStreamSynchronize(c->stream)); cb(user, b0, nb); }
        }
        HIP_OK(hipStreamSynchronize(c->stream));
        return timing_collect(c);
    }
    if (pipeline_init(c)) return -1;
    if (n_blocks > chunk && !c->res2) HIP_OK(hipMalloc(&c->res2, c->ws_blocks * 128 * BIG * 8));
    uint64_t *buf[2] = { c->res, c->res2 };
    size_t prev_b0 = 0, prev_nb = 0; int prev = -1;
    int k = 0;
    // a failure half way must not leave a copy in flight into the caller's memory: both streams are drained before the error returns
    auto drained = [&](int rc) { if (rc) { const std::string keep = g_err; hipStreamSynchronize(c->copy_stream); hipStreamSynchronize(c->stream); g_err = keep; } return rc; };
#define PIPE_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return drained(fail(std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
    for (size_t b0 = 0; b0 < n_blocks; b0 += chunk, k ^= 1) {
        const size_t nb = std::min(chunk, n_blocks - b0);
        // buf[k] was the source of the copy two chunks ago: the compute stream may not write it before that copy is done
        if (b0 >= 2 * chunk) PIPE_OK(hipStreamWaitEvent(c->stream, c->ev_copied[k], 0));
        if (transcipher_chunk(c, ct + b0 * 16, nb, iv, b0, rounds_done, buf[k], reference_order, buf[k])) return drained(-1);
        PIPE_OK(hipEventRecord(c->ev_done[k], c->stream));
        PIPE_OK(hipStreamWaitEvent(c->copy_stream, c->ev_done[k], 0));
        PIPE_OK(hipMemcpyAsync(out + b0 * 128 * BIG, buf[k], nb * 128 * BIG * 8, hipMemcpyDeviceToHost, c->copy_stream));
        PIPE_OK(hipEventRecord(c->ev_copied[k], c->copy_stream));
        if (prev >= 0 && cb) { PIPE_OK(hipEventSynchronize(c->ev_copied[prev])); cb(user, prev_b0, prev_nb); }
        prev = k; prev_b0 = b0; prev_nb = nb;
    }
    PIPE_OK(hipStreamSynchronize(c->copy_stream));
    PIPE_OK(hipStreamSynchronize(c->stream));
#undef PIPE_OK
    if (cb) cb(user, prev_b0, prev_nb);
    return timing_collect(c);
}

extern "C" int tx_transcipher(tx_ctx *c, const uint8_t *ct_blocks, size_t n_blocks, const uint8_t *iv, uint64_t *out_lwe,
                              int out_is_device)
{ return transcipher_impl(c, ct_blocks, n_blocks, iv, 9, out_lwe, out_is_device, false); }

extern "C" int tx_transcipher_chunks(tx_ctx *c, const uint8_t *ct_blocks, size_t n_blocks, const uint8_t *iv, uint64_t *out_lwe,
                                     int out_is_device, tx_chunk_fn on_chunk, void *user)
{ return transcipher_impl(c, ct_blocks, n_blocks, iv, 9, out_lwe, out_is_device, false, on_chunk, user); }

extern "C" size_t tx_chunk_blocks(void) { return CHUNK_BLOCKS; }

extern "C" int tx_set_ctr_column_sharing(tx_ctx *c, int enabled)
{
    if (!c) return fail("tx_set_ctr_column_sharing: null context");
    c->ctr_share = enabled ? 1 : 0;
    return 0;
}

extern "C" int tx_set_decomposer_tie_rule(tx_ctx *c, int rule)
{
    if (!c) return fail("tx_set_decomposer_tie_rule: null context");
    if (rule != TX_TIE_BALANCED && rule != TX_TIE_TFHE_RS) return fail("tx_set_decomposer_tie_rule: rule must be TX_TIE_BALANCED (0) or TX_TIE_TFHE_RS (1)");
#ifdef TXD_EXPERIMENTS
    if (rule == TX_TIE_TFHE_RS) return fail("tx_set_decomposer_tie_rule: the experiment build's superseded kernels have the balanced rule only");
#endif
    c->tie_on = rule == TX_TIE_BALANCED ? 1 : 0;
    return 0;
}

extern "C" int tx_transcipher_partial(tx_ctx *c, const uint8_t *ct_blocks, size_t n_blocks, int rounds_done, uint64_t *out_state)
{
    if (rounds_done < 0 || rounds_done > 9) return fail("tx_transcipher_partial: rounds_done out of range");
    return transcipher_impl(c, ct_blocks, n_blocks, nullptr, rounds_done, out_state, 0, true);
}

// ---- stage entry points (host buffers) ----
#define STAGE_PROLOGUE(name)                                                    \
    if (!c || !c->have_keys) return fail(name ": context without keys");        \
    HIP_OK(hipSetDevice(c->device));                                            \
    timing_reset(c)

extern "C" int tx_stage_lwe_keyswitch(tx_ctx *c, const uint64_t *lwe_big, size_t count, uint64_t *lwe_small)
{
    STAGE_PROLOGUE("tx_stage_lwe_keyswitch");
    DevBuf in, out; const size_t sw = c->p.n + 1;
    if (in.alloc(count * BIG * 8) || out.alloc(count * sw * 8)) return fail("hipMalloc failed");
    HIP_OK(hipMemcpyAsync(in.p, lwe_big, count * BIG * 8, hipMemcpyHostToDevice, c->stream));
    run_ks(c, (uint64_t *)in.p, count, (uint64_t *)out.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(lwe_small, out.p, count * sw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}
extern "C" int tx_stage_blind_rotate(tx_ctx *c, const uint64_t *lwe_small, size_t count, uint64_t *acc)
{
    STAGE_PROLOGUE("tx_stage_blind_rotate");
    DevBuf in, out; const size_t sw = c->p.n + 1;
    if (in.alloc(count * sw * 8) || out.alloc(count * GL * 8)) return fail("hipMalloc failed");
    HIP_OK(hipMemcpyAsync(in.p, lwe_small, count * sw * 8, hipMemcpyHostToDevice, c->stream));
    run_br(c, (uint64_t *)in.p, count, (uint64_t *)out.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(acc, out.p, count * GL * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}
extern "C" int tx_stage_extract_trace(tx_ctx *c, const uint64_t *acc, size_t count, uint64_t *glev)
{
    STAGE_PROLOGUE("tx_stage_extract_trace");
    DevBuf in, out;
    if (in.alloc(count * GL * 8) || out.alloc(count * CBS_L * GL * 8)) return fail("hipMalloc failed");
    HIP_OK(hipMemcpyAsync(in.p, acc, count * GL * 8, hipMemcpyHostToDevice, c->stream));
    run_extract_trace(c, (uint64_t *)in.p, count, (uint64_t *)out.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(glev, out.p, count * CBS_L * GL * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}
extern "C" int tx_stage_scheme_switch_fourier(tx_ctx *c, const uint64_t *glev, size_t count, double *ggsw_f)
{
    STAGE_PROLOGUE("tx_stage_scheme_switch_fourier");
    DevBuf in, out; const size_t ow = count * CBS_L * KP * KP * H * sizeof(cplx);
    if (in.alloc(count * CBS_L * GL * 8) || out.alloc(ow)) return fail("hipMalloc failed");
    HIP_OK(hipMemcpyAsync(in.p, glev, count * CBS_L * GL * 8, hipMemcpyHostToDevice, c->stream));
    run_ss(c, (uint64_t *)in.p, count, (cplx *)out.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(ggsw_f, out.p, ow, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}
// a4-a9 as the pipeline runs them (run_cbs: key switch, blind rotation by launch size, then the fused persistent
// k_trace_scheme_switch4 -- or the separate kernels under TX_TRACE_VARIANT / TX_SS_VARIANT = 0), on host buffers, in chunks of
// the transcipher's own chunk size (16 384 LWEs)
extern "C" int tx_stage_circuit_bootstrap(tx_ctx *c, const uint64_t *lwe_big, size_t count, double *ggsw_f)
{
    STAGE_PROLOGUE("tx_stage_circuit_bootstrap");
    if (!lwe_big || !ggsw_f) return fail("tx_stage_circuit_bootstrap: null argument");
    if (count == 0) return timing_collect(c);
    const size_t chunk = std::min(count, CHUNK_BLOCKS * 128), gw = (size_t)CBS_L * KP * KP * H * sizeof(cplx);
    if (ensure_ws(c, (chunk + 127) / 128)) return -1;
    DevBuf in;
    if (in.alloc(chunk * BIG * 8)) return fail("hipMalloc failed");
    for (size_t i0 = 0; i0 < count; i0 += chunk) {
        const size_t m = std::min(chunk, count - i0);
        HIP_OK(hipMemcpyAsync(in.p, lwe_big + i0 * BIG, m * BIG * 8, hipMemcpyHostToDevice, c->stream));
        if (run_cbs(c, (const uint64_t *)in.p, m, c->ggsw_f)) return -1;
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync((char *)ggsw_f + i0 * gw, c->ggsw_f, m * gw, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return timing_collect(c);
}
extern "C" int tx_stage_lut_eval(tx_ctx *c, const double *ggsw_f, size_t nbytes, int ctr, int set, int nlut, uint64_t *out)
{
    STAGE_PROLOGUE("tx_stage_lut_eval");
    const uint64_t *base = ctr ? c->ctr_luts : c->luts;
    if (!base) return fail("tx_stage_lut_eval: LUTs not loaded");
    const size_t mult_sz = (size_t)16 * 2 * GL, set_sz = (ctr ? 3 : 4) * mult_sz;
    if (set < 0 || set > 9 || nlut < 1 || nlut > (ctr ? 3 : 4)) return fail("tx_stage_lut_eval: bad set/nlut");
    DevBuf in, o; const size_t iw = nbytes * 8 * CBS_L * KP * KP * H * sizeof(cplx), ow = (size_t)nlut * nbytes * 8 * BIG * 8;
    if (in.alloc(iw) || o.alloc(ow)) return fail("hipMalloc failed");
    HIP_OK(hipMemcpyAsync(in.p, ggsw_f, iw, hipMemcpyHostToDevice, c->stream));
    run_lut(c, (cplx *)in.p, nbytes, base + (size_t)set * set_sz, mult_sz, nlut, (uint64_t *)o.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(out, o.p, ow, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}

// ---- a12: max of n_values u16 (SRV_CMP:34-98, 203-351), log-depth tournament ----
// A tournament level is processed in chunks of at most MAX_CHUNK_PAIRS pairs, so the circuit-bootstrapping workspace is
// that of a 128-block transcipher chunk whatever n_values is (the Fourier GGSWs of one bit are 504 KiB: 32 768 values at
// once would need 270 GB).  Chunking only splits batches: every pair sees the same operations in the same order.
static const size_t MAX_CHUNK_PAIRS = 512;                  // 2 x 512 x 16 bits = 16 384 = CHUNK_BLOCKS x 128

// in_is_device / out_is_device: the pointer is device memory on the context's GPU (what tx_transcipher(.., out_is_device = 1)
// left there): the tournament then reads the values where they are -- the first level straight from the caller's buffer, which
// is never written -- and nothing crosses PCIe (SRV_CMP:203-263 consumes what the transciphering stage left behind).
static int max_impl(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe, int in_is_device, int out_is_device)
{
    if (!c || !lwe_bits || !out_lwe || n_values == 0) return fail("tx_max_u16: bad argument");
    if (!c->have_keys) return fail("tx_max_u16: evaluation keys not loaded");
    HIP_OK(hipSetDevice(c->device));
    timing_reset(c);
    const size_t v16 = (size_t)16 * BIG;
    size_t cnt = n_values;
    const hipMemcpyKind out_kind = out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (cnt == 1) {                                         // nothing to compare (a rank holding one value in the sharded max)
        if (!in_is_device && !out_is_device) { if (out_lwe != lwe_bits) memcpy(out_lwe, lwe_bits, v16 * 8); }
        else {
            HIP_OK(hipMemcpyAsync(out_lwe, lwe_bits, v16 * 8, in_is_device ? out_kind : (out_is_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost), c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
        }
        return timing_collect(c);
    }
    const size_t cp = std::min(cnt / 2, MAX_CHUNK_PAIRS);   // pairs per chunk
    const size_t maxu = cp * 16;                            // cmux units per chunk
    if (ensure_ws(c, (2 * cp * 16 + 127) / 128)) return -1;
    DevBuf cur, nxt, A, B, E, M0, M1, sela, selb;
    // level buffers: a host input is uploaded into `cur` and the levels alternate between `cur` and `nxt`; a device input is
    // read in place and the levels alternate between two buffers of half its size
    if (cur.alloc((in_is_device ? (cnt + 1) / 2 : cnt) * v16 * 8) || nxt.alloc(((cnt + 1) / 2) * v16 * 8)) return fail("hipMalloc failed");
    if (A.alloc(maxu * GL * 8) || B.alloc(maxu * GL * 8) || E.alloc(maxu * GL * 8) || M0.alloc(maxu * GL * 8) ||
        M1.alloc(maxu * GL * 8) || sela.alloc(maxu * 16 * sizeof(int)) || selb.alloc(maxu * 16 * sizeof(int)))
        return fail("hipMalloc failed");
    if (!in_is_device) HIP_OK(hipMemcpyAsync(cur.p, lwe_bits, cnt * v16 * 8, hipMemcpyHostToDevice, c->stream));
    const uint64_t *pc = in_is_device ? lwe_bits : (const uint64_t *)cur.p;
    uint64_t *pn = (uint64_t *)nxt.p, *spare = in_is_device ? (uint64_t *)cur.p : nullptr;
    {   // selector tables, built ONCE for a full chunk: for scan step i (bit index i, MSB-first storage) unit (q, t)
        // selects GGSW (2q) * 16 + i (operand a) or (2q + 1) * 16 + i (operand b) of the chunk.  A chunk with fewer
        // pairs uses the first `units` entries of each step's row: rows are maxu entries apart.
        std::vector<int> ha(maxu * 16), hb(maxu * 16);
        for (int i = 0; i < 16; i++)
            for (size_t q = 0; q < cp; q++)
                for (int t = 0; t < 16; t++) {
                    ha[(size_t)i * maxu + q * 16 + t] = (int)((2 * q) * 16 + i);
                    hb[(size_t)i * maxu + q * 16 + t] = (int)((2 * q + 1) * 16 + i);
                }
        HIP_OK(hipMemcpyAsync(sela.p, ha.data(), ha.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(selb.p, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));         // the host vectors go out of scope
    }
    while (cnt > 1) {
        const size_t all_pairs = cnt / 2, nx = all_pairs + (cnt & 1);
        for (size_t p0 = 0; p0 < all_pairs; p0 += cp) {
            const size_t pairs = std::min(cp, all_pairs - p0), units = pairs * 16;
            const uint64_t *src = pc + 2 * p0 * v16;        // values 2 p0 .. 2 (p0 + pairs) - 1 of this level
            if (run_cbs(c, src, 2 * pairs * 16, c->ggsw_f)) return -1;
            // operands as GLWEs: A[q][t] = value 2q bit t, B = value 2q+1 bit t: one strided launch each
            hipLaunchKernelGGL(k_lwe_to_glwe, dim3(ew_grid(units * GL)), dim3(256), 0, c->stream, src, (uint64_t *)A.p, (int)units, 2, 0);
            hipLaunchKernelGGL(k_lwe_to_glwe, dim3(ew_grid(units * GL)), dim3(256), 0, c->stream, src, (uint64_t *)B.p, (int)units, 2, 1);
            hipMemcpyAsync(E.p, A.p, units * GL * 8, hipMemcpyDeviceToDevice, c->stream);   // e = a (SURVEY 3.2 fix)
            for (int i = 15; i >= 0; i--) {                 // .rev(): least significant bit first
                const int *sa = (const int *)sela.p + (size_t)i * maxu, *sb = (const int *)selb.p + (size_t)i * maxu;
                ev_begin(c, F_LUT);                         // (no early return between ev_begin and ev_end)
                hipMemcpyAsync(M0.p, E.p, units * GL * 8, hipMemcpyDeviceToDevice, c->stream);
                hipLaunchKernelGGL(k_cmux, dim3(grid_for(units)), dim3(256), LDS_BYTES, c->stream, (uint64_t *)M0.p,
                                   (const uint64_t *)B.p, c->ggsw_f, sb, (int)units, c->tw, c->tie_on);     // b_i ? B : e
                hipMemcpyAsync(M1.p, A.p, units * GL * 8, hipMemcpyDeviceToDevice, c->stream);
                hipLaunchKernelGGL(k_cmux, dim3(grid_for(units)), dim3(256), LDS_BYTES, c->stream, (uint64_t *)M1.p,
                                   (const uint64_t *)E.p, c->ggsw_f, sb, (int)units, c->tw, c->tie_on);     // b_i ? e : A
                hipMemcpyAsync(E.p, M0.p, units * GL * 8, hipMemcpyDeviceToDevice, c->stream);
                hipLaunchKernelGGL(k_cmux, dim3(grid_for(units)), dim3(256), LDS_BYTES, c->stream, (uint64_t *)E.p,
                                   (const uint64_t *)M1.p, c->ggsw_f, sa, (int)units, c->tw, c->tie_on);    // a_i ? m1 : m0
                ev_end(c);
            }
            hipLaunchKernelGGL(k_extract0, dim3(ew_grid(units * BIG)), dim3(256), 0, c->stream, (const uint64_t *)E.p,
                               pn + p0 * v16, (int)units);
            HIP_OK(hipGetLastError());                      // sticky: also reports a failed copy or launch of the scan above
        }
        if (cnt & 1) HIP_OK(hipMemcpyAsync(pn + all_pairs * v16, pc + (cnt - 1) * v16, v16 * 8, hipMemcpyDeviceToDevice, c->stream));
        // levels are ordered by the stream: no host synchronisation.  The level just written becomes the input; the next one
        // goes to the buffer the old input occupied (host input) or to the spare half-size buffer (device input, first level)
        uint64_t *written = pn;
        pn = spare ? spare : const_cast<uint64_t *>(pc);
        spare = nullptr;
        pc = written;
        cnt = nx;
    }
    HIP_OK(hipMemcpyAsync(out_lwe, pc, v16 * 8, out_kind, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}
extern "C" int tx_max_u16(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe)
{ return max_impl(c, lwe_bits, n_values, out_lwe, 0, 0); }
extern "C" int tx_max_u16_io(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe, int in_is_device, int out_is_device)
{ return max_impl(c, lwe_bits, n_values, out_lwe, in_is_device, out_is_device); }

// ---- inner product mod 2^16 (harness mini-workload 1), bit-sliced over CMux + circuit bootstrapping ----
namespace {
struct Job { int sel; int c0, f0, c1, f1, out; };           // cmux: out = sel ? (c1 ^ f1) : (c0 ^ f0); src -1 = zero
struct Level { std::vector<std::pair<int, int>> sel; std::vector<Job> jobs; };

// Deterministic schedule shared (by construction, not by code) with the oracle: slots 0..32P-1 are the
// input bits (value v, bit b -> v*16 + 15 - b); every gate output takes the next slot number.
struct IpSchedule {
    std::vector<Level> levels;
    int result[16];
    int n_slots, zero_slot;
};
IpSchedule ip_schedule(int P)
{
    IpSchedule s;
    int next = 32 * P;
    auto in = [](int v, int b) { return v * 16 + 15 - b; };
    std::vector<std::vector<int>> cols(16);
    {   // partial products: x_b AND y_j = cmux(sel y_j; 0, x_b)
        Level L;
        for (int p = 0; p < P; p++)
            for (int j = 0; j < 16; j++) {
                const int selidx = (int)L.sel.size();
                L.sel.push_back({in(p + P, j), -1});
                for (int b = 0; b + j < 16; b++) {
                    L.jobs.push_back({selidx, -1, 0, in(p, b), 0, next});
                    cols[b + j].push_back(next++);
                }
            }
        s.levels.push_back(L);
    }
    auto full_adder = [&](Level &L, int a, int b, int c, int k, int &sum, int &carry) {
        const int selidx = (int)L.sel.size();
        L.sel.push_back({a, b});                           // t = a ^ b
        sum = next++;
        L.jobs.push_back({selidx, c, 0, c, 1, sum});       // t ? !c : c
        carry = -1;
        if (k < 15) { carry = next++; L.jobs.push_back({selidx, a, 0, c, 0, carry}); }   // t ? c : a
    };
    for (;;) {  // carry-save levels
        bool done = true;
        for (int k = 0; k < 16; k++) if (cols[k].size() > 2) done = false;
        if (done) break;
        Level L;
        std::vector<std::vector<int>> nc(16);
        for (int k = 0; k < 16; k++) {
            const std::vector<int> &q = cols[k];
            size_t i = 0;
            for (; i + 3 <= q.size(); i += 3) {
                int sum, carry;
                full_adder(L, q[i], q[i + 1], q[i + 2], k, sum, carry);
                nc[k].push_back(sum);
                if (carry >= 0) nc[k + 1].push_back(carry);
            }
            for (; i < q.size(); i++) nc[k].push_back(q[i]);
        }
        cols = nc;
        s.levels.push_back(L);
    }
    s.zero_slot = next++;
    int carry_in = -1;
    for (int k = 0; k < 16; k++) {   // ripple row
        std::vector<int> bits = cols[k];
        if (carry_in >= 0) bits.push_back(carry_in);
        carry_in = -1;
        if (bits.empty()) { s.result[k] = s.zero_slot; continue; }
        if (bits.size() == 1) { s.result[k] = bits[0]; continue; }
        Level L;
        if (bits.size() == 2) {      // half adder, selector a: sum = a ? !b : b, carry = a ? b : 0
            L.sel.push_back({bits[0], -1});
            s.result[k] = next++;
            L.jobs.push_back({0, bits[1], 0, bits[1], 1, s.result[k]});
            if (k < 15) { carry_in = next++; L.jobs.push_back({0, -1, 0, bits[1], 0, carry_in}); }
        } else {
            int sum, carry;
            full_adder(L, bits[0], bits[1], bits[2], k, sum, carry);
            s.result[k] = sum; carry_in = carry;
        }
        s.levels.push_back(L);
    }
    s.n_slots = next;
    return s;
}
}  // namespace

static int ip_impl(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe, int in_is_device, int out_is_device)
{
    if (!c || !lwe_bits || !out_lwe || n_values < 2 || (n_values & 1)) return fail("tx_inner_product_u16: need an even number of values");
    if (!c->have_keys) return fail("tx_inner_product_u16: evaluation keys not loaded");
    HIP_OK(hipSetDevice(c->device));
    timing_reset(c);
    const int P = (int)(n_values / 2);
    IpSchedule s = ip_schedule(P);
    // a level is executed in chunks of at most IP_CHUNK selectors (one batched circuit bootstrap + one
    // batched cmux launch per chunk) so that the Fourier GGSW workspace stays bounded (8 GiB)
    const size_t IP_CHUNK = 16384;
    size_t max_sel = 0, max_jobs = 0;
    for (auto &L : s.levels) {
        for (size_t s0 = 0; s0 < L.sel.size(); s0 += IP_CHUNK) {
            const size_t s1 = std::min(L.sel.size(), s0 + IP_CHUNK);
            size_t nj = 0;
            for (auto &j : L.jobs) if ((size_t)j.sel >= s0 && (size_t)j.sel < s1) nj++;
            max_sel = std::max(max_sel, s1 - s0); max_jobs = std::max(max_jobs, nj);
        }
    }
    if (ensure_ws(c, (max_sel + 127) / 128)) return -1;
    DevBuf pool, t, g0, g1, di0, di1, dsel, dc0, df0, dc1, df1, dout;
    if (pool.alloc((size_t)s.n_slots * BIG * 8) || t.alloc(max_sel * BIG * 8) || g0.alloc(max_jobs * GL * 8) || g1.alloc(max_jobs * GL * 8) ||
        di0.alloc(max_sel * 4) || di1.alloc(max_sel * 4) || dsel.alloc(max_jobs * 4) || dc0.alloc(max_jobs * 4) || df0.alloc(max_jobs * 4) ||
        dc1.alloc(max_jobs * 4) || df1.alloc(max_jobs * 4) || dout.alloc(max_jobs * 4))
        return fail("hipMalloc failed");
    HIP_OK(hipMemsetAsync(pool.p, 0, (size_t)s.n_slots * BIG * 8, c->stream));
    HIP_OK(hipMemcpyAsync(pool.p, lwe_bits, n_values * 16 * BIG * 8, in_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    std::vector<int> hi0, hi1, hs, h0, hf0, h1, hf1, ho;
    for (auto &L : s.levels) {
        size_t jpos = 0;                                   // jobs are ordered by selector index
        for (size_t s0 = 0; s0 < L.sel.size(); s0 += IP_CHUNK) {
            const size_t s1 = std::min(L.sel.size(), s0 + IP_CHUNK);
            const int ns = (int)(s1 - s0);
            size_t j1 = jpos;
            while (j1 < L.jobs.size() && (size_t)L.jobs[j1].sel < s1) j1++;
            const int nj = (int)(j1 - jpos);
            hi0.resize(ns); hi1.resize(ns);
            for (int i = 0; i < ns; i++) { hi0[i] = L.sel[s0 + i].first; hi1[i] = L.sel[s0 + i].second; }
            hs.resize(nj); h0.resize(nj); hf0.resize(nj); h1.resize(nj); hf1.resize(nj); ho.resize(nj);
            for (int i = 0; i < nj; i++) {
                const Job &jb = L.jobs[jpos + i];
                hs[i] = jb.sel - (int)s0; h0[i] = jb.c0; hf0[i] = jb.f0; h1[i] = jb.c1; hf1[i] = jb.f1; ho[i] = jb.out;
            }
            jpos = j1;
            HIP_OK(hipMemcpyAsync(di0.p, hi0.data(), ns * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(di1.p, hi1.data(), ns * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(dsel.p, hs.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(dc0.p, h0.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(df0.p, hf0.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(dc1.p, h1.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(df1.p, hf1.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(dout.p, ho.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));       // host vectors are reused by the next chunk
            hipLaunchKernelGGL(k_lwe_gather_add, dim3(ew_grid((size_t)ns * BIG)), dim3(256), 0, c->stream, (const uint64_t *)pool.p,
                               (const int *)di0.p, (const int *)di1.p, (uint64_t *)t.p, ns);
            if (run_cbs(c, (const uint64_t *)t.p, ns, c->ggsw_f)) return -1;
            hipLaunchKernelGGL(k_build_glwe, dim3(ew_grid((size_t)nj * GL)), dim3(256), 0, c->stream, (const uint64_t *)pool.p,
                               (const int *)dc0.p, (const int *)df0.p, (uint64_t *)g0.p, nj);
            hipLaunchKernelGGL(k_build_glwe, dim3(ew_grid((size_t)nj * GL)), dim3(256), 0, c->stream, (const uint64_t *)pool.p,
                               (const int *)dc1.p, (const int *)df1.p, (uint64_t *)g1.p, nj);
            ev_begin(c, F_LUT);
            hipLaunchKernelGGL(k_cmux, dim3(grid_for(nj)), dim3(256), LDS_BYTES, c->stream, (uint64_t *)g0.p, (const uint64_t *)g1.p,
                               c->ggsw_f, (const int *)dsel.p, nj, c->tw, c->tie_on);
            ev_end(c);
            hipLaunchKernelGGL(k_extract0_scatter, dim3(ew_grid((size_t)nj * BIG)), dim3(256), 0, c->stream, (const uint64_t *)g0.p,
                               (const int *)dout.p, (uint64_t *)pool.p, nj);
            HIP_OK(hipGetLastError());
        }
    }
    for (int k = 0; k < 16; k++)     // MSB first
        HIP_OK(hipMemcpyAsync(out_lwe + (size_t)(15 - k) * BIG, (uint64_t *)pool.p + (size_t)s.result[k] * BIG, BIG * 8,
                              out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return timing_collect(c);
}
extern "C" int tx_inner_product_u16(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe)
{ return ip_impl(c, lwe_bits, n_values, out_lwe, 0, 0); }
extern "C" int tx_inner_product_u16_io(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe, int in_is_device,
                                       int out_is_device)
{ return ip_impl(c, lwe_bits, n_values, out_lwe, in_is_device, out_is_device); }
