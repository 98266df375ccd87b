/*
 * tfhe_mi355x.h -- C ABI of the MI355X-native AES-under-TFHE transciphering engine
 * (libtfhe_mi355x.so, built from temp-fhe-transciphering_amd/csrc for gfx950).
 *
 * The reference (code-perspective/temp-fhe-transciphering) has no FFI: its only boundary is the
 * process-level stage contract of harness/run_submission.py:68-115.  This ABI is the in-process
 * boundary the reference's stage executables would bind if their hot path were offloaded; each
 * entry point names the reference interface it replaces.  Plain pointers and sizes only, no C++
 * or torch types; return value 0 = success, negative = error (tx_last_error() has the text);
 * no exception crosses the boundary.
 *
 * Threads: a context belongs to one GPU and calls on ONE context must be serialised by the caller
 * (the reference is single-threaded, submission/cbs_lib/Cargo.toml:31-34).  DISTINCT contexts --
 * on distinct GPUs, or several on one GPU -- may be driven concurrently from distinct threads:
 * the library keeps no mutable state outside a context except tx_last_error()'s text, which is
 * per thread (a thread reads the text of ITS last failed call), and the per-device kernel
 * attributes tx_create sets under a lock.  Every entry point selects its context's device for
 * the calling thread (hipSetDevice is per-thread state).  This is how the stage executables
 * shard an instance across the GPUs of a node: one context and one host thread per device.
 *
 * All ciphertext and key material at the boundary is the reference's standard domain:
 * unsigned 64-bit torus words (q = 2^64), layouts given below.  Fourier-domain forms are private.
 */
#ifndef TFHE_MI355X_H
#define TFHE_MI355X_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors `AesParam<u64>` (submission/cbs_lib/src/aes_params.rs:4-27); the values in use are
 * `AES_TIGHT` (submission/cbs_lib/src/aes_instances.rs:76-97).  The HIP kernels are specialised
 * for N = 1024, k = 2 and the AES_TIGHT decomposition constants; `n` (a multiple of 256, at most
 * 1024) is a run-time value.  tx_create() rejects anything else. */
typedef struct {
    uint32_t n, N, k;
    uint32_t pbs_base_log, pbs_level;
    uint32_t ks_base_log, ks_level, ks_N;
    uint32_t auto_base_log, auto_level, auto_split;
    uint32_t ss_base_log, ss_level;
    uint32_t cbs_base_log, cbs_level;
    uint32_t log_lut_count;
    double   lwe_std, glwe_std;          /* only used by the client side */
} tx_params;

typedef struct tx_ctx tx_ctx;

/* `&*AES_TIGHT` (server_encrypted_aes_decryption.rs:637) */
void tx_params_aes_tight(tx_params *p);

const char *tx_last_error(void);

/* Replaces the key-conversion prologue of `main` (server_encrypted_aes_decryption.rs:640-687):
 * creates a context on HIP device `device`. */
/* The product library reads NO environment variables.  The experiment build (libtfhe_mi355x_exp.so, -DTXD_EXPERIMENTS: the
 * superseded kernels kept for A/B tests) reads and validates TX_BR_VARIANT, TX_TRACE_VARIANT, TX_SS_VARIANT, TX_LUT_VARIANT,
 * TX_BR5_WGS, TX_TRSS_WGS_PER_CU, TX_CTR_SHARE once here; an ill-formed or out-of-range value makes its tx_create fail. */
tx_ctx *tx_create(const tx_params *p, int device);
void    tx_destroy(tx_ctx *c);

/* Standard-domain evaluation keys (host pointers), transformed once on the GPU and kept resident:
 *   bsk       LweBootstrapKeyOwned<u64>      [n][pbs_level][K][K][N]         (bsk.bin,  :630)
 *   ksk       GlweKeyswitchKeyOwned<u64>     [kN/ks_N][ks_level][n/ks_N+1][ks_N]  (ksk.bin, :631)
 *   auto_keys standard-domain GLWE key-switch keys for t = N/2^(i-1)+1, i = 1..log2 N:
 *             [log2 N][k][auto_level][K][N]  (auto_keys.bin :633-643 holds their Fourier form)
 *   ss_key    GgswCiphertextListOwned<u64>   [k][ss_level][K][K][N]          (ss_key.bin, :632)
 * levels are stored level 1 (coarsest) first, as tfhe-rs does. K = k + 1.
 * Each buffer comes with its length in 64-bit words; a length that is not exactly the one the
 * context's parameters imply (a truncated file, keys generated for another n) is rejected before
 * anything is read, as the reference's bincode deserialisation into typed containers would. */
int tx_load_eval_keys(tx_ctx *c, const uint64_t *bsk, size_t bsk_words, const uint64_t *ksk, size_t ksk_words,
                      const uint64_t *auto_keys, size_t auto_words, const uint64_t *ss_key, size_t ss_words);
/* the word counts tx_load_eval_keys expects for this context (in the order bsk, ksk, auto, ss) */
int tx_eval_key_words(const tx_ctx *c, size_t out4[4]);

/* Announces the size of the instances to come (AES blocks per call; more than tx_chunk_blocks() is the same as that many): the
 * workspace -- 0.1 GB per block, sized to the instance, not to the chunk limit -- is then allocated by tx_load_eval_keys
 * underneath the key transforms (or at once if the keys are resident) instead of inside the first tx_transcipher call.
 * Optional: every call grows the workspace on demand. */
int tx_reserve(tx_ctx *c, size_t n_blocks);

/* `AllRdKeys` (submission/src/data_struct.rs:10-26; trans_key.bin, SRV_AES :635) as one blob:
 * [set 0..8][mult x9,x11,x13,x14][byte 16][acc 2][K][N] then round 0 [byte 16][acc 2][K][N];
 * set 0 = fused rounds 10/9, sets 1..8 = rounds 8..1.  words = (9*4+1)*16*2*K*N. */
int tx_load_round_luts(tx_ctx *c, const uint64_t *luts, size_t words);

/* Forward-cipher tables for CTR mode (harness sizes >= 1, aes_keygen_and_encrypt.py:49-55; the
 * reference bins do not implement it, building blocks cbs_lib/src/aes_he.rs:686-786):
 * [set 0..9][mult x1,x2,x3][byte 16][acc 2][K][N]; words = 10*3*16*2*K*N. */
int tx_load_ctr_luts(tx_ctx *c, const uint64_t *luts, size_t words);

/* Replaces `aes_to_lwe_trasnciphering` (server_encrypted_aes_decryption.rs:28-191), batched over
 * n_blocks independent 16-byte AES blocks.  iv == NULL: the reference's ECB inverse-cipher
 * schedule; iv != NULL: CTR mode on counter blocks iv + i (the second round of a CTR batch bootstraps every DISTINCT
 * state column once: the counters of a batch agree in all but their last bytes, so most columns of that round hold the
 * same ciphertexts in every block; same results as evaluating each, tx_set_ctr_column_sharing(c, 0) does).  out_lwe: n_blocks*128 LWE
 * ciphertexts of kN+1 words, bit order MSB first inside each byte (:182-189).
 * out_is_device != 0: out_lwe is a device pointer on the context's GPU (result stays in HBM). */
int tx_transcipher(tx_ctx *c, const uint8_t *ct_blocks, size_t n_blocks, const uint8_t *iv,
                   uint64_t *out_lwe, int out_is_device);

/* The same call with progress: instances larger than tx_chunk_blocks() blocks are processed chunk by chunk, and with a host
 * `out_lwe` the LWEs of chunk k travel to the caller's memory on a copy stream while chunk k + 1 computes.  on_chunk (may be
 * NULL) is called on the calling thread once the LWEs of blocks [first_block, first_block + n_blocks) have ARRIVED in out_lwe --
 * for every chunk but the last while the GPU is already computing the next one, so that what the caller does with them (the
 * reference's stage writes them to io/<size>/ciphertext_aes_download, server_encrypted_aes_decryption.rs:699-705) overlaps the
 * computation.  The callback must not call into the same context.  Pinned memory (tx_host_alloc) makes the copies DMAs. */
typedef void (*tx_chunk_fn)(void *user, size_t first_block, size_t n_blocks);
int tx_transcipher_chunks(tx_ctx *c, const uint8_t *ct_blocks, size_t n_blocks, const uint8_t *iv,
                          uint64_t *out_lwe, int out_is_device, tx_chunk_fn on_chunk, void *user);
size_t tx_chunk_blocks(void);

/* CTR schedule: evaluate every state column of the second round (enabled = 0) instead of each distinct one once (default 1).
 * Same ciphertexts either way; the switch exists for equal-work measurements (bench.py's value_without_sharing). */
int tx_set_ctr_column_sharing(tx_ctx *c, int enabled);

/* Which way a TOP-LEVEL tie of the signed gadget decomposition goes (digit = B/2 exactly at level 1, no bits above it to decide).
 * tfhe-rs 0.5.4's SignedDecomposer -- what the reference calls at cbs_lib/src/fourier_glwe_keyswitch.rs:264,282 and
 * pbs.rs:153-159 -- always takes +B/2.  For the circuit-bootstrapping gadget (B = 4: one coefficient in four is a tie) that gives
 * the top digit a mean of +B/8, which multiplies the structured noise of the scheme-switched GGSW rows into a coherent term and
 * costs the bootstrap 3.7 sigma of margin.  TX_TIE_BALANCED (the default) sends the tie to +B/2 on even and
 * to -B/2 on odd coefficient indices: the same value mod 2^64, zero mean.  TX_TIE_TFHE_RS makes every decomposition of the path
 * (key switch, blind rotation, trace, scheme switch, LUT tree, mini-workloads) follow tfhe-rs digit for digit, so that the product
 * can be put beside a build of the reference at the ciphertext level; its outputs equal the oracle's under
 * txo_set_top_digit_rule(1) bit for bit.  Takes effect with the next call on the context. */
#define TX_TIE_BALANCED 0
#define TX_TIE_TFHE_RS  1
int tx_set_decomposer_tie_rule(tx_ctx *c, int rule);

/* Page-locked host memory, usable with every context of the process (hipHostMalloc, portable): result buffers allocated here
 * are written by DMA at link speed instead of through the runtime's staging copies.  No context needed; NULL on failure. */
void *tx_host_alloc(size_t bytes);
int   tx_host_free(void *p);
/* The same for memory the caller already has (a region of an anonymous mapping, a buffer of the host language's allocator):
 * page-locks [p, p + bytes) for every device (hipHostRegister, portable).  A large result can be pinned region by region while
 * earlier chunks compute -- a copy whose destination is not (yet) registered is still correct, only slower. */
int   tx_host_register(void *p, size_t bytes);
int   tx_host_unregister(void *p);
/* number of HIP devices visible to the process (what the stage executables shard across by default); negative on error */
int   tx_device_count(void);

/* Device memory on the context's GPU for the `*_is_device` arguments, for a host that has no HIP binding of its own (the
 * reference's Rust stages): what tx_transcipher(.., out_is_device = 1) writes and tx_max_u16_io / tx_inner_product_u16_io
 * (.., in_is_device = 1) read.  A pointer from the caller's own hipMalloc / a torch tensor on the same device works as well.
 * tx_device_alloc returns NULL on failure; the others 0 / negative. */
void *tx_device_alloc(tx_ctx *c, size_t bytes);
int   tx_device_free(tx_ctx *c, void *p);
int   tx_device_upload(tx_ctx *c, void *dst_device, const void *src_host, size_t bytes);
int   tx_device_download(tx_ctx *c, void *dst_host, const void *src_device, size_t bytes);

/* Host logic of the CTR schedule's shared second round, exposed for tests and tooling (needs no context and no GPU): for the
 * counter blocks iv + first_block + b, b < n_blocks, rep_of[b * 4 + c] is the index (into col_of) of the distinct column that
 * column c of block b equals, col_of[k] = block * 4 + column of the k-th distinct column.  Both arrays hold n_blocks * 4
 * ints.  Returns the number of distinct columns (what the second round bootstraps), negative on error. */
int tx_ctr_shared_columns(const uint8_t *iv, size_t first_block, size_t n_blocks, int *rep_of, int *col_of);

/* Same schedule stopped after `rounds_done` bootstrapped rounds (0..9), state returned in the
 * reference's internal order (LSB first); used by the parity tests only. */
int tx_transcipher_partial(tx_ctx *c, const uint8_t *ct_blocks, size_t n_blocks, int rounds_done,
                           uint64_t *out_state);

/* Replaces the body of `main` in server_encrypted_compute.rs:203-356 (max of n_values 16-bit
 * values given as n_values*16 LWE bits, MSB first); log-depth tournament.
 * out: 16 LWE ciphertexts. */
int tx_max_u16(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe);
/* The same with the location of each buffer stated (mirrors out_is_device of tx_transcipher): in_is_device != 0: lwe_bits is
 * device memory on the context's GPU, e.g. what tx_transcipher(.., out_is_device = 1) left there -- the reference's compute
 * stage consumes what the transciphering stage left behind (server_encrypted_compute.rs:203-263), here without a round trip
 * through host memory; the input is only read.  out_is_device != 0: the 16 result LWEs are written to device memory. */
int tx_max_u16_io(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe, int in_is_device, int out_is_device);

/* Mini-workload 1 of the harness (harness/cleartext_impl.py:65-70; the reference has no server
 * implementation of it): sum over i < n/2 of (v[i] * v[i + n/2] mod 2^16), mod 2^16, for n_values
 * (even) 16-bit values given as n_values*16 LWE bits, MSB first.  Bit-sliced: partial products by
 * CMux with circuit-bootstrapped selectors, carry-save reduction with full adders
 * (carry = (a^b) ? c : a, sum = (a^b) ? !c : c, one circuit bootstrap of a^b per adder), final
 * ripple-carry row.  out: 16 LWE ciphertexts, MSB first. */
int tx_inner_product_u16(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe);
/* with device-resident input and / or output, as tx_max_u16_io */
int tx_inner_product_u16_io(tx_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out_lwe, int in_is_device,
                            int out_is_device);

/* ---- stage-level entry points (one per row of SURVEY.md section 8a), host pointers ---- */
/* a4 keyswitch_lwe_ciphertext_by_glwe_keyswitch (fourier_glwe_keyswitch.rs:344-379) */
int tx_stage_lwe_keyswitch(tx_ctx *c, const uint64_t *lwe_big, size_t count, uint64_t *lwe_small);
/* a5 gen_blind_rotate_local_assign on the CBS test vector (pbs.rs:70-161, ggsw_conv.rs:250-300) */
int tx_stage_blind_rotate(tx_ctx *c, const uint64_t *lwe_small, size_t count, uint64_t *acc);
/* a6+a7 extract / preprocess / LWE->GLWE / trace_assign (ggsw_conv.rs:304-316) -> GLev */
int tx_stage_extract_trace(tx_ctx *c, const uint64_t *acc, size_t count, uint64_t *glev);
/* a8+a9 switch_scheme + convert_standard_ggsw_ciphertext_to_fourier (ggsw_conv.rs:163-193,
 * SRV_AES:430-436); output is the private Fourier layout, count*cbs_level*K*K*N doubles */
int tx_stage_scheme_switch_fourier(tx_ctx *c, const uint64_t *glev, size_t count, double *ggsw_f);
/* a4-a9 in one call, as the round pipeline runs them: circuit_bootstrap of `count` big LWEs
 * (lwe_msb_bit_to_glev_by_trace_with_preprocessing + switch_scheme + GGSW->Fourier, ggsw_conv.rs:231-318, 163-193;
 * SRV_AES:350-436): key switch, blind rotation (kernel by launch size), then extraction, trace, scheme switch and forward
 * transform in the fused persistent kernel the pipeline uses (k_trace_scheme_switch4).  Output as
 * tx_stage_scheme_switch_fourier: count*cbs_level*K*K*N doubles */
int tx_stage_circuit_bootstrap(tx_ctx *c, const uint64_t *lwe_big, size_t count, double *ggsw_f);
/* a10 evaluate_8_to_8_cipher_lut (SRV_AES:550-588) for nlut tables of loaded set `set`
 * (ctr == 0: round tables, nlut = 4 or 1 for set 9; ctr != 0: CTR tables).
 * out[(lut*nbytes + byte)*8 + bit] big LWEs */
int tx_stage_lut_eval(tx_ctx *c, const double *ggsw_f, size_t nbytes, int ctr, int set, int nlut,
                      uint64_t *out);

/* ---- measurement ---- */
/* HIP-event time (ms) spent in each kernel family during the last tx_transcipher / tx_max_u16
 * call, measured on the stream the kernels were launched on, and the number of launches.  The pipeline runs
 * extraction, trace and scheme switch as one kernel: its time is reported under ms_trace and ms_scheme_switch
 * stays 0 there (the stage entry points below run and time the two separately). */
typedef struct {
    double ms_keyswitch, ms_blind_rotate, ms_trace, ms_scheme_switch, ms_lut_eval, ms_linear, ms_total;
    uint32_t n_keyswitch, n_blind_rotate, n_trace, n_scheme_switch, n_lut_eval, n_linear;
    /* the blind-rotation kernel that processed most units of the call (tx_last_blind_rotate_kernel names it): its time,
     * launches and units -- a call may use more than one kernel (dispatch by launch size, split tails) */
    uint32_t n_br_main;
    double ms_br_main;
    uint64_t units_br_main;
    /* units (bit ciphertexts) all blind-rotation launches of the call processed together: the work actually done, e.g.
     * 8 x 8192 + 2144 for the medium instance under the CTR schedule with its shared second round (round 4) */
    uint64_t units_blind_rotate;
    /* != 0: a HIP event call failed while the call was being timed; every ms_* field is 0, the call's RESULTS are complete
     * and the call returned 0 (tx_last_error() names the event call).  The n_* / units_* fields of the blind rotation are
     * taken from the launches and stay valid. */
    uint32_t timing_failed;
    uint32_t reserved_;
} tx_timing;
/* tx_timing is APPEND-ONLY: fields are only ever added at the end (round 3: n_br_main, ms_br_main, units_br_main; round 4:
 * units_blind_rotate; round 5: timing_failed).  tx_get_timing writes tx_timing_size() bytes: a binding compiled against an older header must
 * either be rebuilt or check tx_timing_size() against its own sizeof before it calls tx_get_timing. */
size_t tx_timing_size(void);
int tx_get_timing(tx_ctx *c, tx_timing *t);
/* name of the blind-rotation kernel that processed most units of the last call ("k_blind_rotate5", "k_blind_rotate6",
 * ...; "" until a call of this context has run a blind rotation, then the one of the LAST call that ran any).  The dispatch is
 * by launch size (in the experiment build also by TX_BR_VARIANT): the measurement reports what ran, not what a copy of the rule predicts. */
const char *tx_last_blind_rotate_kernel(const tx_ctx *c);
/* Host logic of that dispatch, exposed for tests and tooling (no context, no GPU): the launches a blind rotation of `count`
 * units is split into on a chip of num_cus CUs -- kernel number (5 = k_blind_rotate5, 6 = k_blind_rotate6, 8 = k_blind_rotate8)
 * and units of each, in unit order (round 5: 8 = k_blind_rotate8 between three and four units per CU, where round 4 had k_blind_rotate7); both arrays hold two entries.  Returns the number of launches (0..2), negative on error. */
int tx_plan_blind_rotate(size_t count, int num_cus, int *variants, size_t *counts);

/* the forward-transform twiddle table [15][64] complex used by the kernels (for table pinning): eight pass-A rows
 * with the lane part of the twist folded in, seven pass-B rows */
int tx_get_twiddles(tx_ctx *c, double *out /* 15*64*2 */);

#ifdef __cplusplus
}
#endif
#endif
