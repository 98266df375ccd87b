/*
 * txclient.h -- client side of the transciphering workload (host CPU, C ABI).
 *
 * Mirrors what the reference's client stages do around the server hot path:
 *   client_key_generation.rs:10-133   -> txc_keygen
 *   client_encode_encrypt.rs:10-66 + submission/src/data_struct.rs:30-269 -> txc_make_round_luts
 *   client_decrypt_decode*.rs:18-39 + help_fun.rs:24-43 -> txc_decrypt_bits
 *   client_postprocess*.rs:18-37      -> txc_bits_to_u16
 * Randomness: ChaCha20 keystream keyed with 256 bits from the OS (getrandom(2), /dev/urandom as a
 * fallback) when seed == 0 -- the production path; the reference uses concrete-csprng's AES-CTR generator
 * seeded by the OS, client_key_generation.rs:91-96 -- or derived from the caller's 64-bit seed (tests and
 * reproducible benchmarks only: 64 bits of seed are not a key).  If the OS gives no entropy every function
 * that draws randomness returns TXC_ERR_ENTROPY; nothing is ever generated from a constant.
 * Key blobs use the layouts of include/tfhe_mi355x.h.
 *
 * Keyed-LUT confidentiality: the reference encrypts only the fused round-10/9 tables and ships the tables
 * of rounds 8..0 as trivial (mask 0, noiseless) GLWEs (data_struct.rs:146-151, :258-263), which discloses
 * those round keys -- hence the AES key -- to the server.  txc_make_round_luts and txc_make_ctr_luts
 * encrypt EVERY table; the reference's form is available only through
 * txc_make_round_luts_ex(.., TXC_LUTS_REFERENCE_TRIVIAL, ..) for format comparisons.
 */
#ifndef TXCLIENT_H
#define TXCLIENT_H
#include <stdint.h>
#include <stddef.h>
#include "../../include/tfhe_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

size_t txc_bsk_words(const tx_params *p);
size_t txc_ksk_words(const tx_params *p);
size_t txc_auto_words(const tx_params *p);
size_t txc_ss_words(const tx_params *p);
size_t txc_lut_words(const tx_params *p);
size_t txc_ctr_lut_words(const tx_params *p);

#define TXC_ERR_ENTROPY (-2)                 /* seed == 0 and the OS returned no randomness */
#define TXC_LUTS_REFERENCE_TRIVIAL 1u        /* rounds 8..0 as trivial GLWEs like the reference: DISCLOSES the AES key */

/* lwe_sk: n words (0/1); glwe_sk: k*N words (0/1), also the big LWE key */
int txc_keygen(const tx_params *p, uint64_t seed, uint64_t *lwe_sk, uint64_t *glwe_sk,
               uint64_t *bsk, uint64_t *ksk, uint64_t *auto_keys, uint64_t *ss_key);
int txc_make_round_luts(const tx_params *p, uint64_t seed, const uint8_t aes_key[16],
                        const uint64_t *glwe_sk, uint64_t *luts);
int txc_make_round_luts_ex(const tx_params *p, uint64_t seed, const uint8_t aes_key[16],
                           const uint64_t *glwe_sk, unsigned flags, uint64_t *luts);
int txc_make_ctr_luts(const tx_params *p, uint64_t seed, const uint8_t aes_key[16],
                      const uint64_t *glwe_sk, uint64_t *luts);
/* bits[i] = round(phase_i / 2^63) mod 2 for count big-LWE ciphertexts (kN+1 words each) */
int txc_decrypt_bits(const tx_params *p, const uint64_t *glwe_sk, const uint64_t *lwe, size_t count, uint8_t *bits);
/* fresh big-LWE encryptions of bits at 2^63 (used to feed tx_max_u16 directly in tests) */
int txc_encrypt_bits(const tx_params *p, uint64_t seed, const uint64_t *glwe_sk, const uint8_t *bits, size_t count, uint64_t *lwe);
/* every 16 bits -> one u16, first bit = MSB */
int txc_bits_to_u16(const uint8_t *bits, size_t nbits, uint16_t *out);

/* cleartext AES-128 (harness/aes_keygen_and_encrypt.py, cleartext_impl.py semantics) */
void txc_aes128_ecb_encrypt(const uint8_t key[16], const uint8_t *in, size_t nblocks, uint8_t *out);
void txc_aes128_ecb_decrypt(const uint8_t key[16], const uint8_t *in, size_t nblocks, uint8_t *out);
void txc_aes128_ctr_crypt(const uint8_t key[16], const uint8_t iv[16], const uint8_t *in, size_t nbytes, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
