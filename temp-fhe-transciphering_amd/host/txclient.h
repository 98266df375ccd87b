/*
 * txclient.h -- client side of the transciphering workload (host CPU, C ABI).
 *
 * Mirrors what the reference's client stages do around the server hot path:
 *   client_key_generation.rs:10-133   -> txc_keygen
 *   client_encode_encrypt.rs:10-66 + submission/src/data_struct.rs:30-269 -> txc_make_round_luts
 *   client_decrypt_decode*.rs:18-39 + help_fun.rs:24-43 -> txc_decrypt_bits
 *   client_postprocess*.rs:18-37      -> txc_bits_to_u16
 * Randomness: ChaCha20 keystream seeded from the caller's 64-bit seed, or from /dev/urandom when
 * seed == 0 (the reference uses concrete-csprng's AES-CTR generator seeded by the OS,
 * client_key_generation.rs:91-96).  Key blobs use the layouts of include/tfhe_mi355x.h.
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

/* lwe_sk: n words (0/1); glwe_sk: k*N words (0/1), also the big LWE key */
int txc_keygen(const tx_params *p, uint64_t seed, uint64_t *lwe_sk, uint64_t *glwe_sk,
               uint64_t *bsk, uint64_t *ksk, uint64_t *auto_keys, uint64_t *ss_key);
int txc_make_round_luts(const tx_params *p, uint64_t seed, const uint8_t aes_key[16],
                        const uint64_t *glwe_sk, uint64_t *luts);
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
