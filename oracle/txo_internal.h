/* txo_internal.h -- ORACLE internals (test infrastructure only; see txoracle.h). */
#ifndef TXO_INTERNAL_H
#define TXO_INTERNAL_H
#include "txoracle.h"

typedef struct { uint64_t s[4]; int have_spare; double spare; } txo_rng;
void     txo_rng_seed(txo_rng *r, uint64_t seed);
uint64_t txo_rng_u64(txo_rng *r);
double   txo_rng_gauss(txo_rng *r);
uint64_t txo_rng_noise(txo_rng *r, double std);

struct txo_ctx {
    txo_params p;
    double *bsk_f;    /* [n][l][K][K] polys of N doubles (N/2 complex, F-layout), scaled by 2/N */
    double *ksk_f;    /* [kin][l][Kout] polys of ks_N doubles */
    double *auto_f;   /* [2 splits][log2 N][k][l][K] polys */
    double *ss_f;     /* [k][l][K][K] polys */
};

/* per-thread scratch that is never freed: malloc/free inside OpenMP regions serialise on glibc arenas */
void *txo_tls(int slot, size_t bytes);
int  txo_get_threads(void);
void txo_poly_rotate(uint32_t N, const uint64_t *in, int64_t s, uint64_t *out);
void txo_poly_auto(uint32_t N, const uint64_t *in, uint32_t t, uint64_t *out);
void txo_sample_extract(const txo_params *p, const uint64_t *glwe, uint32_t nth, uint64_t *lwe);
void txo_glwe_encrypt(const txo_params *p, txo_rng *rng, const uint64_t *sk, const uint64_t *pt,
                      double std, uint64_t *out);
void txo_extprod_add(const txo_params *p, uint64_t *out, const double *gF, const uint64_t *in,
                     uint32_t B, uint32_t l);
#endif
