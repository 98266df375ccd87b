/*
 * txo_aes.c -- ORACLE (test infrastructure only; see txoracle.h): cleartext AES-128, keyed LUT
 * accumulators, the homomorphic round schedules (a1, a2, a3, a11) and the max workload (a12).
 */
#include "txoracle.h"
#include "txo_internal.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- cleartext AES-128 (submission/src/aes_manager.rs:131-433) ---------------- */
static uint8_t SBOX[256], INV_SBOX[256];
static int sbox_ready = 0;

static uint8_t gf_mul(uint8_t a, uint8_t b)
{   /* aes_manager.rs:483-499 */
    uint8_t r = 0;
    for (int i = 0; i < 8; i++) {
        if (b & 1) r ^= a;
        a = (a & 0x80) ? (uint8_t)((a << 1) ^ 0x1B) : (uint8_t)(a << 1);
        b >>= 1;
    }
    return r;
}
static void build_sbox(void)
{
    if (sbox_ready) return;
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        s ^= 0x63;
        SBOX[x] = s; INV_SBOX[s] = (uint8_t)x;
    }
    sbox_ready = 1;
}
void txo_aes128_expand(const uint8_t key[16], uint8_t rk[11][16])
{   /* aes_manager.rs:163-192; byte index = 4*col + row */
    build_sbox();
    static const uint8_t RCON[11] = { 0, 1, 2, 4, 8, 16, 32, 64, 128, 0x1B, 0x36 };
    memcpy(rk[0], key, 16);
    for (int r = 1; r <= 10; r++) {
        const uint8_t *pk = rk[r - 1];
        uint8_t *ck = rk[r];
        uint8_t t[4] = { SBOX[pk[13]], SBOX[pk[14]], SBOX[pk[15]], SBOX[pk[12]] };
        t[0] ^= RCON[r];
        for (int i = 0; i < 4; i++) ck[i] = pk[i] ^ t[i];
        for (int i = 4; i < 16; i++) ck[i] = pk[i] ^ ck[i - 4];
    }
}
static void shift_rows(uint8_t s[16])
{   /* state[col][row] = buf[(row+col)%4][row] */
    uint8_t b[16]; memcpy(b, s, 16);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) s[4 * c + r] = b[4 * ((r + c) % 4) + r];
}
static void inv_shift_rows(uint8_t s[16])
{   /* state[col][row] = buf[(4+col-row)%4][row] */
    uint8_t b[16]; memcpy(b, s, 16);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) s[4 * c + r] = b[4 * ((4 + c - r) % 4) + r];
}
static void mix_columns(uint8_t s[16])
{
    uint8_t b[16]; memcpy(b, s, 16);
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++)
        s[4 * c + r] = gf_mul(b[4 * c + r], 2) ^ gf_mul(b[4 * c + (r + 1) % 4], 3) ^ b[4 * c + (r + 2) % 4] ^ b[4 * c + (r + 3) % 4];
}
static void inv_mix_columns(uint8_t s[16])
{
    uint8_t b[16]; memcpy(b, s, 16);
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++)
        s[4 * c + r] = gf_mul(b[4 * c + r], 14) ^ gf_mul(b[4 * c + (r + 1) % 4], 11) ^
                       gf_mul(b[4 * c + (r + 2) % 4], 13) ^ gf_mul(b[4 * c + (r + 3) % 4], 9);
}
void txo_aes128_encrypt_block(const uint8_t rk[11][16], const uint8_t in[16], uint8_t out[16])
{
    build_sbox();
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[0][i];
    for (int r = 1; r < 10; r++) {
        for (int i = 0; i < 16; i++) s[i] = SBOX[s[i]];
        shift_rows(s); mix_columns(s);
        for (int i = 0; i < 16; i++) s[i] ^= rk[r][i];
    }
    for (int i = 0; i < 16; i++) s[i] = SBOX[s[i]];
    shift_rows(s);
    for (int i = 0; i < 16; i++) out[i] = s[i] ^ rk[10][i];
}
void txo_aes128_decrypt_block(const uint8_t rk[11][16], const uint8_t in[16], uint8_t out[16])
{   /* aes_manager.rs:225-238 */
    build_sbox();
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[10][i];
    for (int r = 9; r >= 1; r--) {
        inv_shift_rows(s);
        for (int i = 0; i < 16; i++) s[i] = INV_SBOX[s[i]] ^ rk[r][i];
        inv_mix_columns(s);
    }
    inv_shift_rows(s);
    for (int i = 0; i < 16; i++) out[i] = INV_SBOX[s[i]] ^ rk[0][i];
}

/* ---------------- keyed LUT accumulators (data_struct.rs:30-269, aes_he.rs:835-877) ---------------- */
/* coefficient i of accumulator acc carries bit (4*acc + i/256) of table[i%256] at 2^63 */
static void pack_lut(const txo_params *p, const uint8_t table[256], uint32_t acc, uint64_t *pt)
{
    for (uint32_t i = 0; i < p->N; i++) {
        uint32_t lut_idx = acc * (p->N / 256) + i / 256;
        pt[i] = (uint64_t)((table[i % 256] >> lut_idx) & 1) << 63;
    }
}
static void emit_lut(const txo_params *p, txo_rng *rng, const uint64_t *sk, int encrypted,
                     const uint8_t table[256], uint64_t *dst /* [2][K][N] */)
{
    uint32_t N = p->N, k = p->k, K = k + 1;
    uint64_t *pt = (uint64_t *)malloc(sizeof(uint64_t) * N);
    for (uint32_t a = 0; a < 2; a++) {
        uint64_t *g = dst + (size_t)a * K * N;
        pack_lut(p, table, a, pt);
        if (encrypted) txo_glwe_encrypt(p, rng, sk, pt, p->glwe_std, g);
        else { memset(g, 0, sizeof(uint64_t) * (size_t)k * N); memcpy(g + (size_t)k * N, pt, sizeof(uint64_t) * N); }
    }
    free(pt);
}

static const uint8_t INV_MULTS[4] = { 9, 11, 13, 14 };

void txo_make_round_luts(const txo_params *p, uint64_t seed, const uint8_t aes_key[16],
                         const uint64_t *glwe_sk, uint64_t *luts)
{
    build_sbox();
    uint8_t rk[11][16]; txo_aes128_expand(aes_key, rk);
    txo_rng rng; txo_rng_seed(&rng, seed);
    size_t gl = (size_t)(p->k + 1) * p->N;
    uint8_t base[256], tab[256];
    for (int set = 0; set < 9; set++)
        for (int m = 0; m < 4; m++)
            for (int j = 0; j < 16; j++) {
                if (set == 0) {
                    /* aes_manager.rs:338-396: AddRK10, InvShiftRows, InvSubBytes, AddRK9 */
                    int c = j / 4, r = j % 4, pre = 4 * ((4 + c - r) % 4) + r;
                    for (int x = 0; x < 256; x++) base[x] = INV_SBOX[x ^ rk[10][pre]] ^ rk[9][j];
                } else {
                    int round = 9 - set;                      /* aes_manager.rs:398-420 */
                    for (int x = 0; x < 256; x++) base[x] = INV_SBOX[x] ^ rk[round][j];
                }
                for (int x = 0; x < 256; x++) tab[x] = gf_mul(base[x], INV_MULTS[m]);
                emit_lut(p, &rng, glwe_sk, set == 0, tab, luts + ((((size_t)set * 4 + m) * 16 + j) * 2) * gl);
            }
    for (int j = 0; j < 16; j++) {                            /* aes_manager.rs:422-432 */
        for (int x = 0; x < 256; x++) tab[x] = INV_SBOX[x] ^ rk[0][j];
        emit_lut(p, &rng, glwe_sk, 0, tab, luts + (((size_t)9 * 4 * 16 + j) * 2) * gl);
    }
}

static int sr_dst(int j) { int c = j / 4, r = j % 4; return 4 * ((4 + c - r) % 4) + r; } /* where byte j lands after ShiftRows */

void txo_make_ctr_luts(const txo_params *p, uint64_t seed, const uint8_t aes_key[16],
                       const uint64_t *glwe_sk, uint64_t *luts)
{   /* forward cipher, cbs_lib/src/aes_he.rs:686-786 (S, 2S, 3S tables), keyed like data_struct.rs */
    build_sbox();
    uint8_t rk[11][16]; txo_aes128_expand(aes_key, rk);
    txo_rng rng; txo_rng_seed(&rng, seed);
    size_t gl = (size_t)(p->k + 1) * p->N;
    uint8_t tab[256];
    memset(luts, 0, sizeof(uint64_t) * txo_ctr_lut_words(p));
    for (int set = 0; set < 10; set++)
        for (int m = 0; m < 3; m++)
            for (int j = 0; j < 16; j++) {
                if (set == 9 && m > 0) continue;
                for (int x = 0; x < 256; x++) {
                    uint8_t s = SBOX[x ^ rk[set][j]];
                    tab[x] = (set == 9) ? (uint8_t)(s ^ rk[10][sr_dst(j)]) : gf_mul(s, (uint8_t)(m + 1));
                }
                emit_lut(p, &rng, glwe_sk, set == 0, tab, luts + ((((size_t)set * 3 + m) * 16 + j) * 2) * gl);
            }
}

/* ---------------- homomorphic linear layers (SRV_AES:195-265) ---------------- */
static void he_inv_mix_columns(size_t big, uint64_t *st, const uint64_t *s9, const uint64_t *s11,
                               const uint64_t *s13, const uint64_t *s14)
{
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
        size_t o = (size_t)(4 * c + r) * 8 * big;
        size_t o11 = (size_t)(4 * c + (r + 1) % 4) * 8 * big, o13 = (size_t)(4 * c + (r + 2) % 4) * 8 * big,
               o9 = (size_t)(4 * c + (r + 3) % 4) * 8 * big;
        for (size_t j = 0; j < 8 * big; j++) st[o + j] = s14[o + j] + s11[o11 + j] + s13[o13 + j] + s9[o9 + j];
    }
}
static void he_mix_columns(size_t big, uint64_t *st, const uint64_t *s1, const uint64_t *s2, const uint64_t *s3)
{   /* aes_he.rs:441-474: out[r] = 2 s[r] + 3 s[r+1] + s[r+2] + s[r+3] */
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
        size_t o = (size_t)(4 * c + r) * 8 * big;
        size_t o1 = (size_t)(4 * c + (r + 1) % 4) * 8 * big, o2 = (size_t)(4 * c + (r + 2) % 4) * 8 * big,
               o3 = (size_t)(4 * c + (r + 3) % 4) * 8 * big;
        for (size_t j = 0; j < 8 * big; j++) st[o + j] = s2[o + j] + s3[o1 + j] + s1[o2 + j] + s1[o3 + j];
    }
}
static void he_inv_shift_rows(size_t big, uint64_t *st)
{   /* dst(r,c) <- src(r,(4-r+c)%4) */
    uint64_t *b = (uint64_t *)malloc(sizeof(uint64_t) * 128 * big);
    memcpy(b, st, sizeof(uint64_t) * 128 * big);
    for (int r = 1; r < 4; r++) for (int c = 0; c < 4; c++)
        memcpy(st + (size_t)(4 * c + r) * 8 * big, b + (size_t)(4 * ((4 - r + c) % 4) + r) * 8 * big, sizeof(uint64_t) * 8 * big);
    free(b);
}
static void he_shift_rows(size_t big, uint64_t *st)
{   /* aes_he.rs:348-368: dst(r,c) <- src(r,(r+c)%4) */
    uint64_t *b = (uint64_t *)malloc(sizeof(uint64_t) * 128 * big);
    memcpy(b, st, sizeof(uint64_t) * 128 * big);
    for (int r = 1; r < 4; r++) for (int c = 0; c < 4; c++)
        memcpy(st + (size_t)(4 * c + r) * 8 * big, b + (size_t)(4 * ((r + c) % 4) + r) * 8 * big, sizeof(uint64_t) * 8 * big);
    free(b);
}
/* a2: aes_he.rs:64-93 */
static void known_rotate(const txo_params *p, const uint8_t bytes[16], const uint64_t *lut_set /* [16][2][K][N] */,
                         uint64_t *out /* 128 x big */)
{
    size_t gl = (size_t)(p->k + 1) * p->N, big = (size_t)p->k * p->N + 1;
    uint32_t num_par = p->N / 256;
    for (int i = 0; i < 128; i++) {
        int byte = i / 8, bit = i % 8;
        uint32_t acc = bit / num_par, lut = bit % num_par;
        txo_sample_extract(p, lut_set + ((size_t)byte * 2 + acc) * gl, lut * 256 + bytes[byte], out + (size_t)i * big);
    }
}

/* one bootstrapped S-box layer over all blocks: KS + CBS + LUT evaluation with nlut tables */
static void sbox_layer(const txo_ctx *c, const uint64_t *state, size_t n_blocks, const uint64_t *lut_set,
                       size_t nlut, size_t set_stride, uint64_t *outs /* nlut x n_blocks*128 x big */)
{
    const txo_params *p = &c->p;
    size_t K = p->k + 1, gl = K * p->N, L = p->cbs_level, nbits = n_blocks * 128;
    double *gf = (double *)malloc(sizeof(double) * nbits * L * K * gl);
    txo_circuit_bootstrap(c, state, nbits, gf);
    txo_lut_eval(c, gf, n_blocks * 16, lut_set, nlut, 2 * gl, set_stride, outs);
    free(gf);
}

void txo_transcipher_partial(const txo_ctx *c, const uint64_t *luts, const uint8_t *ct,
                             size_t n_blocks, int rounds_done, uint64_t *out_state)
{
    const txo_params *p = &c->p;
    size_t K = p->k + 1, gl = K * p->N, big = (size_t)p->k * p->N + 1, blk = 128 * big;
    size_t set_sz = (size_t)4 * 16 * 2 * gl, mult_sz = (size_t)16 * 2 * gl;
    uint64_t *sm = (uint64_t *)malloc(sizeof(uint64_t) * 4 * n_blocks * blk);
    uint64_t *st = out_state;
    /* rounds 10/9: SRV_AES:89-128 */
    for (size_t b = 0; b < n_blocks; b++) {
        uint8_t t[16]; memcpy(t, ct + 16 * b, 16); inv_shift_rows(t);
        for (int m = 0; m < 4; m++) known_rotate(p, t, luts + (size_t)m * mult_sz, sm + ((size_t)m * n_blocks + b) * blk);
        he_inv_mix_columns(big, st + b * blk, sm + (0 * n_blocks + b) * blk, sm + (1 * n_blocks + b) * blk,
                           sm + (2 * n_blocks + b) * blk, sm + (3 * n_blocks + b) * blk);
        he_inv_shift_rows(big, st + b * blk);
    }
    /* rounds 8..1: SRV_AES:130-163 */
    for (int s = 1; s <= 8 && s <= rounds_done; s++) {
        sbox_layer(c, st, n_blocks, luts + (size_t)s * set_sz, 4, mult_sz, sm);
        for (size_t b = 0; b < n_blocks; b++) {
            he_inv_mix_columns(big, st + b * blk, sm + (0 * n_blocks + b) * blk, sm + (1 * n_blocks + b) * blk,
                               sm + (2 * n_blocks + b) * blk, sm + (3 * n_blocks + b) * blk);
            he_inv_shift_rows(big, st + b * blk);
        }
    }
    /* last round: SRV_AES:166-180 */
    if (rounds_done >= 9) {
        sbox_layer(c, st, n_blocks, luts + (size_t)9 * set_sz, 1, 0, sm);
        memcpy(st, sm, sizeof(uint64_t) * n_blocks * blk);
    }
    free(sm);
}

static void msb_first(size_t big, uint64_t *st, size_t n_bytes)
{   /* SRV_AES:182-189 */
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * 8 * big);
    for (size_t b = 0; b < n_bytes; b++) {
        memcpy(t, st + b * 8 * big, sizeof(uint64_t) * 8 * big);
        for (int i = 0; i < 8; i++) memcpy(st + (b * 8 + i) * big, t + (size_t)(7 - i) * big, sizeof(uint64_t) * big);
    }
    free(t);
}

void txo_transcipher(const txo_ctx *c, const uint64_t *luts, const uint8_t *ct, size_t n_blocks, uint64_t *out)
{
    txo_transcipher_partial(c, luts, ct, n_blocks, 9, out);
    msb_first((size_t)c->p.k * c->p.N + 1, out, n_blocks * 16);
}

void txo_transcipher_ctr(const txo_ctx *c, const uint64_t *ctr_luts, const uint8_t iv[16],
                         const uint8_t *ct, size_t n_blocks, uint64_t *out)
{
    const txo_params *p = &c->p;
    size_t K = p->k + 1, gl = K * p->N, big = (size_t)p->k * p->N + 1, blk = 128 * big;
    size_t set_sz = (size_t)3 * 16 * 2 * gl, mult_sz = (size_t)16 * 2 * gl;
    uint64_t *sm = (uint64_t *)malloc(sizeof(uint64_t) * 3 * n_blocks * blk);
    uint64_t *st = out;
    /* round 1 on the public counter blocks iv + i (128-bit big-endian, aes_keygen_and_encrypt.py:52-54) */
    for (size_t b = 0; b < n_blocks; b++) {
        uint8_t ctr[16]; memcpy(ctr, iv, 16);
        uint64_t add = b;
        for (int i = 15; i >= 0 && add; i--) { uint64_t v = (uint64_t)ctr[i] + (add & 0xFF); ctr[i] = (uint8_t)v; add = (add >> 8) + (v >> 8); }
        for (int m = 0; m < 3; m++) known_rotate(p, ctr, ctr_luts + (size_t)m * mult_sz, sm + ((size_t)m * n_blocks + b) * blk);
    }
    for (int s = 1; s <= 9; s++) {
        /* finish round s: ShiftRows, MixColumns (AddRoundKey s is folded into the next table) */
        for (size_t b = 0; b < n_blocks; b++) {
            for (int m = 0; m < 3; m++) he_shift_rows(big, sm + ((size_t)m * n_blocks + b) * blk);
            he_mix_columns(big, st + b * blk, sm + (0 * n_blocks + b) * blk, sm + (1 * n_blocks + b) * blk,
                           sm + (2 * n_blocks + b) * blk);
        }
        /* S-box of round s+1 */
        sbox_layer(c, st, n_blocks, ctr_luts + (size_t)s * set_sz, s == 9 ? 1 : 3, mult_sz, sm);
    }
    for (size_t b = 0; b < n_blocks; b++) {
        memcpy(st + b * blk, sm + b * blk, sizeof(uint64_t) * blk);
        he_shift_rows(big, st + b * blk);
        for (int i = 0; i < 128; i++)
            if ((ct[16 * b + i / 8] >> (i % 8)) & 1) st[b * blk + (size_t)i * big + big - 1] += (uint64_t)1 << 63;
    }
    free(sm);
    msb_first(big, out, n_blocks * 16);
}

/* ---------------- a12: max (SRV_CMP:34-98, 203-351) ---------------- */
/* cmux_assign(ct0, ct1, g): ct1 -= ct0; ct0 += g (x) ct1 */
static void cmux(const txo_ctx *c, uint64_t *ct0, uint64_t *ct1, const double *g)
{
    size_t gl = (size_t)(c->p.k + 1) * c->p.N;
    for (size_t j = 0; j < gl; j++) ct1[j] -= ct0[j];
    txo_extprod_add(&c->p, ct0, g, ct1, c->p.cbs_base_log, c->p.cbs_level);
}
static void lwe_to_glwe_const(const txo_params *p, const uint64_t *lwe, uint64_t *glwe)
{
    uint32_t N = p->N, k = p->k;
    memset(glwe, 0, sizeof(uint64_t) * (size_t)(k + 1) * N);
    glwe[(size_t)k * N] = lwe[(size_t)k * N];
    for (uint32_t q = 0; q < k; q++) {
        glwe[(size_t)q * N] = lwe[(size_t)q * N];
        for (uint32_t j = 1; j < N; j++) glwe[(size_t)q * N + j] = (uint64_t)0 - lwe[(size_t)q * N + N - j];
    }
}
/* out = max(a, b); a_g, b_g: 16 Fourier GGSWs each (MSB first); a_l, b_l: 16 big LWEs */
static void max_of_two(const txo_ctx *c, const double *a_g, const double *b_g, const uint64_t *a_l,
                       const uint64_t *b_l, uint64_t *out)
{
    const txo_params *p = &c->p;
    size_t K = p->k + 1, gl = K * p->N, big = (size_t)p->k * p->N + 1, gsz = (size_t)p->cbs_level * K * gl;
    #pragma omp parallel for num_threads(txo_get_threads()) schedule(dynamic)
    for (int t = 0; t < 16; t++) {
        uint64_t *A = (uint64_t *)malloc(sizeof(uint64_t) * gl * 5);
        uint64_t *B = A + gl, *e = A + 2 * gl, *m0 = A + 3 * gl, *m1 = A + 4 * gl;
        uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * gl);
        lwe_to_glwe_const(p, a_l + (size_t)t * big, A);
        lwe_to_glwe_const(p, b_l + (size_t)t * big, B);
        memcpy(e, A, sizeof(uint64_t) * gl);      /* reference leaves e stale (SRV_CMP:54-55); SURVEY 3.2 fix */
        for (int i = 15; i >= 0; i--) {           /* .rev(): LSB first */
            memcpy(m0, e, sizeof(uint64_t) * gl); memcpy(tmp, B, sizeof(uint64_t) * gl);
            cmux(c, m0, tmp, b_g + (size_t)i * gsz);               /* b_i ? B : e */
            memcpy(m1, A, sizeof(uint64_t) * gl); memcpy(tmp, e, sizeof(uint64_t) * gl);
            cmux(c, m1, tmp, b_g + (size_t)i * gsz);               /* b_i ? e : A */
            memcpy(e, m0, sizeof(uint64_t) * gl);
            cmux(c, e, m1, a_g + (size_t)i * gsz);                 /* a_i ? m1 : m0 */
        }
        txo_sample_extract(p, e, 0, out + (size_t)t * big);
        free(A); free(tmp);
    }
}
void txo_max_u16(const txo_ctx *c, const uint64_t *lwe_bits, size_t n_values, int tournament, uint64_t *out16)
{
    const txo_params *p = &c->p;
    size_t K = p->k + 1, gl = K * p->N, big = (size_t)p->k * p->N + 1, gsz = (size_t)p->cbs_level * K * gl;
    size_t v16 = 16 * big;
    if (n_values == 1) { memcpy(out16, lwe_bits, sizeof(uint64_t) * v16); return; }
    if (!tournament) {
        double *g_in = (double *)malloc(sizeof(double) * n_values * 16 * gsz);
        double *g_mid = (double *)malloc(sizeof(double) * 16 * gsz);
        uint64_t *mid = (uint64_t *)malloc(sizeof(uint64_t) * v16);
        txo_circuit_bootstrap(c, lwe_bits, n_values * 16, g_in);
        max_of_two(c, g_in, g_in + 16 * gsz, lwe_bits, lwe_bits + v16, out16);
        for (size_t i = 2; i < n_values; i++) {
            memcpy(mid, out16, sizeof(uint64_t) * v16);
            txo_circuit_bootstrap(c, mid, 16, g_mid);
            max_of_two(c, g_mid, g_in + i * 16 * gsz, mid, lwe_bits + i * v16, out16);
        }
        free(g_in); free(g_mid); free(mid);
        return;
    }
    /* log-depth tournament: survivors halve each level */
    size_t cnt = n_values;
    uint64_t *cur = (uint64_t *)malloc(sizeof(uint64_t) * cnt * v16);
    memcpy(cur, lwe_bits, sizeof(uint64_t) * cnt * v16);
    while (cnt > 1) {
        size_t pairs = cnt / 2, nxt = pairs + (cnt & 1);
        double *g = (double *)malloc(sizeof(double) * 2 * pairs * 16 * gsz);
        uint64_t *o = (uint64_t *)malloc(sizeof(uint64_t) * nxt * v16);
        txo_circuit_bootstrap(c, cur, 2 * pairs * 16, g);
        for (size_t q = 0; q < pairs; q++)
            max_of_two(c, g + (2 * q) * 16 * gsz, g + (2 * q + 1) * 16 * gsz, cur + (2 * q) * v16, cur + (2 * q + 1) * v16, o + q * v16);
        if (cnt & 1) memcpy(o + pairs * v16, cur + (cnt - 1) * v16, sizeof(uint64_t) * v16);
        free(g); free(cur); cur = o; cnt = nxt;
    }
    memcpy(out16, cur, sizeof(uint64_t) * v16);
    free(cur);
}

/* ---------------- inner product mod 2^16, bit-sliced ---------------- */
typedef struct { int sel, c0, f0, c1, f1, out; } ip_job;
typedef struct { int *sel0, *sel1; int nsel; ip_job *jobs; int njobs; } ip_level;
typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *q, int x) { if (q->n == q->cap) { q->cap = q->cap ? 2 * q->cap : 16; q->v = (int *)realloc(q->v, sizeof(int) * q->cap); } q->v[q->n++] = x; }

typedef struct { ip_level *lv; int nlv, cap; } ip_sched;
static ip_level *new_level(ip_sched *s, int maxsel, int maxjobs)
{
    if (s->nlv == s->cap) { s->cap = s->cap ? 2 * s->cap : 32; s->lv = (ip_level *)realloc(s->lv, sizeof(ip_level) * s->cap); }
    ip_level *L = &s->lv[s->nlv++];
    L->sel0 = (int *)malloc(sizeof(int) * (maxsel + 1)); L->sel1 = (int *)malloc(sizeof(int) * (maxsel + 1));
    L->jobs = (ip_job *)malloc(sizeof(ip_job) * (maxjobs + 2)); L->nsel = 0; L->njobs = 0;
    return L;
}
static void full_adder(ip_level *L, int a, int b, int cc, int k, int *next, int *sum, int *carry)
{
    int si = L->nsel; L->sel0[si] = a; L->sel1[si] = b; L->nsel++;
    *sum = (*next)++;
    ip_job js = { si, cc, 0, cc, 1, *sum }; L->jobs[L->njobs++] = js;           /* t ? !c : c */
    *carry = -1;
    if (k < 15) { *carry = (*next)++; ip_job jc = { si, a, 0, cc, 0, *carry }; L->jobs[L->njobs++] = jc; }   /* t ? c : a */
}

void txo_inner_product_u16(const txo_ctx *c, const uint64_t *lwe_bits, size_t n_values, uint64_t *out16)
{
    const txo_params *p = &c->p;
    size_t K = p->k + 1, gl = K * p->N, big = (size_t)p->k * p->N + 1, gsz = (size_t)p->cbs_level * K * gl;
    int P = (int)(n_values / 2), next = 32 * P;
    ip_sched S = { 0, 0, 0 };
    ivec cols[16]; memset(cols, 0, sizeof cols);
    {   /* partial products */
        ip_level *L = new_level(&S, 16 * P, 136 * P);
        for (int pr = 0; pr < P; pr++)
            for (int j = 0; j < 16; j++) {
                int si = L->nsel; L->sel0[si] = (pr + P) * 16 + 15 - j; L->sel1[si] = -1; L->nsel++;
                for (int b = 0; b + j < 16; b++) {
                    ip_job jb = { si, -1, 0, pr * 16 + 15 - b, 0, next }; L->jobs[L->njobs++] = jb;
                    iv_push(&cols[b + j], next++);
                }
            }
    }
    for (;;) {
        int done = 1, tot = 0;
        for (int k = 0; k < 16; k++) { if (cols[k].n > 2) done = 0; tot += cols[k].n; }
        if (done) break;
        ip_level *L = new_level(&S, tot / 3 + 16, 2 * (tot / 3 + 16));
        ivec nc[16]; memset(nc, 0, sizeof nc);
        for (int k = 0; k < 16; k++) {
            int i = 0;
            for (; i + 3 <= cols[k].n; i += 3) {
                int sum, carry;
                full_adder(L, cols[k].v[i], cols[k].v[i + 1], cols[k].v[i + 2], k, &next, &sum, &carry);
                iv_push(&nc[k], sum);
                if (carry >= 0) iv_push(&nc[k + 1], carry);
            }
            for (; i < cols[k].n; i++) iv_push(&nc[k], cols[k].v[i]);
        }
        for (int k = 0; k < 16; k++) { free(cols[k].v); cols[k] = nc[k]; }
    }
    int zero_slot = next++, result[16], carry_in = -1;
    for (int k = 0; k < 16; k++) {
        int bits[3], nb = 0;
        for (int i = 0; i < cols[k].n; i++) bits[nb++] = cols[k].v[i];
        if (carry_in >= 0) bits[nb++] = carry_in;
        carry_in = -1;
        if (nb == 0) { result[k] = zero_slot; continue; }
        if (nb == 1) { result[k] = bits[0]; continue; }
        ip_level *L = new_level(&S, 1, 2);
        if (nb == 2) {
            L->sel0[0] = bits[0]; L->sel1[0] = -1; L->nsel = 1;
            result[k] = next++;
            ip_job js = { 0, bits[1], 0, bits[1], 1, result[k] }; L->jobs[L->njobs++] = js;
            if (k < 15) { carry_in = next++; ip_job jc = { 0, -1, 0, bits[1], 0, carry_in }; L->jobs[L->njobs++] = jc; }
        } else {
            int sum, carry;
            full_adder(L, bits[0], bits[1], bits[2], k, &next, &sum, &carry);
            result[k] = sum; carry_in = carry;
        }
    }
    /* evaluate */
    uint64_t *pool = (uint64_t *)calloc((size_t)next * big, sizeof(uint64_t));
    memcpy(pool, lwe_bits, sizeof(uint64_t) * n_values * 16 * big);
    for (int li = 0; li < S.nlv; li++) {
        ip_level *L = &S.lv[li];
        uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)L->nsel * big);
        for (int i = 0; i < L->nsel; i++)
            for (size_t w = 0; w < big; w++)
                t[(size_t)i * big + w] = pool[(size_t)L->sel0[i] * big + w] + (L->sel1[i] >= 0 ? pool[(size_t)L->sel1[i] * big + w] : 0);
        double *gf = (double *)malloc(sizeof(double) * (size_t)L->nsel * gsz);
        txo_circuit_bootstrap(c, t, (size_t)L->nsel, gf);
        #pragma omp parallel for num_threads(txo_get_threads()) schedule(dynamic)
        for (int i = 0; i < L->njobs; i++) {
            ip_job jb = L->jobs[i];
            uint64_t *g0 = (uint64_t *)calloc(gl, sizeof(uint64_t)), *g1 = (uint64_t *)calloc(gl, sizeof(uint64_t));
            if (jb.c0 >= 0) lwe_to_glwe_const(p, pool + (size_t)jb.c0 * big, g0);
            if (jb.c1 >= 0) lwe_to_glwe_const(p, pool + (size_t)jb.c1 * big, g1);
            if (jb.f0) g0[(size_t)p->k * p->N] += (uint64_t)1 << 63;
            if (jb.f1) g1[(size_t)p->k * p->N] += (uint64_t)1 << 63;
            cmux(c, g0, g1, gf + (size_t)jb.sel * gsz);
            txo_sample_extract(p, g0, 0, pool + (size_t)jb.out * big);
            free(g0); free(g1);
        }
        free(t); free(gf);
    }
    for (int k = 0; k < 16; k++) memcpy(out16 + (size_t)(15 - k) * big, pool + (size_t)result[k] * big, sizeof(uint64_t) * big);
    free(pool);
    for (int k = 0; k < 16; k++) free(cols[k].v);
    for (int li = 0; li < S.nlv; li++) { free(S.lv[li].sel0); free(S.lv[li].sel1); free(S.lv[li].jobs); }
    free(S.lv);
}
