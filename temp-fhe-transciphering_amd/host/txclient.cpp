// txclient.cpp -- client-side key generation, keyed-LUT encryption, decryption (see txclient.h).
#include "txclient.h"
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/random.h>

namespace {

// ---------------------------------------------------------------- ChaCha20 keystream generator
struct ChaCha {
    uint32_t st[16];
    uint8_t buf[64];
    int pos = 64;
    bool have_spare = false;
    double spare = 0;
    bool ok = true;             // false: no entropy for an unseeded generator; every entry point then fails
    static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    explicit ChaCha(uint64_t seed, uint64_t stream)
    {
        static const uint32_t sigma[4] = { 0x61707865, 0x3320646e, 0x79622d32, 0x6b206574 };
        uint8_t key[32] = { 0 };
        if (seed == 0) {
            // production path: 256 bits from the OS; no entropy is an ERROR (never a constant key:
            // secret keys, masks and noise would all be predictable)
            ok = os_entropy(key, sizeof key);
        } else {
            for (int i = 0; i < 8; i++) key[i] = (uint8_t)(seed >> (8 * i));
            for (int i = 8; i < 32; i++) key[i] = (uint8_t)(0xA5 ^ (i * 29));
        }
        for (int i = 0; i < 4; i++) st[i] = sigma[i];
        for (int i = 0; i < 8; i++) memcpy(&st[4 + i], key + 4 * i, 4);
        st[12] = 0; st[13] = 0;
        st[14] = (uint32_t)stream; st[15] = (uint32_t)(stream >> 32);
    }
    static bool os_entropy(uint8_t *dst, size_t len)
    {
        size_t got = 0;
        while (got < len) {
            ssize_t r = getrandom(dst + got, len - got, 0);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got == len) return true;
        FILE *f = fopen("/dev/urandom", "rb");                     // kernels without getrandom(2)
        const bool rd = f && fread(dst, 1, len, f) == len;
        if (f) fclose(f);
        return rd;
    }
    void block()
    {
        uint32_t x[16];
        memcpy(x, st, sizeof x);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) { uint32_t v = x[i] + st[i]; memcpy(buf + 4 * i, &v, 4); }
        if (++st[12] == 0) ++st[13];
        pos = 0;
    }
    uint64_t u64()
    {
        if (pos > 56) block();
        uint64_t v; memcpy(&v, buf + pos, 8); pos += 8;
        return v;
    }
    double unit() { return ((double)(u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    double gauss()
    {
        if (have_spare) { have_spare = false; return spare; }
        double u = unit(), v = unit(), m = std::sqrt(-2.0 * std::log(u));
        spare = m * std::sin(6.283185307179586476925 * v); have_spare = true;
        return m * std::cos(6.283185307179586476925 * v);
    }
    uint64_t noise(double std) { return (uint64_t)(int64_t)std::llrint(gauss() * std * 18446744073709551616.0); }
};

// ---------------------------------------------------------------- AES-128
struct Aes {
    uint8_t sbox[256], inv[256];
    uint8_t rk[11][16];
    static uint8_t mul(uint8_t a, uint8_t b)
    {
        uint8_t r = 0;
        for (int i = 0; i < 8; i++) { if (b & 1) r ^= a; a = (a & 0x80) ? (uint8_t)((a << 1) ^ 0x1B) : (uint8_t)(a << 1); b >>= 1; }
        return r;
    }
    explicit Aes(const uint8_t key[16])
    {
        // S-box from the field inverse and the affine map
        uint8_t invf[256] = { 0 };
        for (int x = 1; x < 256; x++)
            for (int y = 1; y < 256; y++) if (mul((uint8_t)x, (uint8_t)y) == 1) { invf[x] = (uint8_t)y; break; }
        for (int x = 0; x < 256; x++) {
            uint8_t r = invf[x], s = invf[x];
            for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
            s ^= 0x63; sbox[x] = s; inv[s] = (uint8_t)x;
        }
        static const uint8_t rcon[11] = { 0, 1, 2, 4, 8, 16, 32, 64, 128, 0x1B, 0x36 };
        memcpy(rk[0], key, 16);
        for (int r = 1; r <= 10; r++) {
            const uint8_t *pk = rk[r - 1]; uint8_t *ck = rk[r];
            uint8_t t[4] = { (uint8_t)(sbox[pk[13]] ^ rcon[r]), sbox[pk[14]], sbox[pk[15]], sbox[pk[12]] };
            for (int i = 0; i < 4; i++) ck[i] = pk[i] ^ t[i];
            for (int i = 4; i < 16; i++) ck[i] = pk[i] ^ ck[i - 4];
        }
    }
    // byte index = 4*col + row (aes_manager.rs:146-161)
    static void shift_rows(uint8_t s[16], bool inverse)
    {
        uint8_t b[16]; memcpy(b, s, 16);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++)
            s[4 * c + r] = b[4 * ((inverse ? 4 + c - r : r + c) % 4) + r];
    }
    static void mix(uint8_t s[16], bool inverse)
    {
        uint8_t b[16]; memcpy(b, s, 16);
        const uint8_t f[4] = { (uint8_t)(inverse ? 14 : 2), (uint8_t)(inverse ? 11 : 3), (uint8_t)(inverse ? 13 : 1), (uint8_t)(inverse ? 9 : 1) };
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) {
            uint8_t v = 0;
            for (int i = 0; i < 4; i++) v ^= mul(b[4 * c + (r + i) % 4], f[i]);
            s[4 * c + r] = v;
        }
    }
    void encrypt(const uint8_t in[16], uint8_t out[16]) const
    {
        uint8_t s[16];
        for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[0][i];
        for (int r = 1; r <= 10; r++) {
            for (int i = 0; i < 16; i++) s[i] = sbox[s[i]];
            shift_rows(s, false);
            if (r < 10) mix(s, false);
            for (int i = 0; i < 16; i++) s[i] ^= rk[r][i];
        }
        memcpy(out, s, 16);
    }
    void decrypt(const uint8_t in[16], uint8_t out[16]) const
    {
        uint8_t s[16];
        for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[10][i];
        for (int r = 9; r >= 0; r--) {
            shift_rows(s, true);
            for (int i = 0; i < 16; i++) s[i] = inv[s[i]] ^ rk[r][i];
            if (r > 0) mix(s, true);
        }
        memcpy(out, s, 16);
    }
};

// ---------------------------------------------------------------- GLWE encryption
struct Dims { uint32_t N, k; };

// out += (sub ? -1 : 1) * a * s for a 0/1 secret polynomial s  (mod X^N + 1)
void mul_binary_acc(uint32_t N, const uint64_t *a, const uint64_t *s, uint64_t *out, bool sub)
{
    std::vector<uint64_t> ext(2 * N);                 // ext[j] = a[j], ext[N + j] = -a[j]
    for (uint32_t j = 0; j < N; j++) { ext[N + j] = a[j]; ext[j] = (uint64_t)0 - a[j]; }
    // coefficient d of a * X^i = ext[N + d - i]
    for (uint32_t i = 0; i < N; i++) {
        if (!s[i]) continue;
        const uint64_t *e = ext.data() + (N - i);
        if (sub) for (uint32_t d = 0; d < N; d++) out[d] -= e[d];
        else     for (uint32_t d = 0; d < N; d++) out[d] += e[d];
    }
}

void glwe_encrypt(Dims dm, ChaCha &rng, const uint64_t *sk, const uint64_t *pt, double std, uint64_t *out)
{
    uint64_t *b = out + (size_t)dm.k * dm.N;
    for (uint32_t j = 0; j < dm.N; j++) b[j] = (pt ? pt[j] : 0) + rng.noise(std);
    for (uint32_t q = 0; q < dm.k; q++) {
        uint64_t *a = out + (size_t)q * dm.N;
        for (uint32_t j = 0; j < dm.N; j++) a[j] = rng.u64();
        mul_binary_acc(dm.N, a, sk + (size_t)q * dm.N, b, false);
    }
}

// GGSW of a small signed polynomial message (tfhe encrypt_constant_ggsw_ciphertext generalised):
// row (lev, r < k) encrypts -m s_r 2^(64 - B lev); row (lev, k) encrypts m 2^(64 - B lev)
void ggsw_encrypt(const tx_params &p, ChaCha &rng, const uint64_t *sk, const std::vector<int64_t> &m, uint32_t B,
                  uint32_t l, double std, uint64_t *out)
{
    const uint32_t N = p.N, k = p.k, K = k + 1;
    std::vector<uint64_t> pt(N), ms(N);
    for (uint32_t lev = 0; lev < l; lev++) {
        const uint32_t sh = 64 - B * (lev + 1);
        for (uint32_t r = 0; r < K; r++) {
            if (r < k) {
                std::fill(ms.begin(), ms.end(), 0);
                for (uint32_t i = 0; i < N; i++) {              // ms = m * s_r
                    if (!m[i]) continue;
                    const uint64_t mi = (uint64_t)m[i];
                    for (uint32_t j = 0; j < N; j++) {
                        if (!sk[(size_t)r * N + j]) continue;
                        uint32_t d = i + j;
                        if (d >= N) ms[d - N] -= mi; else ms[d] += mi;
                    }
                }
                for (uint32_t j = 0; j < N; j++) pt[j] = ((uint64_t)0 - ms[j]) << sh;
            } else {
                for (uint32_t j = 0; j < N; j++) pt[j] = ((uint64_t)m[j]) << sh;
            }
            glwe_encrypt({ N, k }, rng, sk, pt.data(), std, out + ((size_t)lev * K + r) * K * N);
        }
    }
}

void poly_auto(uint32_t N, const uint64_t *in, uint32_t t, uint64_t *out)
{   // X -> X^t (cbs_lib/src/utils.rs:475-490)
    for (uint32_t i = 0; i < N; i++) {
        uint64_t it = (uint64_t)i * t;
        out[it % N] = ((it / N) & 1) ? (uint64_t)0 - in[i] : in[i];
    }
}

uint32_t ilog2(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) l++; return l; }

void emit_lut(const tx_params &p, ChaCha &rng, const uint64_t *sk, bool encrypted, const uint8_t table[256], uint64_t *dst)
{   // data_struct.rs:130-151, aes_he.rs:865-871: coefficient i -> bit (4 acc + i/256) of table[i % 256] at 2^63
    const uint32_t N = p.N, k = p.k, K = k + 1;
    std::vector<uint64_t> pt(N);
    for (uint32_t a = 0; a < 2; a++) {
        uint64_t *g = dst + (size_t)a * K * N;
        for (uint32_t i = 0; i < N; i++) pt[i] = (uint64_t)((table[i % 256] >> (a * (N / 256) + i / 256)) & 1) << 63;
        if (encrypted) glwe_encrypt({ N, k }, rng, sk, pt.data(), p.glwe_std, g);
        else { memset(g, 0, sizeof(uint64_t) * (size_t)k * N); memcpy(g + (size_t)k * N, pt.data(), sizeof(uint64_t) * N); }
    }
}

}  // namespace

extern "C" {

size_t txc_bsk_words(const tx_params *p)  { size_t K = p->k + 1; return (size_t)p->n * p->pbs_level * K * K * p->N; }
size_t txc_ksk_words(const tx_params *p)  { return (size_t)(p->k * p->N / p->ks_N) * p->ks_level * (p->n / p->ks_N + 1) * p->ks_N; }
size_t txc_auto_words(const tx_params *p) { size_t K = p->k + 1; return (size_t)ilog2(p->N) * p->k * p->auto_level * K * p->N; }
size_t txc_ss_words(const tx_params *p)   { size_t K = p->k + 1; return (size_t)p->k * p->ss_level * K * K * p->N; }
size_t txc_lut_words(const tx_params *p)  { size_t K = p->k + 1; return (size_t)(9 * 4 + 1) * 16 * 2 * K * p->N; }
size_t txc_ctr_lut_words(const tx_params *p) { size_t K = p->k + 1; return (size_t)10 * 3 * 16 * 2 * K * p->N; }

int txc_keygen(const tx_params *pp, uint64_t seed, uint64_t *lwe_sk, uint64_t *glwe_sk, uint64_t *bsk, uint64_t *ksk,
               uint64_t *auto_keys, uint64_t *ss_key)
{
    const tx_params &p = *pp;
    const uint32_t N = p.N, k = p.k, K = k + 1, n = p.n;
    if (n % p.ks_N || (k * N) % p.ks_N) return -1;
    ChaCha secret(seed, 1), enc(seed, 2);
    if (!secret.ok || !enc.ok) return TXC_ERR_ENTROPY;
    // cbs_lib/src/keygen.rs:208-217
    for (uint32_t i = 0; i < n; i++) lwe_sk[i] = secret.u64() & 1;
    for (uint32_t i = 0; i < k * N; i++) glwe_sk[i] = secret.u64() & 1;
    // keygen.rs:219-227: bsk[i] = GGSW(s_i) under the GLWE key, GLWE noise
    std::vector<int64_t> m(N, 0);
    for (uint32_t i = 0; i < n; i++) {
        m[0] = (int64_t)lwe_sk[i];
        ggsw_encrypt(p, enc, glwe_sk, m, p.pbs_base_log, p.pbs_level, p.glwe_std, bsk + (size_t)i * p.pbs_level * K * K * N);
    }
    // keygen.rs:230-240, glwe_keyswitch.rs:183-218: GLWE-form KSK at ks_N from the big to the small key, LWE noise
    {
        const uint32_t Nc = p.ks_N, kin = k * N / Nc, kout = n / Nc, Kout = kout + 1;
        std::vector<uint64_t> pt(Nc);
        for (uint32_t q = 0; q < kin; q++)
            for (uint32_t lev = 0; lev < p.ks_level; lev++) {
                const uint32_t sh = 64 - p.ks_base_log * (lev + 1);
                for (uint32_t j = 0; j < Nc; j++) pt[j] = ((uint64_t)0 - glwe_sk[(size_t)q * Nc + j]) << sh;
                glwe_encrypt({ Nc, kout }, enc, lwe_sk, pt.data(), p.lwe_std, ksk + ((size_t)q * p.ks_level + lev) * Kout * Nc);
            }
    }
    // automorphism.rs:62-120,152-178: KSK from S(X^t) to S(X), t = N/2^(i-1) + 1
    {
        std::vector<uint64_t> before(N), pt(N);
        const uint32_t L = ilog2(N);
        for (uint32_t i = 1; i <= L; i++) {
            const uint32_t t = N / (1u << (i - 1)) + 1;
            for (uint32_t q = 0; q < k; q++) {
                poly_auto(N, glwe_sk + (size_t)q * N, t, before.data());
                for (uint32_t lev = 0; lev < p.auto_level; lev++) {
                    const uint32_t sh = 64 - p.auto_base_log * (lev + 1);
                    for (uint32_t j = 0; j < N; j++) pt[j] = ((uint64_t)0 - before[j]) << sh;
                    glwe_encrypt({ N, k }, enc, glwe_sk, pt.data(), p.glwe_std,
                                 auto_keys + ((((size_t)(i - 1) * k + q) * p.auto_level) + lev) * K * N);
                }
            }
        }
    }
    // ggsw_conv.rs:15-77: ss_key[i] = GGSW(-s_i)
    for (uint32_t i = 0; i < k; i++) {
        for (uint32_t j = 0; j < N; j++) m[j] = -(int64_t)glwe_sk[(size_t)i * N + j];
        ggsw_encrypt(p, enc, glwe_sk, m, p.ss_base_log, p.ss_level, p.glwe_std, ss_key + (size_t)i * p.ss_level * K * K * N);
    }
    return 0;
}

int txc_make_round_luts(const tx_params *pp, uint64_t seed, const uint8_t aes_key[16], const uint64_t *glwe_sk, uint64_t *luts)
{
    return txc_make_round_luts_ex(pp, seed, aes_key, glwe_sk, 0, luts);
}

int txc_make_round_luts_ex(const tx_params *pp, uint64_t seed, const uint8_t aes_key[16], const uint64_t *glwe_sk,
                           unsigned flags, uint64_t *luts)
{
    const tx_params &p = *pp;
    Aes aes(aes_key);
    ChaCha enc(seed, 3);
    if (!enc.ok) return TXC_ERR_ENTROPY;
    // The reference encrypts only the fused round-10/9 tables and ships rounds 8..0 as TRIVIAL (mask 0,
    // noiseless) GLWEs (data_struct.rs:146-151, :258-263): the table bits are then readable by the server,
    // any one round key falls out of them, and the AES-128 key schedule is invertible.  Here every table is a
    // real GLWE encryption unless the caller asks for the reference's form explicitly.
    const bool trivial = (flags & TXC_LUTS_REFERENCE_TRIVIAL) != 0;
    const size_t gl = (size_t)(p.k + 1) * p.N;
    static const uint8_t mults[4] = { 9, 11, 13, 14 };
    uint8_t base[256], tab[256];
    for (int set = 0; set < 9; set++)
        for (int m = 0; m < 4; m++)
            for (int j = 0; j < 16; j++) {
                if (set == 0) {          // aes_manager.rs:338-396 (AddRK10, InvShiftRows, InvSubBytes, AddRK9)
                    const int c = j / 4, r = j % 4, pre = 4 * ((4 + c - r) % 4) + r;
                    for (int x = 0; x < 256; x++) base[x] = aes.inv[x ^ aes.rk[10][pre]] ^ aes.rk[9][j];
                } else {                 // aes_manager.rs:398-420, round 9 - set
                    for (int x = 0; x < 256; x++) base[x] = aes.inv[x] ^ aes.rk[9 - set][j];
                }
                for (int x = 0; x < 256; x++) tab[x] = Aes::mul(base[x], mults[m]);
                emit_lut(p, enc, glwe_sk, set == 0 || !trivial, tab, luts + ((((size_t)set * 4 + m) * 16 + j) * 2) * gl);
            }
    for (int j = 0; j < 16; j++) {       // aes_manager.rs:422-432
        for (int x = 0; x < 256; x++) tab[x] = aes.inv[x] ^ aes.rk[0][j];
        emit_lut(p, enc, glwe_sk, !trivial, tab, luts + (((size_t)9 * 4 * 16 + j) * 2) * gl);
    }
    return 0;
}

int txc_make_ctr_luts(const tx_params *pp, uint64_t seed, const uint8_t aes_key[16], const uint64_t *glwe_sk, uint64_t *luts)
{   // forward cipher tables: set s holds m * S[x ^ rk_s] (m = 1,2,3); set 9: S[x ^ rk9] ^ rk10[ShiftRows(j)]
    const tx_params &p = *pp;
    Aes aes(aes_key);
    ChaCha enc(seed, 4);
    if (!enc.ok) return TXC_ERR_ENTROPY;
    const size_t gl = (size_t)(p.k + 1) * p.N;
    memset(luts, 0, sizeof(uint64_t) * txc_ctr_lut_words(pp));
    uint8_t tab[256];
    for (int set = 0; set < 10; set++)
        for (int m = 0; m < 3; m++)
            for (int j = 0; j < 16; j++) {
                if (set == 9 && m > 0) continue;
                const int c = j / 4, r = j % 4, dst = 4 * ((4 + c - r) % 4) + r;
                for (int x = 0; x < 256; x++) {
                    const uint8_t s = aes.sbox[x ^ aes.rk[set][j]];
                    tab[x] = (set == 9) ? (uint8_t)(s ^ aes.rk[10][dst]) : Aes::mul(s, (uint8_t)(m + 1));
                }
                // every set is a real encryption: a trivial table would hand rk_set to the server in the clear
                emit_lut(p, enc, glwe_sk, true, tab, luts + ((((size_t)set * 3 + m) * 16 + j) * 2) * gl);
            }
    return 0;
}

int txc_decrypt_bits(const tx_params *p, const uint64_t *sk, const uint64_t *lwe, size_t count, uint8_t *bits)
{   // help_fun.rs:24-43
    const size_t dim = (size_t)p->k * p->N;
    for (size_t c = 0; c < count; c++) {
        const uint64_t *x = lwe + c * (dim + 1);
        uint64_t ph = x[dim];
        for (size_t j = 0; j < dim; j++) if (sk[j]) ph -= x[j];
        const uint64_t rounding = (ph & ((uint64_t)1 << 62)) << 1;
        bits[c] = (uint8_t)(((ph + rounding) >> 63) & 1);
    }
    return 0;
}

int txc_encrypt_bits(const tx_params *p, uint64_t seed, const uint64_t *sk, const uint8_t *bits, size_t count, uint64_t *lwe)
{
    const size_t dim = (size_t)p->k * p->N;
    ChaCha enc(seed, 5);
    if (!enc.ok) return TXC_ERR_ENTROPY;
    for (size_t c = 0; c < count; c++) {
        uint64_t *x = lwe + c * (dim + 1);
        uint64_t b = ((uint64_t)(bits[c] & 1) << 63) + enc.noise(p->glwe_std);
        for (size_t j = 0; j < dim; j++) { x[j] = enc.u64(); if (sk[j]) b += x[j]; }
        x[dim] = b;
    }
    return 0;
}

int txc_bits_to_u16(const uint8_t *bits, size_t nbits, uint16_t *out)
{   // client_postprocess_aes_decryption.rs:24-29
    if (nbits % 16) return -1;
    for (size_t i = 0; i < nbits / 16; i++) {
        uint16_t v = 0;
        for (int b = 0; b < 16; b++) v = (uint16_t)((v << 1) | (bits[i * 16 + b] & 1));
        out[i] = v;
    }
    return 0;
}

void txc_aes128_ecb_encrypt(const uint8_t key[16], const uint8_t *in, size_t nblocks, uint8_t *out)
{ Aes a(key); for (size_t i = 0; i < nblocks; i++) a.encrypt(in + 16 * i, out + 16 * i); }
void txc_aes128_ecb_decrypt(const uint8_t key[16], const uint8_t *in, size_t nblocks, uint8_t *out)
{ Aes a(key); for (size_t i = 0; i < nblocks; i++) a.decrypt(in + 16 * i, out + 16 * i); }
void txc_aes128_ctr_crypt(const uint8_t key[16], const uint8_t iv[16], const uint8_t *in, size_t nbytes, uint8_t *out)
{
    Aes a(key);
    uint8_t ctr[16], ks[16];
    memcpy(ctr, iv, 16);
    for (size_t off = 0; off < nbytes; off += 16) {
        a.encrypt(ctr, ks);
        for (size_t i = 0; i < 16 && off + i < nbytes; i++) out[off + i] = in[off + i] ^ ks[i];
        for (int i = 15; i >= 0; i--) if (++ctr[i]) break;      // 128-bit big-endian counter
    }
}

}  // extern "C"
