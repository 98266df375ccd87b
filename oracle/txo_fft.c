/*
 * txo_fft.c -- ORACLE (test infrastructure only; see txoracle.h).
 *
 * Negacyclic f64 transform restating what tfhe-rs 0.5.4 `Fft::forward_as_integer /
 * forward_as_torus / add_backward_as_torus` (concrete-fft 0.4.1; not in /root/reference,
 * call sites cbs_lib/src/fourier_glwe_keyswitch.rs:301,331, fourier_glev_ciphertext.rs:155)
 * compute: fold the N real coefficients into N/2 complex points (a[j] + i a[j+N/2]), multiply by
 * the 2N-th root twist, complex FFT of size N/2.  Pointwise products in that domain are
 * negacyclic products.  The butterfly network (radix-8 x 3 for N = 1024, radix-2 for smaller N),
 * the output ordering ("F-layout") and the exact operation order are the ones the HIP kernels
 * use, so integer results match the GPU bit for bit.  No fp contraction: compile with
 * -ffp-contract=off; every fused operation is an explicit fma().
 */
#include "txoracle.h"
#include "txo_internal.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

typedef struct { double re, im; } cplx;

static inline cplx cmul(cplx x, cplx w)
{   /* x * w */
    cplx r;
    double t = x.im * w.im;
    r.re = fma(x.re, w.re, -t);
    double u = x.im * w.re;
    r.im = fma(x.re, w.im, u);
    return r;
}
static inline cplx cmulc(cplx x, cplx w)
{   /* x * conj(w) */
    cplx r;
    double t = x.im * w.im;
    r.re = fma(x.re, w.re, t);
    double u = x.re * w.im;
    r.im = fma(x.im, w.re, -u);
    return r;
}
static inline cplx cadd(cplx a, cplx b) { cplx r = { a.re + b.re, a.im + b.im }; return r; }
static inline cplx csub(cplx a, cplx b) { cplx r = { a.re - b.re, a.im - b.im }; return r; }

#define SQRT_HALF 0.70710678118654752440

/* 8-point DFT, kernel exp(s*2*pi*i*n*k/8), s = +1 (fwd) or -1 (inv); in place, natural order. */
static inline void dft8(cplx v[8], int s)
{
    cplx a0 = cadd(v[0], v[4]), b0 = csub(v[0], v[4]);
    cplx a1 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]);
    cplx a2 = cadd(v[2], v[6]), b2 = csub(v[2], v[6]);
    cplx a3 = cadd(v[3], v[7]), b3 = csub(v[3], v[7]);
    /* odd branch: t1 = w8 b1, t2 = s i b2, t3 = w8^3 b3, w8 = (1 + s i)/sqrt2.  The factor sqrt(1/2) of t1 and t3 is
     * not multiplied out: u1 = sqrt2 t1, u3 = sqrt2 t3, and it enters the last butterflies as a fused multiply-add */
    cplx u1, t2, u3;
    if (s > 0) {
        u1.re = b1.re - b1.im;    u1.im = b1.re + b1.im;
        t2.re = -b2.im;           t2.im = b2.re;
        u3.re = -(b3.re + b3.im); u3.im = b3.re - b3.im;
    } else {
        u1.re = b1.re + b1.im;    u1.im = b1.im - b1.re;
        t2.re = b2.im;            t2.im = -b2.re;
        u3.re = b3.im - b3.re;    u3.im = -(b3.re + b3.im);
    }
    /* two 4-point DFTs */
    cplx p0, q0, p1, q1, d;
    /* even outputs from a */
    p0 = cadd(a0, a2); q0 = csub(a0, a2); p1 = cadd(a1, a3); d = csub(a1, a3);
    if (s > 0) { q1.re = -d.im; q1.im = d.re; } else { q1.re = d.im; q1.im = -d.re; }
    v[0] = cadd(p0, p1); v[4] = csub(p0, p1); v[2] = cadd(q0, q1); v[6] = csub(q0, q1);
    /* odd outputs: p0 +- sqrt(1/2) (u1 + u3), q0 +- sqrt(1/2) (+-i)(u1 - u3) */
    p0 = cadd(b0, t2); q0 = csub(b0, t2); p1 = cadd(u1, u3); d = csub(u1, u3);
    if (s > 0) { q1.re = -d.im; q1.im = d.re; } else { q1.re = d.im; q1.im = -d.re; }
    v[1].re = fma(p1.re, SQRT_HALF, p0.re);  v[1].im = fma(p1.im, SQRT_HALF, p0.im);
    v[5].re = fma(-p1.re, SQRT_HALF, p0.re); v[5].im = fma(-p1.im, SQRT_HALF, p0.im);
    v[3].re = fma(q1.re, SQRT_HALF, q0.re);  v[3].im = fma(q1.im, SQRT_HALF, q0.im);
    v[7].re = fma(-q1.re, SQRT_HALF, q0.re); v[7].im = fma(-q1.im, SQRT_HALF, q0.im);
}

/* ---------------- tables ---------------- */
/* The twist e^{i pi j/1024}, j = 64 n1 + n2, is split: its n1 part C16[n1] = e^{i pi n1/16} multiplies the
 * inputs of pass A (eight constants, no table), its n2 part is folded into the pass-A twiddles:
 * rows 0-7: e^{i pi n2 (1 + 4 k1)/1024} (k1 = 0..7); rows 8-14: pass B (c = 1..7). */
static cplx TW1024[15][64];
static cplx C16[8];
static int  tables_ready = 0;

typedef struct { uint32_t N; cplx *twist; cplx *w; } small_tab;   /* generic radix-2 tables */
static small_tab SMALL[16];

static cplx unit(long double num, long double den)
{   /* exp(i*pi*num/den) */
    const long double PI = 3.14159265358979323846264338327950288L;
    cplx r;
    long double a = PI * num / den;
    r.re = (double)cosl(a);
    r.im = (double)sinl(a);
    return r;
}

static void build_tables(void)
{
    if (tables_ready) return;
    for (int m = 0; m < 8; m++) C16[m] = unit(m, 16);
    for (int k1 = 0; k1 < 8; k1++)
        for (int t = 0; t < 64; t++) TW1024[k1][t] = unit(t * (1 + 4 * k1), 1024);
    for (int c = 1; c < 8; c++)
        for (int t = 0; t < 64; t++) TW1024[7 + c][t] = unit(2 * ((t & 7) * c), 64);
    tables_ready = 1;
}
const double *txo_twiddles_1024(void) { build_tables(); return (const double *)TW1024; }
const double *txo_twist_consts_1024(void) { build_tables(); return (const double *)C16; }

static const small_tab *small_tables(uint32_t N)
{
    int lg = 0; while ((1u << lg) < N) lg++;
    small_tab *t = &SMALL[lg];
    if (t->N == N) return t;
    uint32_t H = N / 2;
    cplx *tw = (cplx *)malloc(sizeof(cplx) * H), *w = (cplx *)malloc(sizeof(cplx) * H);
    for (uint32_t j = 0; j < H; j++) { tw[j] = unit(j, N); w[j] = unit(2 * j, H); }
    t->twist = tw; t->w = w; t->N = N;
    return t;
}

/* ---------------- N = 1024 ---------------- */
static void fwd1024(const double *a, cplx *out)
{
    cplx c[512], z[512];
    /* pass A: lane n2, regs n1 -> k1 */
    for (int n2 = 0; n2 < 64; n2++) {
        cplx v[8];
        for (int n1 = 0; n1 < 8; n1++) {
            cplx x = { a[64 * n1 + n2], a[64 * n1 + n2 + 512] };
            v[n1] = n1 ? cmul(x, C16[n1]) : x;
        }
        dft8(v, +1);
        for (int k1 = 0; k1 < 8; k1++) z[k1 * 64 + n2] = cmul(v[k1], TW1024[k1][n2]);
    }
    /* pass B: lane (k1,b), regs a -> c */
    for (int k1 = 0; k1 < 8; k1++)
        for (int b = 0; b < 8; b++) {
            cplx v[8];
            for (int a8 = 0; a8 < 8; a8++) v[a8] = z[k1 * 64 + 8 * a8 + b];
            dft8(v, +1);
            for (int cc = 1; cc < 8; cc++) v[cc] = cmul(v[cc], TW1024[7 + cc][b]);
            for (int cc = 0; cc < 8; cc++) c[k1 * 64 + b * 8 + cc] = v[cc];   /* y[k1][b][c] */
        }
    /* pass C: lane (k1,c), regs b -> d */
    for (int k1 = 0; k1 < 8; k1++)
        for (int cc = 0; cc < 8; cc++) {
            cplx v[8];
            for (int b = 0; b < 8; b++) v[b] = c[k1 * 64 + b * 8 + cc];
            dft8(v, +1);
            for (int d = 0; d < 8; d++) out[d * 64 + 8 * k1 + cc] = v[d];
        }
}

static void inv1024(const cplx *in, double *a)
{
    cplx y[512], z[512];
    for (int k1 = 0; k1 < 8; k1++)
        for (int cc = 0; cc < 8; cc++) {
            cplx v[8];
            for (int d = 0; d < 8; d++) v[d] = in[d * 64 + 8 * k1 + cc];
            dft8(v, -1);
            for (int b = 0; b < 8; b++) y[k1 * 64 + b * 8 + cc] = v[b];
        }
    for (int k1 = 0; k1 < 8; k1++)
        for (int b = 0; b < 8; b++) {
            cplx v[8];
            v[0] = y[k1 * 64 + b * 8];
            for (int cc = 1; cc < 8; cc++) v[cc] = cmulc(y[k1 * 64 + b * 8 + cc], TW1024[7 + cc][b]);
            dft8(v, -1);
            for (int a8 = 0; a8 < 8; a8++) z[k1 * 64 + 8 * a8 + b] = v[a8];
        }
    for (int n2 = 0; n2 < 64; n2++) {
        cplx v[8];
        for (int k1 = 0; k1 < 8; k1++) v[k1] = cmulc(z[k1 * 64 + n2], TW1024[k1][n2]);
        dft8(v, -1);
        for (int n1 = 0; n1 < 8; n1++) {
            int j = 64 * n1 + n2;
            cplx r = n1 ? cmulc(v[n1], C16[n1]) : v[n1];
            a[j] = r.re; a[j + 512] = r.im;
        }
    }
}

/* ---------------- generic radix-2 (N = 256 key switch) ----------------
 * forward: DIF, natural in -> bit-reversed out; inverse: DIT, bit-reversed in -> natural out. */
static void fwd_small(uint32_t N, const double *a, cplx *x)
{
    const small_tab *t = small_tables(N);
    uint32_t H = N / 2;
    for (uint32_t j = 0; j < H; j++) { cplx v = { a[j], a[j + H] }; x[j] = cmul(v, t->twist[j]); }
    for (uint32_t span = H / 2; span >= 1; span >>= 1) {
        uint32_t step = H / (2 * span);
        for (uint32_t base = 0; base < H; base += 2 * span)
            for (uint32_t j = 0; j < span; j++) {
                cplx u = x[base + j], v = x[base + j + span];
                x[base + j] = cadd(u, v);
                x[base + j + span] = cmul(csub(u, v), t->w[j * step]);
            }
    }
}
static void inv_small(uint32_t N, const cplx *in, double *a)
{
    const small_tab *t = small_tables(N);
    uint32_t H = N / 2;
    cplx *x = (cplx *)txo_tls(1, sizeof(cplx) * H);
    memcpy(x, in, sizeof(cplx) * H);
    for (uint32_t span = 1; span < H; span <<= 1) {
        uint32_t step = H / (2 * span);
        for (uint32_t base = 0; base < H; base += 2 * span)
            for (uint32_t j = 0; j < span; j++) {
                cplx u = x[base + j], v = cmulc(x[base + j + span], t->w[j * step]);
                x[base + j] = cadd(u, v);
                x[base + j + span] = csub(u, v);
            }
    }
    for (uint32_t j = 0; j < H; j++) { cplx r = cmulc(x[j], t->twist[j]); a[j] = r.re; a[j + H] = r.im; }
}

/* ---------------- conversions ---------------- */
static inline uint64_t dbits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* round-half-even(x) mod 2^64, as the 7-operation sequence the GPU uses */
static inline uint64_t f64_to_torus(double x)
{
    double m  = x * 0x1p-64;
    double y  = m - nearbyint(m);                 /* in [-1/2, 1/2], exact */
    double t1 = fma(y, 0x1p64, 0x1.8p84);         /* ulp 2^32: low mantissa = round(z / 2^32) */
    double h  = t1 - 0x1.8p84;
    double lo = fma(y, 0x1p64, -h);               /* in [-2^31, 2^31], exact */
    double t2 = lo + 0x1.8p52;                    /* ulp 1 */
    uint64_t b1 = dbits(t1), b2 = dbits(t2);
    uint32_t rlo = (uint32_t)b2;
    uint32_t rhi = (uint32_t)b1 + (uint32_t)(b2 >> 32) - 0x43380000u;
    return ((uint64_t)rhi << 32) | rlo;
}

void txo_fft_forward_int(uint32_t N, const int64_t *in, double *out)
{
    build_tables();
    double *a = (double *)txo_tls(0, sizeof(double) * N);
    for (uint32_t j = 0; j < N; j++) a[j] = (double)in[j];
    if (N == 1024) fwd1024(a, (cplx *)out); else fwd_small(N, a, (cplx *)out);
}

/* key-side transform: torus value as signed 64-bit -> f64 (one rounding), transform, then scale
 * by 2/N exactly so that the unnormalised inverse yields the true product */
void txo_fft_forward_torus_scaled(uint32_t N, const uint64_t *in, double *out)
{
    build_tables();
    double *a = (double *)txo_tls(0, sizeof(double) * N);
    for (uint32_t j = 0; j < N; j++) a[j] = (double)(int64_t)in[j];
    if (N == 1024) fwd1024(a, (cplx *)out); else fwd_small(N, a, (cplx *)out);
    double sc = 2.0 / (double)N;
    for (uint32_t j = 0; j < N; j++) out[j] *= sc;
}

void txo_fft_inverse_add(uint32_t N, const double *in, uint64_t *acc)
{
    build_tables();
    double *a = (double *)txo_tls(0, sizeof(double) * N);
    if (N == 1024) inv1024((const cplx *)in, a); else inv_small(N, (const cplx *)in, a);
    for (uint32_t j = 0; j < N; j++) acc[j] += f64_to_torus(a[j]);
}

/* exact reference product: out = small * big mod (X^N+1, 2^64) */
void txo_poly_mul_exact(uint32_t N, const int64_t *small, const uint64_t *big, uint64_t *out)
{
    for (uint32_t i = 0; i < N; i++) out[i] = 0;
    for (uint32_t i = 0; i < N; i++) {
        uint64_t s = (uint64_t)small[i];
        if (!s) continue;
        for (uint32_t j = 0; j < N; j++) {
            uint32_t d = i + j;
            uint64_t pr = s * big[j];
            if (d < N) out[d] += pr; else out[d - N] -= pr;
        }
    }
}

/* ---------------- per-thread scratch ---------------- */
#define TXO_TLS_SLOTS 24
static __thread void  *tls_ptr[TXO_TLS_SLOTS];
static __thread size_t tls_cap[TXO_TLS_SLOTS];
void *txo_tls(int slot, size_t bytes)
{
    if (tls_cap[slot] < bytes) {
        free(tls_ptr[slot]);
        tls_ptr[slot] = malloc(bytes);
        tls_cap[slot] = bytes;
    }
    return tls_ptr[slot];
}
