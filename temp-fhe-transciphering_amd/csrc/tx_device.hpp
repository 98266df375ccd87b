// tx_device.hpp -- wave-level building blocks for the gfx950 kernels.
//
// Execution model: ONE 64-lane wavefront owns one polynomial-sized unit of work (one GLWE
// ciphertext, one key-switch, ...).  A polynomial of N = 1024 torus coefficients lives in the
// wave's registers as 16 values per lane: lane t holds coefficients t + 64 m, m = 0..15.  For the
// f64 negacyclic transform (tfhe-rs fft64 semantics, see oracle/txo_fft.c) coefficients j and
// j + 512 form one complex point, so lane t holds the 8 complex points t + 64 m -- exactly the
// operand set of the first radix-8 butterfly.  512 = 8 x 8 x 8: three in-register radix-8 passes
// separated by two 8x8 transposes through a padded per-wave LDS tile.  The transform output stays
// in "F-layout" (lane 8 k1 + c, register d  <->  frequency k1 + 8 c + 64 d); keys are stored in
// that layout as [d][lane] complex so every key load is one coalesced 1 KiB wave access.
//
// Floating point: every fused multiply-add is explicit (__builtin_fma) and the translation unit
// is compiled with -ffp-contract=off, so results are bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace txd {

typedef double2 cplx;

constexpr int N = 1024;           // polynomial size            (aes_instances.rs:79)
constexpr int H = 512;            // complex points
constexpr int KP = 3;             // k + 1 polynomials per GLWE (aes_instances.rs:80)
constexpr int GL = KP * N;        // words per GLWE
constexpr int BIG = 2 * N + 1;    // words per big LWE
constexpr int TW_ROWS = 15;       // 8 pass-A rows (lane part of the twist folded in) + 7 pass-B rows of 64 complex
constexpr int XPITCH = 65;        // padded pitch (complex) of the per-wave transpose tile
constexpr int SCRATCH_CPLX = 8 * XPITCH;   // 8320 bytes per wave

__device__ __forceinline__ cplx cmul(cplx x, cplx w)
{
    double t = x.y * w.y;
    double u = x.y * w.x;
    return make_double2(__builtin_fma(x.x, w.x, -t), __builtin_fma(x.x, w.y, u));
}
__device__ __forceinline__ cplx cmulc(cplx x, cplx w)   // x * conj(w)
{
    double t = x.y * w.y;
    double u = x.x * w.y;
    return make_double2(__builtin_fma(x.x, w.x, t), __builtin_fma(x.y, w.x, -u));
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }

#define TXD_SQRT_HALF 0.70710678118654752440

// register part of the twist: exp(i pi m / 16), m = 0..7 (oracle C16[]); m is a compile-time constant at
// every call site (fully unrolled loops), so the values become literals
#include "tx_fft_consts.inc"
__device__ __forceinline__ cplx c16(int m)
{
    switch (m) {
#define TXD_C16_CASE(M, RE, IM) case M: return make_double2(RE, IM);
        TXD_C16_TABLE(TXD_C16_CASE)
#undef TXD_C16_CASE
    }
    return make_double2(1.0, 0.0);
}

// 8-point DFT in registers, kernel exp(S 2 pi i n k / 8).  Same operand pairing as oracle dft8().
template <int S>
__device__ __forceinline__ void dft8(cplx (&v)[8])
{
    cplx a0 = cadd(v[0], v[4]), b0 = csub(v[0], v[4]);
    cplx a1 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]);
    cplx a2 = cadd(v[2], v[6]), b2 = csub(v[2], v[6]);
    cplx a3 = cadd(v[3], v[7]), b3 = csub(v[3], v[7]);
    // odd branch: the factor sqrt(1/2) of w8 b1 and w8^3 b3 is not multiplied out (u1, u3 below are sqrt2 times
    // those); it enters the last butterflies as a fused multiply-add -- four operations less per 8-point DFT
    cplx u1, t2, u3;
    if (S > 0) {
        u1 = make_double2(b1.x - b1.y, b1.x + b1.y);
        t2 = make_double2(-b2.y, b2.x);
        u3 = make_double2(-(b3.x + b3.y), b3.x - b3.y);
    } else {
        u1 = make_double2(b1.x + b1.y, b1.y - b1.x);
        t2 = make_double2(b2.y, -b2.x);
        u3 = make_double2(b3.y - b3.x, -(b3.x + b3.y));
    }
    cplx p0, q0, p1, q1, d;
    p0 = cadd(a0, a2); q0 = csub(a0, a2); p1 = cadd(a1, a3); d = csub(a1, a3);
    q1 = (S > 0) ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
    v[0] = cadd(p0, p1); v[4] = csub(p0, p1); v[2] = cadd(q0, q1); v[6] = csub(q0, q1);
    p0 = cadd(b0, t2); q0 = csub(b0, t2); p1 = cadd(u1, u3); d = csub(u1, u3);
    q1 = (S > 0) ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
    v[1] = make_double2(__builtin_fma(p1.x, TXD_SQRT_HALF, p0.x), __builtin_fma(p1.y, TXD_SQRT_HALF, p0.y));
    v[5] = make_double2(__builtin_fma(-p1.x, TXD_SQRT_HALF, p0.x), __builtin_fma(-p1.y, TXD_SQRT_HALF, p0.y));
    v[3] = make_double2(__builtin_fma(q1.x, TXD_SQRT_HALF, q0.x), __builtin_fma(q1.y, TXD_SQRT_HALF, q0.y));
    v[7] = make_double2(__builtin_fma(-q1.x, TXD_SQRT_HALF, q0.x), __builtin_fma(-q1.y, TXD_SQRT_HALF, q0.y));
}

// Order LDS traffic of one wave: the hardware executes a wave's DS instructions in order, so a
// compiler-level fence is all that is needed between a tile write and the dependent tile read.
// all loads issued before this point stay before it, all users after it: the loads go out back to
// back and the (automatic, counted) waits land at the first uses
__device__ __forceinline__ void issue_fence() { __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8x8 transposes through the per-wave tile (pitch 65 complex => conflict-free b128 stores/loads).
//   xchg_hi: element (lane = 8a+b, reg r)  ->  (lane = 8r+b, reg a)
//   xchg_lo: element (lane = 8g+b, reg r)  ->  (lane = 8g+r, reg b)
#ifdef TXD_XCHG_HI_VALU
// xchg_hi without LDS: the 3-bit register index is swapped with lane bits 5, 4, 3 one bit at a time --
// v_permlane32_swap (bit 5), v_permlane16_swap (bit 4), and a pair of bank-masked row_ror:8 DPP moves (bit 3).
// The LDS store path (79 B/clk per CU, shared by the four waves) is the scarcer resource than VALU issue.
__device__ __forceinline__ void xchg_hi(cplx (&v)[8], cplx *, int)
{
#ifdef TXD_ABL_NOXCHG
    return;
#endif
    uint32_t d[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint64_t a = (uint64_t)__double_as_longlong(v[r].x), b = (uint64_t)__double_as_longlong(v[r].y);
        d[r][0] = (uint32_t)a; d[r][1] = (uint32_t)(a >> 32); d[r][2] = (uint32_t)b; d[r][3] = (uint32_t)(b >> 32);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {                       // register bit 2 <-> lane bit 5
            auto t = __builtin_amdgcn_permlane32_swap(d[r][k], d[r + 4][k], false, false);
            d[r][k] = t[0]; d[r + 4][k] = t[1];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {                       // register bit 1 <-> lane bit 4
            const int r = (q & 1) | ((q & 2) << 1);
            auto t = __builtin_amdgcn_permlane16_swap(d[r][k], d[r + 2][k], false, false);
            d[r][k] = t[0]; d[r + 2][k] = t[1];
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2) {                    // register bit 0 <-> lane bit 3
            const uint32_t a = d[r][k], b = d[r + 1][k];
            d[r + 1][k] = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xF, 0x3, false);    // lanes 0-7 of a row <- a[lane + 8]
            d[r][k] = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false);        // lanes 8-15 <- b[lane - 8]
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
        v[r] = make_double2(__longlong_as_double((long long)(((uint64_t)d[r][1] << 32) | d[r][0])),
                            __longlong_as_double((long long)(((uint64_t)d[r][3] << 32) | d[r][2])));
    issue_fence();
}
#else
__device__ __forceinline__ void xchg_hi(cplx (&v)[8], cplx *tile, int lane)
{
#ifdef TXD_ABL_NOXCHG
    return;
#endif
    const int a = lane >> 3, b = lane & 7;
    // no fence in front: the stores may sink into the producing butterflies as soon as each value
    // is final (cplx-typed accesses to the same tile keep their program order by aliasing rules)
#pragma unroll
    for (int r = 0; r < 8; r++) tile[a * XPITCH + 8 * r + b] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = tile[r * XPITCH + lane];
    issue_fence();
}
#endif
__device__ __forceinline__ void xchg_lo(cplx (&v)[8], cplx *tile, int lane)
{
#ifdef TXD_ABL_NOXCHG
    return;
#endif
    const int g8 = lane & ~7, b = lane & 7;
#pragma unroll
    for (int r = 0; r < 8; r++) tile[b * XPITCH + g8 + r] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = tile[r * XPITCH + lane];
    issue_fence();
}

// forward transform of the 8 complex points held by this lane; tw = [15][64] table in LDS.  The twist
// exp(i pi (64 m + lane)/1024) is applied as constant c16(m) before the first butterfly and as the lane part
// folded into the pass-A twiddles (rows 0..7) after it.
__device__ __forceinline__ void fft_fwd(cplx (&v)[8], const cplx *tw, cplx *tile, int lane)
{
#pragma unroll
    for (int m = 1; m < 8; m++) v[m] = cmul(v[m], c16(m));
    dft8<+1>(v);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = cmul(v[r], tw[r * 64 + lane]);
    xchg_hi(v, tile, lane);
    dft8<+1>(v);
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmul(v[r], tw[(7 + r) * 64 + lane]);
    xchg_lo(v, tile, lane);
    dft8<+1>(v);
}
// unnormalised inverse; on return v[m] = (a[t + 64 m], a[t + 64 m + 512]) as doubles
__device__ __forceinline__ void fft_inv(cplx (&v)[8], const cplx *tw, cplx *tile, int lane)
{
    dft8<-1>(v);
    xchg_lo(v, tile, lane);
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmulc(v[r], tw[(7 + r) * 64 + lane]);
    dft8<-1>(v);
    xchg_hi(v, tile, lane);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = cmulc(v[r], tw[r * 64 + lane]);
    dft8<-1>(v);
#pragma unroll
    for (int m = 1; m < 8; m++) v[m] = cmulc(v[m], c16(m));
}


// diagnostic only (-DTXD_STAMPS): cycle stamps inside a transform; st == nullptr compiles to nothing
struct FftStamp { unsigned long long acc[4]; unsigned long long t; };
__device__ __forceinline__ void fft_stamp(FftStamp *st, int k)
{
    if (st) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long n = __builtin_amdgcn_s_memtime();
        st->acc[k] += n - st->t;
        st->t = n;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// P polynomials transformed together: the twiddles of a pass are loaded once (batched, so the
// LDS reads are all in flight before the first use) and the P tile round trips are issued back to
// back on the same tile -- the LDS unit executes one wave's DS instructions in order, so poly p+1's
// tile writes cannot overtake poly p's tile reads -- which lets butterflies of one polynomial
// cover the LDS latency of the next.  Arithmetic per polynomial is identical to fft_fwd/fft_inv.
template <int P>
__device__ __forceinline__ void fft_fwd_n(cplx (&v)[P][8], const cplx *tw, cplx *tile, int lane, FftStamp *st = nullptr)
{
    if (st) st->t = __builtin_amdgcn_s_memtime();
    // software pipeline over the P polynomials: polynomial p's tile round trip is in flight while
    // polynomial p+1's butterflies issue; the waits land at the next pass's first use of v[p].
    {
        cplx w[8];
#pragma unroll
        for (int r = 0; r < 8; r++) w[r] = tw[r * 64 + lane];
        issue_fence();
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int m = 1; m < 8; m++) v[p][m] = cmul(v[p][m], c16(m));
        fft_stamp(st, 0);
#pragma unroll
        for (int p = 0; p < P; p++) {
            dft8<+1>(v[p]);
#pragma unroll
            for (int r = 0; r < 8; r++) v[p][r] = cmul(v[p][r], w[r]);
            xchg_hi(v[p], tile, lane);
        }
    }
    fft_stamp(st, 1);
    {
        cplx w[7];
#pragma unroll
        for (int r = 1; r < 8; r++) w[r - 1] = tw[(7 + r) * 64 + lane];
        issue_fence();
#pragma unroll
        for (int p = 0; p < P; p++) {
            dft8<+1>(v[p]);
#pragma unroll
            for (int r = 1; r < 8; r++) v[p][r] = cmul(v[p][r], w[r - 1]);
            xchg_lo(v[p], tile, lane);
        }
    }
    fft_stamp(st, 2);
#pragma unroll
    for (int p = 0; p < P; p++) dft8<+1>(v[p]);
    fft_stamp(st, 3);
}
template <int P>
__device__ __forceinline__ void fft_inv_n(cplx (&v)[P][8], const cplx *tw, cplx *tile, int lane)
{
#pragma unroll
    for (int p = 0; p < P; p++) {
        dft8<-1>(v[p]);
        xchg_lo(v[p], tile, lane);
    }
    {
        cplx w[7];
#pragma unroll
        for (int r = 1; r < 8; r++) w[r - 1] = tw[(7 + r) * 64 + lane];
        issue_fence();
#pragma unroll
        for (int p = 0; p < P; p++) {
#pragma unroll
            for (int r = 1; r < 8; r++) v[p][r] = cmulc(v[p][r], w[r - 1]);
            dft8<-1>(v[p]);
            xchg_hi(v[p], tile, lane);
        }
    }
    {
        cplx w[8];
#pragma unroll
        for (int r = 0; r < 8; r++) w[r] = tw[r * 64 + lane];
        issue_fence();
#pragma unroll
        for (int p = 0; p < P; p++) {
#pragma unroll
            for (int r = 0; r < 8; r++) v[p][r] = cmulc(v[p][r], w[r]);
            dft8<-1>(v[p]);
#pragma unroll
            for (int m = 1; m < 8; m++) v[p][m] = cmulc(v[p][m], c16(m));
        }
    }
}

// round-half-even(x) mod 2^64 (oracle f64_to_torus) as the two dwords of the result.  Same values as
//   m = x 2^-64;  y = m - rint(m);  t1 = fma(y, 2^64, 1.5 2^84);  h = t1 - 1.5 2^84;  lo = fma(y, 2^64, -h);
//   t2 = lo + 1.5 2^52;  result = (bits(t1).lo + bits(t2).hi - 0x43380000) : bits(t2).lo
// with y 2^64 formed exactly as x - rint(m) 2^64 (one fma whose destination is the dead x), so that the two
// fmas with a non-inline addend -- each of which costs two v_mov to re-materialise the constant it overwrites --
// become plain adds against constants the compiler can keep in scalar registers.
__device__ __forceinline__ void f64_to_torus2(double x, uint32_t &rlo, uint32_t &rhi)
{
    double r = __builtin_rint(x * 0x1p-64);
    double ys = __builtin_fma(r, -0x1p64, x);               // = (m - rint(m)) 2^64, exact
    double t1 = ys + 0x1.8p84;
    double h = t1 - 0x1.8p84;
    double lo = ys - h;
    double t2 = lo + 0x1.8p52;
    uint64_t b1 = (uint64_t)__double_as_longlong(t1), b2 = (uint64_t)__double_as_longlong(t2);
    rlo = (uint32_t)b2;
    rhi = (uint32_t)b1 + (uint32_t)(b2 >> 32) - 0x43380000u;
}
__device__ __forceinline__ uint64_t f64_to_torus(double x)
{
    uint32_t rlo, rhi;
    f64_to_torus2(x, rlo, rhi);
    return ((uint64_t)rhi << 32) | rlo;
}
// cur + round(x) mod 2^64 with the carry handled on dwords (no 64-bit pair has to be assembled by v_mov)
__device__ __forceinline__ uint64_t torus_add_f64(uint64_t cur, double x)
{
    uint32_t rlo, rhi;
    f64_to_torus2(x, rlo, rhi);
    const uint32_t cl = (uint32_t)cur, ch = (uint32_t)(cur >> 32);
    const uint32_t nl = cl + rlo;
    const uint32_t nh = ch + rhi + (uint32_t)(nl < cl);
    return ((uint64_t)nh << 32) | nl;
}
// torus word as signed 64-bit -> f64 with a single rounding
__device__ __forceinline__ double torus_to_f64(uint64_t x)
{
    return __builtin_fma((double)(int32_t)(x >> 32), 0x1p32, (double)(uint32_t)x);
}

// ---- signed gadget decomposition (tfhe SignedDecomposer), streamed finest level first ----
template <int B, int L>
struct Decomp {
    static constexpr int BL = B * L;
    static constexpr int NONREP = 64 - BL;
    typedef typename std::conditional<(BL <= 31), uint32_t, uint64_t>::type state_t;
    // closest representable, shifted down: state in [0, 2^BL)
    __device__ static __forceinline__ state_t init(uint64_t x)
    {
        uint64_t s = ((x >> (NONREP - 1)) + 1) >> 1;
        return (state_t)(s & ((((uint64_t)1) << BL) - 1));
    }
    // next digit (level L first, then L-1, ...).  `tie` is 0 except when the TOP level (level 1)
    // is extracted on an odd coefficient index, where it is 1 << (B-1): the tie res == B/2, which
    // tfhe-rs always resolves to +B/2 at the top level, then goes to -B/2 (same value mod 2^64).
    // This removes the +B/8 mean of the top digit that otherwise dominates the LUT output noise
    // (oracle/txo_core.c, txo_decompose_idx).
    __device__ static __forceinline__ int32_t next(state_t &st, state_t tie = 0)
    {
        const state_t mask = (((state_t)1) << B) - 1;
        state_t res = st & mask;
        st >>= B;
        state_t carry = ((res - 1) | st | tie) & res;
        carry >>= (B - 1);
        st += carry;
        return (int32_t)res - (int32_t)(carry << B);
    }
    __device__ static __forceinline__ state_t tie_mask(int lane) { return (state_t)(lane & 1) << (B - 1); }
};

// ---- coefficient-domain helpers on the 16-words-per-lane register layout ----
// Write a polynomial held in registers to the wave's scratch in natural order.
__device__ __forceinline__ void poly_to_lds(const uint64_t (&p)[16], uint64_t *s, int lane)
{
    wave_lds_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) s[lane + 64 * m] = p[m];
    wave_lds_fence();
}
// coefficient j of  s(X) * X^rot  (rot in [0, 2N)), s in LDS natural order
__device__ __forceinline__ uint64_t rot_read(const uint64_t *s, int j, int rot)
{
    int src = j - rot;                 // in (-2N, N)
    uint64_t neg = 0;
    if (src < 0) { src += N; neg = ~neg; }
    if (src < 0) { src += N; neg = ~neg; }
    uint64_t x = s[src];
    return (x ^ neg) - neg;
}


// all 16 coefficients of this lane of  s(X) * X^rot - own[], reads issued together
__device__ __forceinline__ void rot_sub16(const uint64_t *s, const uint64_t (&own)[16], int rot, int lane, uint64_t (&out)[16])
{
    uint64_t x[16], neg[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        int src = lane + 64 * m - rot;
        uint64_t ng = 0;
        if (src < 0) { src += N; ng = ~ng; }
        if (src < 0) { src += N; ng = ~ng; }
        x[m] = s[src];
        neg[m] = ng;
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[m] = ((x[m] ^ neg[m]) - neg[m]) - own[m];
}

}  // namespace txd
