// tx_kernels.hpp -- gfx950 kernels of the transciphering hot path.
//
// Work decomposition: one wavefront per unit (ciphertext / key-switch / LUT accumulator), four
// units per 256-thread workgroup, which share the twiddle table in LDS.  Units are independent,
// so a launch of U units is ceil(U/4) workgroups; with U in the thousands (128 bits x blocks) the
// grid is many times the 256 CUs.  All workgroups walk the evaluation key in the same order, so
// the key stream (72 KiB per blind-rotation step) is served from L2 / Infinity Cache, not HBM.
#pragma once
#include "tx_device.hpp"
#include <type_traits>

namespace txd {

// AES_TIGHT decomposition constants (cbs_lib/src/aes_instances.rs:82-95), compile-time here
constexpr int PBS_B = 23, PBS_L = 1;
constexpr int KS_B = 4, KS_L = 3, KS_N = 256;
constexpr int AUTO_B = 13, AUTO_L = 3, AUTO_SPLIT = 41;
constexpr int SS_B = 17, SS_L = 2;
constexpr int CBS_B = 2, CBS_L = 7;
constexpr int LOG_LUT = 3;
constexpr int LOGN = 10;

#ifdef TXD_STAMPS       // diagnostic build: s_memtime per phase of k_blind_rotate, summed over the steps of one wave
__device__ unsigned long long txd_stamps[16];                   // read back by tx_debug_stamps(); no output depends on it
__device__ unsigned long long txd_stamps_sink[16];
__device__ unsigned long long txd_wave_rt[2 * 4096];           // per wave slot: s_memrealtime at the start / the end of its share (k_blind_rotate5)              // where every other wave of a generated loop writes its stamps
#define TXD_STAMP_DECL() unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime(); FftStamp fst = {{0, 0, 0, 0}, 0}
#define TXD_FST (&fst)
#define TXD_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long st_n = __builtin_amdgcn_s_memtime(); \
                          st_acc[k] += st_n - st_t; st_t = st_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef TXD_STAMP_THREAD
#define TXD_STAMP_THREAD 0
#endif
#ifndef TXD_STAMP_BLOCK
#define TXD_STAMP_BLOCK 64
#endif
#define TXD_STAMP_FLUSH(steps) do { if (blockIdx.x == TXD_STAMP_BLOCK && threadIdx.x == TXD_STAMP_THREAD) { for (int q = 0; q < 6; q++) txd_stamps[q] = st_acc[q]; \
                                    txd_stamps[6] = (unsigned long long)(steps); for (int q = 0; q < 4; q++) txd_stamps[8 + q] = fst.acc[q]; } } while (0)
#else
#define TXD_STAMP_DECL() do {} while (0)
#define TXD_FST nullptr
#define TXD_STAMP(k) do {} while (0)
#define TXD_STAMP_FLUSH(steps) do {} while (0)
#endif
#ifdef TXD_PHASE_MARK   // diagnostic build: phase comments in the ISA, counted by tools/isa_phases.py
#define TXD_PHASE(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; TXPHASE " #k ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TXD_PHASE(k) do {} while (0)
#endif

constexpr int WAVES = 4;                                   // units per workgroup
constexpr size_t LDS_BYTES = (size_t)(TW_ROWS * 64 + WAVES * SCRATCH_CPLX) * sizeof(cplx);

struct WaveCtx {
    const cplx *tw;     // LDS twiddles
    cplx *tile;         // this wave's LDS tile
    uint64_t *pl;       // same memory viewed as 1024 torus words
    int lane;
    int unit;           // global unit index of this wave
    int tie_lane;       // lane & tie_on: parity of the coefficient index where the top-level tie goes to -B/2 (balanced rule),
                        // 0 everywhere under tfhe-rs' rule (tx_set_decomposer_tie_rule)
};

// loads the twiddle table into LDS; returns the per-wave context.  Contains the only
// workgroup-wide barrier of every kernel.
__device__ __forceinline__ WaveCtx wave_setup(const cplx *tw_g, unsigned char *lds,
                                              size_t wave_bytes = SCRATCH_CPLX * sizeof(cplx))
{
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < TW_ROWS * 64; i += blockDim.x) tw[i] = tw_g[i];
    __syncthreads();
    WaveCtx w;
    int wave = threadIdx.x >> 6;
    w.tw = tw;
    w.tile = reinterpret_cast<cplx *>(lds + TW_ROWS * 64 * sizeof(cplx) + wave * wave_bytes);
    w.pl = reinterpret_cast<uint64_t *>(w.tile);
    w.lane = threadIdx.x & 63;
    w.unit = blockIdx.x * WAVES + wave;
    w.tie_lane = w.lane;
    return w;
}

__device__ __forceinline__ uint32_t modswitch(uint64_t x)
{   // fast_pbs_modulus_switch(x, N, 0, LOG_LUT)   (pbs.rs:84-89)
    uint64_t o = x >> (64 - LOGN - 2 + LOG_LUT);
    o += (o & 1);
    o >>= 1;
    return (uint32_t)(o << LOG_LUT);
}

__device__ __forceinline__ void fmac8(cplx (&acc)[8], const cplx (&d)[8], const cplx *key, int lane)
{   // pointwise acc += d * key on this lane's 8 frequencies; key is [8][64] complex (F-layout)
#pragma unroll
    for (int r = 0; r < 8; r++) {
        cplx g = key[r * 64 + lane];
        double re = acc[r].x, im = acc[r].y;
        re = __builtin_fma(d[r].x, g.x, re);
        re = __builtin_fma(-d[r].y, g.y, re);
        im = __builtin_fma(d[r].x, g.y, im);
        im = __builtin_fma(d[r].y, g.x, im);
        acc[r] = make_double2(re, im);
    }
}

__device__ __forceinline__ void load_key8(cplx (&g)[8], const cplx *key, int lane)
{
#ifdef TXD_ABL_NOKEY
#pragma unroll
    for (int r = 0; r < 8; r++) g[r] = make_double2(1.0 + r, 0.5 * lane);
    asm volatile("" :: "v"(key));
    return;
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) g[r] = key[r * 64 + lane];
    issue_fence();
}
__device__ __forceinline__ void fmac8r(cplx (&acc)[8], const cplx (&d)[8], const cplx (&g)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        double re = acc[r].x, im = acc[r].y;
        re = __builtin_fma(d[r].x, g[r].x, re);
        re = __builtin_fma(-d[r].y, g[r].y, re);
        im = __builtin_fma(d[r].x, g[r].y, im);
        im = __builtin_fma(d[r].y, g[r].x, im);
        acc[r] = make_double2(re, im);
    }
}

// P consecutive decomposition levels (finest of the group first) of one row: digits -> f64,
// transformed together, then accumulated against the key: F[c] += D[i] * key[lev(i)][c], for fixed c
// in increasing i (the oracle's order: row outer, level finest-first inner).  Key loads are
// software-pipelined MAC_DEPTH polynomials ahead (2: 16 KB in flight per wave; the LUT tree's multiply-accumulate waits for the
// GGSW rows, one wave per SIMD -- 2.74 -> 2.55 ms per launch of 1024 bytes x 3 tables; 3 ahead spills), the first of them
// in flight during the forward transforms (two there spill as well).  lev0 = level index (0 = level 1) of the group's first digit.
// transform P digit polynomials (already f64) and accumulate them against key polynomials
// key[(lev0 - i) * lev_stride + c * col_stride], i = 0..P-1
#ifndef MAC_DEPTH
#define MAC_DEPTH 2
#endif
#ifndef MAC_EARLY
#define MAC_EARLY 1
#endif
template <int P, int NCOL>
__device__ __forceinline__ void mac_group(cplx (&F)[NCOL][8], cplx (&d)[P][8], int lev0, const cplx *key,
                                          size_t lev_stride, size_t col_stride, const WaveCtx &w)
{
    constexpr int T = P * NCOL, DEPTH = MAC_DEPTH;          // key polynomials of the group; how many are requested ahead of the one in use
    cplx g[DEPTH + 1][8];
    // polynomial j = (i, c) = (j / NCOL, j % NCOL): level lev0 - i, column c
#pragma unroll
    for (int j = 0; j < MAC_EARLY && j < DEPTH && j < T; j++)        // in flight during the forward transforms
        load_key8(g[j % (DEPTH + 1)], key + (size_t)(lev0 - j / NCOL) * lev_stride + (size_t)(j % NCOL) * col_stride, w.lane);
    fft_fwd_n<P>(d, w.tw, w.tile, w.lane);
#pragma unroll
    for (int j = MAC_EARLY; j < DEPTH && j < T; j++)
        load_key8(g[j % (DEPTH + 1)], key + (size_t)(lev0 - j / NCOL) * lev_stride + (size_t)(j % NCOL) * col_stride, w.lane);
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int n = j + DEPTH;
        if (n < T) load_key8(g[n % (DEPTH + 1)], key + (size_t)(lev0 - n / NCOL) * lev_stride + (size_t)(n % NCOL) * col_stride, w.lane);
        fmac8r(F[j % NCOL], d[j / NCOL], g[j % (DEPTH + 1)]);
    }
}
template <int B, int L, int P, int NCOL, typename ST>
__device__ __forceinline__ void extprod_group(cplx (&F)[NCOL][8], ST (&slo)[8], ST (&shi)[8], int lev0,
                                              const cplx *key, size_t lev_stride, size_t col_stride, const WaveCtx &w)
{
    typedef Decomp<B, L> D;
    cplx d[P][8];
#pragma unroll
    for (int i = 0; i < P; i++) {
        const ST tie = (lev0 - i == 0) ? D::tie_mask(w.tie_lane) : (ST)0;  // top level: balanced tie (or none: tfhe-rs' rule)
#pragma unroll
        for (int m = 0; m < 8; m++) d[i][m] = make_double2((double)D::next(slo[m], tie), (double)D::next(shi[m], tie));
    }
    mac_group<P, NCOL>(F, d, lev0, key, lev_stride, col_stride, w);
}

// One row of an external product: decompose the 16 coefficients of this lane (lo[m] = coeff
// t+64m, hi[m] = coeff t+64m+512), level L (finest) first, in groups of up to 3 levels.
// key points at GGSW element [lev 0][row r][col 0]; strides in complex units.
template <int B, int L, int NCOL>
__device__ __forceinline__ void extprod_row(cplx (&F)[NCOL][8], const uint64_t (&lo)[8], const uint64_t (&hi)[8],
                                            const cplx *key, size_t lev_stride, size_t col_stride, const WaveCtx &w)
{
    typedef Decomp<B, L> D;
    typename D::state_t slo[8], shi[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { slo[m] = D::init(lo[m]); shi[m] = D::init(hi[m]); }
    constexpr int G3 = L / 3, REM = L % 3;
    int lev = L - 1;
#pragma unroll 1
    for (int gidx = 0; gidx < G3; gidx++) {
        extprod_group<B, L, 3, NCOL>(F, slo, shi, lev, key, lev_stride, col_stride, w);
        lev -= 3;
    }
    if (REM == 2) extprod_group<B, L, 2, NCOL>(F, slo, shi, lev, key, lev_stride, col_stride, w);
    if (REM == 1) extprod_group<B, L, 1, NCOL>(F, slo, shi, lev, key, lev_stride, col_stride, w);
}

// inverse-transform a Fourier accumulator and add the rounded result into 16 torus words
__device__ __forceinline__ void inv_add(cplx (&F)[8], uint64_t (&p)[16], const WaveCtx &w)
{
    fft_inv(F, w.tw, w.tile, w.lane);
#pragma unroll
    for (int m = 0; m < 8; m++) { p[m] = torus_add_f64(p[m], F[m].x); p[m + 8] = torus_add_f64(p[m + 8], F[m].y); }
}
// the three accumulators of a GLWE together
__device__ __forceinline__ void inv_add3(cplx (&F)[KP][8], uint64_t (&p)[KP][16], const WaveCtx &w)
{
    fft_inv_n<KP>(F, w.tw, w.tile, w.lane);
#pragma unroll
    for (int c = 0; c < KP; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) { p[c][m] = torus_add_f64(p[c][m], F[c][m].x); p[c][m + 8] = torus_add_f64(p[c][m + 8], F[c][m].y); }
}

template <int NCOL>
__device__ __forceinline__ void zero_f(cplx (&F)[NCOL][8])
{
#pragma unroll
    for (int c = 0; c < NCOL; c++)
#pragma unroll
        for (int r = 0; r < 8; r++) F[c][r] = make_double2(0.0, 0.0);
}

// ------------------------------------------------------------------------------------------
// key-side transform: torus polynomial -> Fourier (F-layout), scaled by 2/N.
// mode 0: as is; mode 1: low `split` bits; mode 2: x >> split   (fourier_glwe_keyswitch.rs:188-199)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_torus(const uint64_t *in, cplx *out, int npoly, int mode, int split,
                                                   const cplx *tw_g)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds);
    if (w.unit >= npoly) return;
    const uint64_t *p = in + (size_t)w.unit * N;
    cplx v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        uint64_t a = p[w.lane + 64 * m], b = p[w.lane + 64 * m + 512];
        if (mode == 1) { a = (a << (64 - split)) >> (64 - split); b = (b << (64 - split)) >> (64 - split); }
        if (mode == 2) { a >>= split; b >>= split; }
        v[m] = make_double2(torus_to_f64(a), torus_to_f64(b));
    }
    fft_fwd(v, w.tw, w.tile, w.lane);
    cplx *o = out + (size_t)w.unit * H;
#pragma unroll
    for (int r = 0; r < 8; r++) o[r * 64 + w.lane] = make_double2(v[r].x * 0x1p-9, v[r].y * 0x1p-9);
}

// ------------------------------------------------------------------------------------------
// N = 256 transform (LWE key switch): 128 complex points, 2 per lane, radix-2 across lanes.
// tws: [128 twist][64 w] complex in LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ cplx shfl_xor_c(cplx v, int mask)
{
    return make_double2(__shfl_xor(v.x, mask), __shfl_xor(v.y, mask));
}
__device__ __forceinline__ void fft128_fwd(cplx &x0, cplx &x1, const cplx *twist, const cplx *wt, int lane)
{
    x0 = cmul(x0, twist[lane]);
    x1 = cmul(x1, twist[lane + 64]);
    {
        cplx u = x0, v = x1;
        x0 = cadd(u, v);
        x1 = cmul(csub(u, v), wt[lane]);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const bool upper = (lane & s) != 0;
        const cplx wv = wt[(lane & (s - 1)) * (64 / s)];
        cplx o0 = shfl_xor_c(x0, s), o1 = shfl_xor_c(x1, s);
        x0 = upper ? cmul(csub(o0, x0), wv) : cadd(x0, o0);
        x1 = upper ? cmul(csub(o1, x1), wv) : cadd(x1, o1);
    }
}
__device__ __forceinline__ void fft128_inv(cplx &x0, cplx &x1, const cplx *twist, const cplx *wt, int lane)
{
#pragma unroll
    for (int s = 1; s <= 32; s <<= 1) {
        const bool upper = (lane & s) != 0;
        const cplx wv = wt[(lane & (s - 1)) * (64 / s)];
        cplx s0 = upper ? cmulc(x0, wv) : x0, s1 = upper ? cmulc(x1, wv) : x1;
        cplx r0 = shfl_xor_c(s0, s), r1 = shfl_xor_c(s1, s);
        x0 = upper ? csub(r0, s0) : cadd(s0, r0);
        x1 = upper ? csub(r1, s1) : cadd(s1, r1);
    }
    {
        cplx u = x0, v = cmulc(x1, wt[lane]);
        x0 = cadd(u, v);
        x1 = csub(u, v);
    }
    x0 = cmulc(x0, twist[lane]);
    x1 = cmulc(x1, twist[lane + 64]);
}

__global__ __launch_bounds__(256) void k_fwd_torus256(const uint64_t *in, cplx *out, int npoly, const cplx *tw256_g)
{
    __shared__ cplx tws[192];
    for (int i = threadIdx.x; i < 192; i += blockDim.x) tws[i] = tw256_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, unit = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (unit >= npoly) return;
    const uint64_t *p = in + (size_t)unit * KS_N;
    cplx x0 = make_double2(torus_to_f64(p[lane]), torus_to_f64(p[lane + 128]));
    cplx x1 = make_double2(torus_to_f64(p[lane + 64]), torus_to_f64(p[lane + 192]));
    fft128_fwd(x0, x1, tws, tws + 128, lane);
    cplx *o = out + (size_t)unit * 128;
    o[lane] = make_double2(x0.x * 0x1p-7, x0.y * 0x1p-7);
    o[lane + 64] = make_double2(x1.x * 0x1p-7, x1.y * 0x1p-7);
}

// ------------------------------------------------------------------------------------------
// a4: LWE key switch 2048 -> n as a GLWE key switch at N_c = 256
// (fourier_glwe_keyswitch.rs:213-379, glwe_conv.rs:27-43).  One wave per LWE.
// ksk_f: [kin = 8][lev 3][Kout][128] complex.
// ------------------------------------------------------------------------------------------
constexpr int KS_MAXKOUT = 5;
__global__ __launch_bounds__(256) void k_lwe_keyswitch(const uint64_t *in, uint64_t *out, int count, int n,
                                                       const cplx *ksk_f, const cplx *tw256_g, int tie_on)
{
    __shared__ cplx tws[192];
    for (int i = threadIdx.x; i < 192; i += blockDim.x) tws[i] = tw256_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, unit = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (unit >= count) return;
    const cplx *twist = tws, *wt = tws + 128;
    const int kout = n / KS_N, Kout = kout + 1, kin = 2 * N / KS_N;
    const uint64_t *ct = in + (size_t)unit * BIG;
    cplx F[KS_MAXKOUT][2];
#pragma unroll
    for (int c = 0; c < KS_MAXKOUT; c++) { F[c][0] = make_double2(0, 0); F[c][1] = make_double2(0, 0); }
    typedef Decomp<KS_B, KS_L> D;
    for (int q = 0; q < kin; q++) {
        const uint64_t *a = ct + q * KS_N;
        // g[0] = a[0], g[j] = -a[256 - j]; this lane holds j = lane + 64 r, r = 0..3
        uint64_t g[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int j = lane + 64 * r;
            g[r] = (j == 0) ? a[0] : (uint64_t)0 - a[KS_N - j];
        }
        D::state_t st[4];
#pragma unroll
        for (int r = 0; r < 4; r++) st[r] = D::init(g[r]);
#pragma unroll 1
        for (int i = 0; i < KS_L; i++) {
            const int lev = KS_L - 1 - i;
            const D::state_t tie = (lev == 0) ? D::tie_mask(lane & tie_on) : 0;      // top level: balanced tie (tie_on = 0: tfhe-rs' rule)
            // complex point lane      = (g[lane],      g[lane + 128])  -> regs r = 0, 2
            // complex point lane + 64 = (g[lane + 64], g[lane + 192])  -> regs r = 1, 3
            double d0 = (double)D::next(st[0], tie), d1 = (double)D::next(st[1], tie);
            double d2 = (double)D::next(st[2], tie), d3 = (double)D::next(st[3], tie);
            cplx x0 = make_double2(d0, d2), x1 = make_double2(d1, d3);
            fft128_fwd(x0, x1, twist, wt, lane);
            const cplx *key = ksk_f + ((size_t)(q * KS_L + lev) * Kout) * 128;
#pragma unroll
            for (int c = 0; c < KS_MAXKOUT; c++)
                if (c < Kout) {
                    cplx g0 = key[c * 128 + lane], g1 = key[c * 128 + lane + 64];
                    double re = F[c][0].x, im = F[c][0].y;
                    re = __builtin_fma(x0.x, g0.x, re); re = __builtin_fma(-x0.y, g0.y, re);
                    im = __builtin_fma(x0.x, g0.y, im); im = __builtin_fma(x0.y, g0.x, im);
                    F[c][0] = make_double2(re, im);
                    re = F[c][1].x; im = F[c][1].y;
                    re = __builtin_fma(x1.x, g1.x, re); re = __builtin_fma(-x1.y, g1.y, re);
                    im = __builtin_fma(x1.x, g1.y, im); im = __builtin_fma(x1.y, g1.x, im);
                    F[c][1] = make_double2(re, im);
                }
        }
    }
    uint64_t *o = out + (size_t)unit * (n + 1);
#pragma unroll
    for (int c = 0; c < KS_MAXKOUT; c++)
        if (c < Kout) {
            cplx x0 = F[c][0], x1 = F[c][1];
            fft128_inv(x0, x1, twist, wt, lane);
            uint64_t r[4] = { f64_to_torus(x0.x), f64_to_torus(x1.x), f64_to_torus(x0.y), f64_to_torus(x1.y) };
            if (c < kout) {
                // sample extract at 0: out[m] = res[0] (m = 0), -res[256 - m] otherwise
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    int j = lane + 64 * rr;
                    if (j == 0) o[c * KS_N] = r[rr]; else o[c * KS_N + KS_N - j] = (uint64_t)0 - r[rr];
                }
            } else if (lane == 0) {
                o[n] = r[0] + ct[2 * N];
            }
        }
}

// ------------------------------------------------------------------------------------------
// a5: multi-value blind rotation of the CBS test vector
// (ggsw_conv.rs:250-300, pbs.rs:70-146).  One wave per input LWE; the accumulator (3 x 1024
// torus words) stays in the wave's LDS slice for all n CMux steps (the rotation X^a is a lane
// permutation, so it has to pass through LDS anyway), the three digit transforms and the three
// Fourier accumulators live in registers; bsk_f: [n][row 3][col 3][512] complex.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t test_vector(int i)
{   // coefficient i in [0, N) of the accumulator built at ggsw_conv.rs:252-261
    const int sh = 63 - CBS_B * ((i & ((1 << LOG_LUT) - 1)) + 1);
    const uint64_t v = (uint64_t)1 << sh;
    return (i < N / 2) ? (uint64_t)0 - v : v;
}

// LDS of the blind-rotation kernel: twiddles, then per wave: transpose tile + the accumulator
// (3 x 1024 torus words, natural order).  One workgroup per CU (154 KiB).
constexpr size_t BR_WAVE_BYTES = SCRATCH_CPLX * sizeof(cplx) + (size_t)GL * sizeof(uint64_t);
constexpr size_t LDS_BYTES_BR = TW_ROWS * 64 * sizeof(cplx) + WAVES * BR_WAVE_BYTES;

// one-level decomposition with B = 23 on the high dword only (closest multiple of 2^41, balanced digit)
// `thr` = 0x400000 on even, 0x3FFFFF on odd coefficient indices (balanced top-level tie)
__device__ __forceinline__ double pbs_digit(uint64_t x, uint32_t thr)
{
    uint32_t s = ((((uint32_t)(x >> 32)) >> 8) + 1) >> 1;       // ((x >> 40) + 1) >> 1
    s &= 0x7FFFFFu;
    int32_t d = (int32_t)s - (int32_t)((s > thr) ? 0x800000u : 0u);
    return (double)d;
}

#ifdef TXD_EXPERIMENTS   // superseded / reference form: only in libtfhe_mi355x_exp.so (tests, A/B timing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_blind_rotate(const uint64_t *lwe, uint64_t *acc_out, int count, int n,
                                                      const cplx *bsk_f, const cplx *tw_g)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds, BR_WAVE_BYTES);
    if (w.unit >= count) return;
#ifdef TXD_EXP_ACTIVE_WAVES
    if ((threadIdx.x >> 6) >= TXD_EXP_ACTIVE_WAVES) return;     // timing experiment only (profiles/README.md)
#endif
    const uint64_t *ct = lwe + (size_t)w.unit * (n + 1);
    const int lane = w.lane;
    uint64_t *accl = reinterpret_cast<uint64_t *>(w.tile + SCRATCH_CPLX);   // [3][1024] in LDS
    {   // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101
        const int rot = 2 * N - (int)modswitch(ct[n]);      // X^(-b) = X^(2N - b); b in [0, 2N]
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int src = lane + 64 * m - rot;                  // in [-2N, N)
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = test_vector(src);
            accl[lane + 64 * m] = 0;
            accl[N + lane + 64 * m] = 0;
            accl[2 * N + lane + 64 * m] = (x ^ neg) - neg;
        }
    }
    const uint32_t thr = 0x400000u - (uint32_t)(lane & 1);
    cplx g[8];
    load_key8(g, bsk_f, lane);                              // key polynomial (step 0, row 0, col 0)
    TXD_STAMP_DECL();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        // a mask element equal to 0 (skipped at pbs.rs:111) rotates by 0: ct1 = 0 and the step adds 0
        const uint32_t rot = modswitch(ct[i]);
        const uint32_t base = ((uint32_t)lane - rot) & (2 * N - 1);
        // ct1 = acc * X^rot - acc (pbs.rs:125-134), one-level decomposition, digits as f64
        cplx d[KP][8];
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < KP; q++) {
            const uint64_t *a = accl + q * N;
            uint64_t own[16], x[16];
#pragma unroll
            for (int m = 0; m < 16; m++) own[m] = a[lane + 64 * m];
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = a[(base + 64 * m) & (N - 1)];
            issue_fence();
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const uint64_t n0 = (uint64_t)0 - (uint64_t)(((base + 64 * m) >> LOGN) & 1);
                const uint64_t n1 = (uint64_t)0 - (uint64_t)(((base + 64 * (m + 8)) >> LOGN) & 1);
                d[q][m] = make_double2(pbs_digit(((x[m] ^ n0) - n0) - own[m], thr),
                                       pbs_digit(((x[m + 8] ^ n1) - n1) - own[m + 8], thr));
            }
        }
        TXD_PHASE(1);
        TXD_STAMP(0);
        fft_fwd_n<KP>(d, w.tw, w.tile, lane, TXD_FST);
        TXD_PHASE(2);
        TXD_STAMP(1);
        const cplx *row = bsk_f + (size_t)i * KP * KP * H;  // [row q][col c][512]
        const int inext = (i + 1 < n) ? i + 1 : i;
        cplx F[KP][8];
        zero_f(F);
#pragma unroll
        for (int c = 0; c < KP; c++)
#pragma unroll
            for (int q = 0; q < KP; q++) {
                cplx gn[8];
                const cplx *nk = (q + 1 < KP) ? row + ((size_t)(q + 1) * KP + c) * H
                               : (c + 1 < KP) ? row + (size_t)(c + 1) * H
                                              : bsk_f + (size_t)inext * KP * KP * H;
                load_key8(gn, nk, lane);
                fmac8r(F[c], d[q], g);
#pragma unroll
                for (int r = 0; r < 8; r++) g[r] = gn[r];
            }
        TXD_PHASE(3);
        TXD_STAMP(2);
        fft_inv_n<KP>(F, w.tw, w.tile, lane);
        TXD_PHASE(4);
        TXD_STAMP(3);
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < KP; c++) {
            uint64_t *a = accl + c * N;
            uint64_t cur[16];
#pragma unroll
            for (int m = 0; m < 16; m++) cur[m] = a[lane + 64 * m];
            issue_fence();
#pragma unroll
            for (int m = 0; m < 8; m++) {
                a[lane + 64 * m] = torus_add_f64(cur[m], F[c][m].x);
                a[lane + 64 * (m + 8)] = torus_add_f64(cur[m + 8], F[c][m].y);
            }
        }
        TXD_STAMP(4);
    }
    TXD_STAMP_FLUSH(n);
    wave_lds_fence();
    uint64_t *o = acc_out + (size_t)w.unit * GL;
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) o[q * N + lane + 64 * m] = accl[q * N + lane + 64 * m];
}
#endif  // TXD_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// a6 + a7: per (ciphertext, level): extract level from the rotated accumulator, divide by N
// (mod switch), LWE -> GLWE const, then the homomorphic trace: 10 x (automorphism + GLWE key
// switch with the 41/23-bit split key)   (ggsw_conv.rs:304-316, automorphism.rs:134-233,
// fourier_glwe_keyswitch.rs:213-342).  auto_f: [split 2][idx 10][q 2][lev 3][col 3][512].
// ------------------------------------------------------------------------------------------
// LDS of the trace kernels: twiddles, then per wave: transpose tile + the GLWE being traced
// (3 x 1024 torus words, natural order) -- same footprint as the blind rotation, one workgroup per CU.
constexpr size_t TRACE_WAVE_BYTES = BR_WAVE_BYTES;
constexpr size_t LDS_BYTES_TRACE = LDS_BYTES_BR;
constexpr int TRACE_KEYS_PASS = 2 * AUTO_L * KP;            // key polynomials per pass over one key split (18)

// The automorphism keys are the same for every unit and the four waves of a workgroup walk them in the same
// order; streamed per wave, a pass is L2->CU fill bound (4 x 144 KB per CU against ~30 B/clk/CU).  So the
// workgroup fetches every key polynomial ONCE: during a pass the four transpose tiles are idle and serve as a
// 4-slot ring; each wave copies a quarter of a key polynomial straight into a slot with two LDS-DMA loads
// (global_load_lds_dwordx4: no staging registers), three polynomials ahead, and all four waves read the slot
// (8 x ds_read_b128).  One s_barrier per key polynomial keeps the waves in step; it waits for this wave's LDS
// traffic and for all but the youngest DMA loads.  A barrier before the first DMA (every wave has finished its
// forward transforms) and one after the last read (before the inverse transforms) fence the tiles' two uses.
//
// INVARIANT (hand-counted waits): ring_step waits with s_waitcnt vmcnt(2 * ahead), i.e. it assumes that the two
// LDS-DMA loads per key polynomial are the ONLY vector-memory operations this wave has in flight between the first
// and the last ring_barrier of a pass.  No global load or store (and no scratch spill) may be added inside
// ring_pass / ring_step or between the ring_barrier calls of trace_lds and k_scheme_switch: it would shift the
// count and turn a slot read into a read-before-arrival race that only the bit-parity tests detect.  Callers
// drain the counter first (s_waitcnt vmcnt(0) before ring_setup, and after the stores of a scheme-switch pass).
// M0 carries the LDS destination, which on gfx950 may lie above 64 KiB (160 KiB of LDS).  The -m gpu stress test
// test_trace_and_scheme_switch_ring_stress (tests/test_gpu_parity.py) drives the ring with a partially idle last workgroup after any change.
struct KeyRing {
    unsigned char *slot0;       // wave 0's tile; slot s = slot0 + s * slot_stride
    uint32_t slot0_addr;        // the same as an LDS byte address (for M0)
    uint32_t slot_stride;       // bytes between the tiles of consecutive waves
    int wave, lane;
};
__device__ __forceinline__ cplx *ring_slot(const KeyRing &k, int s)
{
    return reinterpret_cast<cplx *>(k.slot0 + (size_t)s * k.slot_stride);
}
// key polynomial j of a trace pass (idx, split): (q, level finest first, column)
struct TraceKeyFn {
    const cplx *auto_f; size_t split_stride; int idx, split;
    __device__ __forceinline__ const cplx *operator()(int j) const
    {
        const int q = j / (AUTO_L * KP), i = (j / KP) % AUTO_L, c = j % KP;
        return auto_f + (size_t)split * split_stride + ((((size_t)idx * 2 + q) * AUTO_L + (AUTO_L - 1 - i)) * KP + c) * H;
    }
};
// one LDS-DMA load: 64 lanes x 16 B from per-lane global addresses to LDS [m0, m0 + 1 KiB).  In asm so that hipcc
// does not count it: with the builtin it drains every DMA in flight (vmcnt(0)) before each read of a ring slot.
__device__ __forceinline__ void glds16(const cplx *src, uint32_t lds_addr)
{
#ifdef TXD_ABL_NODMA
    return;
#endif
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void ring_dma(const KeyRing &k, const cplx *poly, int slot)
{
    const cplx *src = poly + k.wave * 128 + k.lane;
    const uint32_t dst = k.slot0_addr + (uint32_t)slot * k.slot_stride + (uint32_t)k.wave * 2048u;   // wave-uniform
    glds16(src, dst);
    glds16(src + 64, dst + 1024u);
}
// this wave's LDS traffic done and all but its VM youngest DMA loads landed, then the workgroup barrier
template <int VM>
__device__ __forceinline__ void ring_barrier()
{
#ifdef TXD_ABL_NOBAR
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(VM) : "memory");
#endif
}

template <int J, typename KeyFn>
__device__ __forceinline__ void ring_step(cplx (&F)[KP][8], cplx (&Dg)[TRACE_KEYS_PASS / KP][8], const KeyRing &k, const KeyFn &key)
{
    constexpr int ahead = (TRACE_KEYS_PASS - 1 - J < 2) ? TRACE_KEYS_PASS - 1 - J : 2;     // polynomials issued after J
    ring_barrier<2 * ahead>();                              // polynomial J landed everywhere; slot (J-1)%4 free
    cplx g[8];
    const cplx *sl = ring_slot(k, J % 4);
#pragma unroll
    for (int r = 0; r < 8; r++) g[r] = sl[r * 64 + k.lane];
    if (J + 3 < TRACE_KEYS_PASS) ring_dma(k, key(J + 3), (J + 3) % 4);
    fmac8r(F[J % KP], Dg[J / KP], g);
}
// one pass over 18 shared key polynomials: F[c] += D[j / 3] * key(j), j = 0..17 (c = j % 3), in the oracle's order
template <typename KeyFn>
__device__ __forceinline__ void ring_pass(cplx (&F)[KP][8], cplx (&Dg)[TRACE_KEYS_PASS / KP][8], const KeyRing &k, const KeyFn &key)
{
    static_assert(TRACE_KEYS_PASS == 18, "unrolled for 18 key polynomials per pass");
    ring_barrier<0>();                                      // every wave is done with its tile
    ring_dma(k, key(0), 0);
    ring_dma(k, key(1), 1);
    ring_dma(k, key(2), 2);
    ring_step<0>(F, Dg, k, key);   ring_step<1>(F, Dg, k, key);   ring_step<2>(F, Dg, k, key);
    ring_step<3>(F, Dg, k, key);   ring_step<4>(F, Dg, k, key);   ring_step<5>(F, Dg, k, key);
    ring_step<6>(F, Dg, k, key);   ring_step<7>(F, Dg, k, key);   ring_step<8>(F, Dg, k, key);
    ring_step<9>(F, Dg, k, key);   ring_step<10>(F, Dg, k, key);  ring_step<11>(F, Dg, k, key);
    ring_step<12>(F, Dg, k, key);  ring_step<13>(F, Dg, k, key);  ring_step<14>(F, Dg, k, key);
    ring_step<15>(F, Dg, k, key);  ring_step<16>(F, Dg, k, key);  ring_step<17>(F, Dg, k, key);
    ring_barrier<0>();                                      // every wave has read the last slot: tiles free again
}

// One trace on the GLWE held in LDS (gl = [3][1024], natural order): for idx = 0..9, t = N/2^idx + 1:
// g += KS_t(g(X^t)).  Per step: the permuted mask p(X^t) is GATHERED from LDS (coefficient i of p(X^t) is
// +-p[i * t^-1 mod 2N]; an odd stride keeps ds_read_b64 conflict-free), decomposed, and its six digit
// polynomials are transformed ONCE; they stay live (the register allocator parks them in the accumulation
// registers, idle at one wave per SIMD) across the two passes over the split key (low 41 bits, high 23 bits);
// each pass = 18 key polynomials from the shared ring, three inverse transforms and a read-modify-write of g.
// Ring invariant (see KeyRing): nothing in this function touches global memory except ring_dma.
__device__ __forceinline__ void trace_lds(uint64_t *gl, const KeyRing &ring, const cplx *auto_f, size_t split_stride, const WaveCtx &w)
{
    typedef Decomp<AUTO_B, AUTO_L> D;
    constexpr int NP = 2 * AUTO_L;                 // digit polynomials per step
    const int lane = w.lane;
    TXD_STAMP_DECL();
#pragma unroll 1
    for (int idx = 0; idx < LOGN; idx++) {
        const uint32_t t = (N >> idx) + 1;
        uint32_t tinv = t;                         // t * t = 1 mod 8; two Newton steps => mod 2^12 >= 2N
        tinv *= 2u - t * tinv;
        tinv *= 2u - t * tinv;
        cplx Dg[NP][8];
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint64_t x[16];
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = gl[q * N + ((((uint32_t)(lane + 64 * m)) * tinv) & (N - 1))];
            issue_fence();
            D::state_t st[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const uint64_t neg = (uint64_t)0 - (uint64_t)(((((uint32_t)(lane + 64 * m)) * tinv) >> LOGN) & 1);
                st[m] = D::init((x[m] ^ neg) - neg);
            }
            cplx d[AUTO_L][8];
#pragma unroll
            for (int i = 0; i < AUTO_L; i++) {
                const D::state_t tie = (i == AUTO_L - 1) ? D::tie_mask(lane) : (D::state_t)0;
#pragma unroll
                for (int m = 0; m < 8; m++)
                    d[i][m] = make_double2((double)D::next(st[m], tie), (double)D::next(st[m + 8], tie));
            }
            fft_fwd_n<AUTO_L>(d, w.tw, w.tile, lane);
#pragma unroll
            for (int i = 0; i < AUTO_L; i++)
#pragma unroll
                for (int r = 0; r < 8; r++) Dg[q * AUTO_L + i][r] = d[i][r];
        }
        TXD_STAMP(0);
        {   // body: g[2] += g[2](X^t)   (the key switch output starts as (0, 0, permuted body))
            uint64_t *b = gl + 2 * N;
            uint64_t own[16], pb[16];
#pragma unroll
            for (int m = 0; m < 16; m++) own[m] = b[lane + 64 * m];
#pragma unroll
            for (int m = 0; m < 16; m++) pb[m] = b[(((uint32_t)(lane + 64 * m)) * tinv) & (N - 1)];
            issue_fence();
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const uint64_t neg = (uint64_t)0 - (uint64_t)(((((uint32_t)(lane + 64 * m)) * tinv) >> LOGN) & 1);
                b[lane + 64 * m] = own[m] + ((pb[m] ^ neg) - neg);
            }
        }
        TXD_STAMP(1);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            cplx F[KP][8];
            zero_f(F);
            ring_pass(F, Dg, ring, TraceKeyFn{auto_f, split_stride, idx, s});
            TXD_STAMP(2);
            fft_inv_n<KP>(F, w.tw, w.tile, lane);
            TXD_STAMP(3);
            const int sh = s ? AUTO_SPLIT : 0;
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < KP; c++) {
                uint64_t *a = gl + c * N;
                uint64_t cur[16];
#pragma unroll
                for (int m = 0; m < 16; m++) cur[m] = a[lane + 64 * m];
                issue_fence();
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    a[lane + 64 * m] = cur[m] + (f64_to_torus(F[c][m].x) << sh);
                    a[lane + 64 * (m + 8)] = cur[m + 8] + (f64_to_torus(F[c][m].y) << sh);
                }
            }
            TXD_STAMP(4);
        }
    }
#ifdef TXD_STAMPS_TRACE
    TXD_STAMP_FLUSH(LOGN);
#endif
    wave_lds_fence();
}

__device__ __forceinline__ KeyRing ring_setup(unsigned char *lds, size_t wave_bytes)
{
    KeyRing k;
    k.slot0 = lds + TW_ROWS * 64 * sizeof(cplx);
    k.slot0_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)k.slot0);
    k.slot_stride = (uint32_t)wave_bytes;
    k.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    k.lane = (int)(threadIdx.x & 63);
    return k;
}

#ifdef TXD_EXPERIMENTS   // superseded / reference form: only in libtfhe_mi355x_exp.so (tests, A/B timing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_extract_trace(const uint64_t *acc, uint64_t *glev, int count,
                                                       const cplx *auto_f, size_t split_stride, const cplx *tw_g)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds, TRACE_WAVE_BYTES);
    uint64_t *gl = reinterpret_cast<uint64_t *>(w.tile + SCRATCH_CPLX);
    const bool active = w.unit < count * CBS_L;     // idle waves of the last workgroup redo the last unit (in their
    if (!active) w.unit = count * CBS_L - 1;        // own LDS): every wave takes part in the key ring's barriers
    const int ctx = w.unit / CBS_L, lvl = w.unit % CBS_L;         // lvl = level - 1 = rotation k
    const uint64_t *a = acc + (size_t)ctx * GL;
    const int lane = w.lane;
    // buf = acc * X^(-lvl); body[0] += 2^(63 - 2 level); extract@0; >> 10; LWE -> GLWE const:
    //   mask: g[j] = buf[0] >> 10 (j = 0), -((-buf[j]) >> 10) otherwise; body: (buf[0] + c) >> 10, 0...
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int j = lane + 64 * m, s = j + lvl;
            uint64_t x = (s < N) ? a[q * N + s] : (uint64_t)0 - a[q * N + s - N];
            gl[q * N + j] = (j == 0) ? (x >> LOGN) : (uint64_t)0 - (((uint64_t)0 - x) >> LOGN);
        }
    {
        const uint64_t b0 = (a[2 * N + lvl] + ((uint64_t)1 << (63 - CBS_B * (lvl + 1)))) >> LOGN;
#pragma unroll
        for (int m = 0; m < 16; m++) gl[2 * N + lane + 64 * m] = (lane + 64 * m == 0) ? b0 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nothing older than the ring's DMA on the VM counter
    const KeyRing ring = ring_setup(lds, TRACE_WAVE_BYTES);
    trace_lds(gl, ring, auto_f, split_stride, w);
    if (!active) return;
    uint64_t *o = glev + (size_t)w.unit * GL;
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) o[q * N + lane + 64 * m] = gl[q * N + lane + 64 * m];
}
#endif  // TXD_EXPERIMENTS

// trace_assign alone on `count` GLWEs in place (unit parity for a7)
#ifdef TXD_EXPERIMENTS   // superseded / reference form: only in libtfhe_mi355x_exp.so (tests, A/B timing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_trace(uint64_t *glwe, int count, const cplx *auto_f, size_t split_stride,
                                               const cplx *tw_g)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds, TRACE_WAVE_BYTES);
    uint64_t *gl = reinterpret_cast<uint64_t *>(w.tile + SCRATCH_CPLX);
    const bool active = w.unit < count;
    if (!active) w.unit = count - 1;
    uint64_t *p = glwe + (size_t)w.unit * GL;
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) gl[q * N + w.lane + 64 * m] = p[q * N + w.lane + 64 * m];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const KeyRing ring = ring_setup(lds, TRACE_WAVE_BYTES);
    trace_lds(gl, ring, auto_f, split_stride, w);
    if (!active) return;
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) p[q * N + w.lane + 64 * m] = gl[q * N + w.lane + 64 * m];
}
#endif  // TXD_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// a8 + a9: scheme switching GLev -> GGSW, emitted directly in the Fourier domain
// (ggsw_conv.rs:163-193, SRV_AES:430-436).  Unit = (ciphertext, level).
// ss_f: [i 2][lev 2][row 3][col 3][512];  out: [unit][row 3][col 3][512] complex.
// ------------------------------------------------------------------------------------------
// three torus polynomials -> Fourier, scaled by 2/N, stored at o + c * H
__device__ __forceinline__ void store_fourier_scaled3(const uint64_t (&p)[KP][16], cplx *o, const WaveCtx &w)
{
    cplx v[KP][8];
#pragma unroll
    for (int c = 0; c < KP; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) v[c][m] = make_double2(torus_to_f64(p[c][m]), torus_to_f64(p[c][m + 8]));
    fft_fwd_n<KP>(v, w.tw, w.tile, w.lane);
#pragma unroll
    for (int c = 0; c < KP; c++)
#pragma unroll
        for (int r = 0; r < 8; r++) o[(size_t)c * H + r * 64 + w.lane] = make_double2(v[c][r].x * 0x1p-9, v[c][r].y * 0x1p-9);
}

// key polynomial j of the scheme-switch pass for mask row `sel`: (GLev polynomial r, level finest first, column)
struct SsKeyFn {
    const cplx *ss_f; int sel;
    __device__ __forceinline__ const cplx *operator()(int j) const
    {
        const int r = j / (SS_L * KP), i = (j / KP) % SS_L, c = j % KP;
        return ss_f + ((((size_t)sel * SS_L + (SS_L - 1 - i)) * KP + r) * KP + c) * H;
    }
};

// The six digit polynomials of the GLev element (3 polynomials x 2 levels) are transformed ONCE and serve both
// mask rows; the scheme-switching key is the same for every unit, so each pass takes its 18 key polynomials from
// the workgroup's shared ring (see the trace kernel).  Same products in the same order as extprod_row.
#ifdef TXD_EXPERIMENTS   // superseded / reference form: only in libtfhe_mi355x_exp.so (tests, A/B timing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_scheme_switch(const uint64_t *glev, cplx *ggsw_f, int units,
                                                       const cplx *ss_f, const cplx *tw_g)
{
    static_assert(SS_L * KP * KP == TRACE_KEYS_PASS && SS_L == 2, "ring pass shape");
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds);
    const bool active = w.unit < units;            // idle waves of the last workgroup shadow the last unit:
    if (!active) w.unit = units - 1;               // every wave takes part in the ring's barriers
    const uint64_t *gp = glev + (size_t)w.unit * GL;
    cplx *o = ggsw_f + (size_t)w.unit * KP * KP * H;
    typedef Decomp<SS_B, SS_L> D;
    cplx Dg[SS_L * KP][8];                         // [r * 2 + i], i = 0 the finest level
#pragma unroll
    for (int half = 0; half < 2; half++) {         // two groups of three digit polynomials, transformed together
        cplx d[3][8];
#pragma unroll
        for (int r = 0; r < KP; r++) {
            if ((r * SS_L) / 3 > half || (r * SS_L + SS_L - 1) / 3 < half) continue;
            uint64_t x[16];
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = gp[r * N + w.lane + 64 * m];
            typename D::state_t st[16];
#pragma unroll
            for (int m = 0; m < 16; m++) st[m] = D::init(x[m]);
#pragma unroll
            for (int i = 0; i < SS_L; i++) {
                const typename D::state_t tie = (i == SS_L - 1) ? D::tie_mask(w.lane) : (typename D::state_t)0;
                const int pidx = r * SS_L + i;
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const double re = (double)D::next(st[m], tie), im = (double)D::next(st[m + 8], tie);
                    if (pidx / 3 == half) d[pidx % 3][m] = make_double2(re, im);
                }
            }
        }
        fft_fwd_n<3>(d, w.tw, w.tile, w.lane);
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int m = 0; m < 8; m++) Dg[half * 3 + q][m] = d[q][m];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nothing older than the ring's DMA on the VM counter
    const KeyRing ring = ring_setup(lds, SCRATCH_CPLX * sizeof(cplx));
    // Ring invariant (see KeyRing): inside ring_pass the ring's DMA loads are the only vector-memory operations in
    // flight; the stores of store_fourier_scaled3 are drained (vmcnt(0)) before the next pass starts.
#pragma unroll 1
    for (int i = 0; i < 2; i++) {
        cplx F[KP][8];
        zero_f(F);
        ring_pass(F, Dg, ring, SsKeyFn{ss_f, i});
        uint64_t rows[KP][16];
#pragma unroll
        for (int c = 0; c < KP; c++)
#pragma unroll
            for (int m = 0; m < 16; m++) rows[c][m] = 0;
        inv_add3(F, rows, w);
        if (active) store_fourier_scaled3(rows, o + (size_t)i * KP * H, w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the stores above are off the VM counter before the next pass
    }
    if (!active) return;
    {
        uint64_t in[KP][16];
#pragma unroll
        for (int q = 0; q < KP; q++)
#pragma unroll
            for (int m = 0; m < 16; m++) in[q][m] = gp[q * N + w.lane + 64 * m];
        store_fourier_scaled3(in, o + (size_t)2 * KP * H, w);
    }
}
#endif  // TXD_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// a10: CMux tree over the 8 selector GGSWs of a byte (SRV_AES:550-588).
// Unit = (byte, lut, acc_idx).  ggsw_f: [byte*8 + bit][lev 7][row 3][col 3][512].
// luts: accumulators [lut][byte%16][acc 2][3][N] with strides in words.
// out: [(lut*nbytes + byte)*8 + 4*acc + i] big LWEs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sample_extract_store(const uint64_t (&p0)[16], const uint64_t (&p1)[16],
                                                     const uint64_t (&p2)[16], int nth, uint64_t *o, const WaveCtx &w)
{
    // lwe[q*N + m] = a_q[nth - m] (m <= nth), -a_q[N + nth - m] otherwise; body = b[nth]
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (q == 0) poly_to_lds(p0, w.pl, w.lane); else poly_to_lds(p1, w.pl, w.lane);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int j = w.lane + 64 * m;
            int src = nth - j;
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = w.pl[src];
            o[q * N + j] = (x ^ neg) - neg;
        }
    }
    poly_to_lds(p2, w.pl, w.lane);
    if (w.lane == 0) o[2 * N] = w.pl[nth];
}

#ifndef LUT_WAVES_PER_EU
#define LUT_WAVES_PER_EU 1, 1
#endif
// pos: AES byte position (0..15) of every byte, i.e. which keyed table it takes; null = byte index mod 16 (whole blocks)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LUT_WAVES_PER_EU))) void k_lut_eval(const cplx *ggsw_f, const uint64_t *luts, size_t lut_stride,
                                                  size_t byte_stride, uint64_t *out, int nbytes, int nlut,
                                                  const cplx *tw_g, const uint8_t *pos, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds);
    w.tie_lane = w.lane & tie_on;
    // XCD-aware unit order: the 2 nlut units of a byte read the same eight selector GGSWs (4 MB), and workgroups go to the eight
    // XCDs round-robin, each XCD with an L2 of its own -- in launch order the units of a byte would sit on two XCDs and every
    // GGSW would cross the fabric twice (7.4 GB per launch counted for 4.1 GB of GGSWs, profiles/r4m_pmc_fetch_size_bench_command.csv).
    // Workgroup b therefore takes the units of logical workgroup (b mod 8) * (gridDim / 8) + b / 8: consecutive logical workgroups
    // run on ONE XCD at adjacent dispatch times (the grid is a multiple of 8; surplus workgroups leave).
    w.unit = (int)(((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * WAVES + (threadIdx.x >> 6));
    if (w.unit >= nbytes * nlut * 2) return;
    const int byte = w.unit / (nlut * 2), lut = (w.unit >> 1) % nlut, acc_idx = w.unit & 1;
    const int lane = w.lane;
    const int table = pos ? (int)pos[byte] : (byte & 15);
    const uint64_t *lp = luts + (size_t)lut * lut_stride + (size_t)table * byte_stride + (size_t)acc_idx * GL;
    uint64_t acc[KP][16];
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) acc[q][m] = lp[q * N + lane + 64 * m];
    const size_t ggsw_sz = (size_t)CBS_L * KP * KP * H;
#pragma unroll 1
    for (int bit = 0; bit < 8; bit++) {
        const int rot = 2 * N - (1 << bit);                 // X^(-2^bit)
        const cplx *gg = ggsw_f + ((size_t)byte * 8 + bit) * ggsw_sz;
        cplx F[KP][8];
        zero_f(F);
#pragma unroll
        for (int r = 0; r < KP; r++) {
            uint64_t lo[8], hi[8], diff[16];
            poly_to_lds(acc[r], w.pl, lane);
            rot_sub16(w.pl, acc[r], rot, lane, diff);
#pragma unroll
            for (int m = 0; m < 8; m++) { lo[m] = diff[m]; hi[m] = diff[m + 8]; }
            extprod_row<CBS_B, CBS_L, KP>(F, lo, hi, gg + (size_t)r * KP * H, (size_t)KP * KP * H, H, w);
        }
        inv_add3(F, acc, w);
    }
    const size_t big = BIG;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        uint64_t *o = out + (((size_t)lut * nbytes + byte) * 8 + acc_idx * 4 + i) * big;
        sample_extract_store(acc[0], acc[1], acc[2], 256 * i, o, w);
    }
}

// ------------------------------------------------------------------------------------------
// cmux (SRV_CMP:81-95 via tfhe cmux_assign): ct0 += g (x) (ct1 - ct0), batched: unit u uses
// selector sel[u] (index into ggsw_f), GLWEs ct0[u], ct1[u] (ct1 is not modified).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LUT_WAVES_PER_EU))) void k_cmux(uint64_t *ct0, const uint64_t *ct1, const cplx *ggsw_f, const int *sel,
                                              int units, const cplx *tw_g, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds);
    w.tie_lane = w.lane & tie_on;
    if (w.unit >= units) return;
    uint64_t *p0 = ct0 + (size_t)w.unit * GL;
    const uint64_t *p1 = ct1 + (size_t)w.unit * GL;
    const cplx *gg = ggsw_f + (size_t)sel[w.unit] * CBS_L * KP * KP * H;
    cplx F[KP][8];
    zero_f(F);
#pragma unroll 1
    for (int r = 0; r < KP; r++) {      // operands are read from global memory row by row, not held in registers
        uint64_t lo[8], hi[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            lo[m] = p1[r * N + w.lane + 64 * m] - p0[r * N + w.lane + 64 * m];
            hi[m] = p1[r * N + w.lane + 64 * m + 512] - p0[r * N + w.lane + 64 * m + 512];
        }
        extprod_row<CBS_B, CBS_L, KP>(F, lo, hi, gg + (size_t)r * KP * H, (size_t)KP * KP * H, H, w);
    }
    uint64_t a0[KP][16];
#pragma unroll
    for (int r = 0; r < KP; r++)
#pragma unroll
        for (int m = 0; m < 16; m++) a0[r][m] = p0[r * N + w.lane + 64 * m];
    inv_add3(F, a0, w);
#pragma unroll
    for (int r = 0; r < KP; r++)
#pragma unroll
        for (int m = 0; m < 16; m++) p0[r * N + w.lane + 64 * m] = a0[r][m];
}

// ------------------------------------------------------------------------------------------
// element-wise / permutation kernels on big LWEs (bandwidth-bound, 16-byte accesses where the
// 2049-word pitch allows: it does not -- 2049 is odd -- so 8-byte coalesced accesses)
// ------------------------------------------------------------------------------------------
// a2: known_rotate_keyed_lut (aes_he.rs:64-93): for block b, bit i: extract coefficient
// lut_idx*256 + byte value of accumulator acc_idx of byte.  One wave per output LWE.
__global__ __launch_bounds__(256) void k_known_rotate(const uint8_t *bytes /* nblocks x 16 */, const uint64_t *lut_set,
                                                      size_t byte_stride, uint64_t *out, int nbits)
{
    const int lane = threadIdx.x & 63, unit = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (unit >= nbits) return;
    const int blk = unit >> 7, i = unit & 127, byte = i >> 3, bit = i & 7;
    const int acc = bit >> 2, lut = bit & 3;
    const int nth = lut * 256 + bytes[blk * 16 + byte];
    const uint64_t *g = lut_set + (size_t)byte * byte_stride + (size_t)acc * GL;
    uint64_t *o = out + (size_t)unit * BIG;
    for (int q = 0; q < 2; q++)
        for (int j = lane; j < N; j += 64) {
            int src = nth - j;
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = g[q * N + src];
            o[q * N + j] = (x ^ neg) - neg;
        }
    if (lane == 0) o[2 * N] = g[2 * N + nth];
}

// a3: he_inv_mix_columns_precomp + he_inv_shift_rows (SRV_AES:195-265) fused:
// st[dst(r,c)] = s14[r,c'] + s11[r+1,c'] + s13[r+2,c'] + s9[r+3,c'] with c' = (4 - r + c) % 4
// forward variant (aes_he.rs:441-474 + 348-368): ShiftRows first, then MixColumns.
// sm: [mult][block][128][BIG]; inverse != 0: mults = x9,x11,x13,x14; else x1,x2,x3.
__global__ __launch_bounds__(256) void k_linear(const uint64_t *sm, uint64_t *st, int nblocks, int inverse)
{
    const size_t total = (size_t)nblocks * 128 * BIG;
    const size_t mstride = total;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t lwe = e / BIG, word = e % BIG;
        int blk = (int)(lwe >> 7), i = (int)(lwe & 127), byte = i >> 3, bit = i & 7;
        int c = byte >> 2, r = byte & 3;
        size_t base = (size_t)blk * 128;
        uint64_t v;
        if (inverse) {
            int cs = (4 - r + c) & 3;          // source column after the inverse shift
            // mixed[r, cs] = s14[r,cs] + s11[r+1,cs] + s13[r+2,cs] + s9[r+3,cs]
            size_t b14 = (base + (size_t)(4 * cs + r) * 8 + bit) * BIG + word;
            size_t b11 = (base + (size_t)(4 * cs + ((r + 1) & 3)) * 8 + bit) * BIG + word;
            size_t b13 = (base + (size_t)(4 * cs + ((r + 2) & 3)) * 8 + bit) * BIG + word;
            size_t b9 = (base + (size_t)(4 * cs + ((r + 3) & 3)) * 8 + bit) * BIG + word;
            v = sm[3 * mstride + b14] + sm[1 * mstride + b11] + sm[2 * mstride + b13] + sm[0 * mstride + b9];
        } else {
            // shifted[r', c] = s[r', (r' + c) % 4];  out[r,c] = 2 sh[r,c] + 3 sh[r+1,c] + sh[r+2,c] + sh[r+3,c]
            int r1 = (r + 1) & 3, r2 = (r + 2) & 3, r3 = (r + 3) & 3;
            size_t a2 = (base + (size_t)(4 * ((r + c) & 3) + r) * 8 + bit) * BIG + word;
            size_t a3 = (base + (size_t)(4 * ((r1 + c) & 3) + r1) * 8 + bit) * BIG + word;
            size_t a1 = (base + (size_t)(4 * ((r2 + c) & 3) + r2) * 8 + bit) * BIG + word;
            size_t a0 = (base + (size_t)(4 * ((r3 + c) & 3) + r3) * 8 + bit) * BIG + word;
            v = sm[1 * mstride + a2] + sm[2 * mstride + a3] + sm[0 * mstride + a1] + sm[0 * mstride + a0];
        }
        st[e] = v;
    }
}

// final byte-wise bit reversal (SRV_AES:182-189); ctr != 0 additionally applies ShiftRows of the
// last forward round and XORs the public ciphertext bits (adds 2^63 to the body).
__global__ __launch_bounds__(256) void k_finalize(const uint64_t *st, uint64_t *out, int nblocks, int ctr,
                                                  const uint8_t *ct_bytes)
{
    const size_t total = (size_t)nblocks * 128 * BIG;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t lwe = e / BIG, word = e % BIG;
        int blk = (int)(lwe >> 7), i = (int)(lwe & 127), byte = i >> 3, bit = 7 - (i & 7);
        int c = byte >> 2, r = byte & 3;
        int sbyte = ctr ? 4 * ((r + c) & 3) + r : byte;
        uint64_t v = st[(((size_t)blk * 128) + (size_t)sbyte * 8 + bit) * BIG + word];
        if (ctr && word == BIG - 1 && ((ct_bytes[blk * 16 + byte] >> bit) & 1)) v += (uint64_t)1 << 63;
        out[e] = v;
    }
}

// State columns (32 LWEs = the four bytes of one AES column of one block) gathered into / scattered from a compact list:
// the CTR schedule bootstraps every DISTINCT column of its second round once (tx_api.hip, ctr_round2_groups).
//   gather:  out[u] = st[col[u]],  col[u] = block * 4 + column
//   scatter: sm[lut][block * 4 + column] = in[lut][rep[block * 4 + column]]   for all nlut tables
__global__ __launch_bounds__(256) void k_gather_cols(const uint64_t *st, const int *col, uint64_t *out, int ncols)
{
    const size_t cw = (size_t)32 * BIG, total = (size_t)ncols * cw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        out[e] = st[(size_t)col[e / cw] * cw + e % cw];
}
__global__ __launch_bounds__(256) void k_scatter_cols(const uint64_t *in, const int *rep, uint64_t *sm, int ncols_all, int ncols_in, int nlut)
{
    const size_t cw = (size_t)32 * BIG, per = (size_t)ncols_all * cw, total = per * nlut;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t lut = e / per, r = e % per;
        sm[e] = in[(lut * ncols_in + (size_t)rep[r / cw]) * cw + r % cw];
    }
}

// LWE (big) -> GLWE with the message in the constant coefficient (glwe_conv.rs:12-44), batched.
// Output unit u reads input LWE (u / 16 * gstride + goff) * 16 + u % 16: with (gstride, goff) = (2, 0) / (2, 1)
// one launch gathers the even / odd 16-bit values of a tournament level (1, 0 = contiguous).
__global__ __launch_bounds__(256) void k_lwe_to_glwe(const uint64_t *lwe, uint64_t *glwe, int count, int gstride, int goff)
{
    const size_t total = (size_t)count * GL;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t u = e / GL; int w = (int)(e % GL), q = w / N, j = w % N;
        const uint64_t *a = lwe + (((u >> 4) * (size_t)gstride + (size_t)goff) * 16 + (u & 15)) * BIG;
        uint64_t v;
        if (q < 2) v = (j == 0) ? a[q * N] : (uint64_t)0 - a[q * N + N - j];
        else v = (j == 0) ? a[2 * N] : 0;
        glwe[e] = v;
    }
}
// ---- bit-sliced arithmetic helpers (inner-product mini-workload) ----
// t[u] = pool[i0[u]] + pool[i1[u]] (i1 < 0: copy): XOR of two encrypted bits at 2^63
__global__ __launch_bounds__(256) void k_lwe_gather_add(const uint64_t *pool, const int *i0, const int *i1, uint64_t *out, int count)
{
    const size_t total = (size_t)count * BIG;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t u = e / BIG; int w = (int)(e % BIG);
        uint64_t v = pool[(size_t)i0[u] * BIG + w];
        if (i1[u] >= 0) v += pool[(size_t)i1[u] * BIG + w];
        out[e] = v;
    }
}
// GLWE with the bit of pool[src[u]] (+ 2^63 if flip[u]) in the constant coefficient; src < 0: zero
__global__ __launch_bounds__(256) void k_build_glwe(const uint64_t *pool, const int *src, const int *flip, uint64_t *glwe, int count)
{
    const size_t total = (size_t)count * GL;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t u = e / GL; int w = (int)(e % GL), q = w / N, j = w % N;
        uint64_t v = 0;
        if (src[u] >= 0) {
            const uint64_t *a = pool + (size_t)src[u] * BIG;
            if (q < 2) v = (j == 0) ? a[q * N] : (uint64_t)0 - a[q * N + N - j];
            else v = (j == 0) ? a[2 * N] : 0;
        }
        if (q == 2 && j == 0 && flip[u]) v += (uint64_t)1 << 63;
        glwe[e] = v;
    }
}
// sample extract at 0 from GLWE u into pool[dst[u]]
__global__ __launch_bounds__(256) void k_extract0_scatter(const uint64_t *glwe, const int *dst, uint64_t *pool, int count)
{
    const size_t total = (size_t)count * BIG;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t u = e / BIG; int w = (int)(e % BIG);
        const uint64_t *g = glwe + u * GL;
        uint64_t v;
        if (w == 2 * N) v = g[2 * N];
        else { int q = w / N, m = w % N; v = (m == 0) ? g[q * N] : (uint64_t)0 - g[q * N + N - m]; }
        pool[(size_t)dst[u] * BIG + w] = v;
    }
}

// sample extract at 0 from GLWEs (batched)
__global__ __launch_bounds__(256) void k_extract0(const uint64_t *glwe, uint64_t *lwe, int count)
{
    const size_t total = (size_t)count * BIG;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t u = e / BIG; int w = (int)(e % BIG);
        const uint64_t *g = glwe + u * GL;
        uint64_t v;
        if (w == 2 * N) v = g[2 * N];
        else { int q = w / N, m = w % N; v = (m == 0) ? g[q * N] : (uint64_t)0 - g[q * N + N - m]; }
        lwe[e] = v;
    }
}

}  // namespace txd
