// tx_br2.hpp -- blind rotation at TWO waves per SIMD (round 2).
//
// Why: one wave alone on a SIMD issues one vector instruction per 4 cycles whatever its width, and every
// LDS / memory round trip of that wave is exposed.  Two co-resident waves interleave: 32-bit instructions
// (a third of the CMux step) issue in 2 cycles, and one wave's tile exchanges and key loads hide behind the
// other's FP64 arithmetic (tools/microbench/occ_bench.hip: 1.5x transform throughput per SIMD).
//
// What has to give: two waves per SIMD means 256 registers per wave and 8 accumulators per CU, and eight
// 24 KiB accumulators do not fit the 160 KiB of LDS.  So the two MASK polynomials of the accumulator live in
// registers (64) and only the BODY polynomial stays in LDS next to the wave's transpose tile:
//   LDS  = twiddles 22.5 KiB + 8 x (tile 8.1 KiB + body 8 KiB) = 151.5 KiB, one 512-thread workgroup per CU;
//   VGPR = mask 64 + three digit transforms 96 + ONE Fourier accumulator 32 + key 32 + temporaries.
// The multiply-accumulate and the inverse transform therefore run column by column (same products, same
// order, same arithmetic per polynomial as k_blind_rotate: bit-identical results).
//
// Work distribution: within a SIMD the older wave wins issue arbitration, so equal static shares would leave
// the younger wave finishing alone; the waves take units from a global counter instead (persistent workgroups,
// one per CU; every wave leaves the loop when the counter passes `count`).
#pragma once
#include "tx_kernels.hpp"

namespace txd {

constexpr int BR2_WAVES = 8;
constexpr size_t BR2_WAVE_BYTES = SCRATCH_CPLX * sizeof(cplx) + (size_t)N * sizeof(uint64_t);
constexpr size_t LDS_BYTES_BR2 = TW_ROWS * 64 * sizeof(cplx) + BR2_WAVES * BR2_WAVE_BYTES;

// digits of  s(X) X^rot - own  for one polynomial whose natural-order image is in LDS at s
__device__ __forceinline__ void br2_digits(const uint64_t *s, const uint64_t (&own)[16], uint32_t base, uint32_t thr, cplx (&d)[8])
{
    uint64_t x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = s[(base + 64 * m) & (N - 1)];
    issue_fence();
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const uint64_t n0 = (uint64_t)0 - (uint64_t)(((base + 64 * m) >> LOGN) & 1);
        const uint64_t n1 = (uint64_t)0 - (uint64_t)(((base + 64 * (m + 8)) >> LOGN) & 1);
        d[m] = make_double2(pbs_digit(((x[m] ^ n0) - n0) - own[m], thr),
                            pbs_digit(((x[m + 8] ^ n1) - n1) - own[m + 8], thr));
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_blind_rotate2(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g,
                     unsigned int *next_unit)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds, BR2_WAVE_BYTES);
    const int lane = w.lane;
    uint64_t *bl = reinterpret_cast<uint64_t *>(w.tile + SCRATCH_CPLX);     // body polynomial, natural order
    const uint32_t thr = 0x400000u - (uint32_t)(lane & 1);
    for (;;) {
        unsigned int u = 0;
        if (lane == 0) u = atomicAdd(next_unit, 1u);
        u = (unsigned int)__shfl((int)u, 0);
        if (u >= (unsigned int)count) break;
        const uint64_t *ct = lwe + (size_t)u * (n + 1);
        uint64_t a0[16], a1[16];
        {   // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101
            const int rot = 2 * N - (int)modswitch(ct[n]);
#pragma unroll
            for (int m = 0; m < 16; m++) {
                int src = lane + 64 * m - rot;
                uint64_t neg = 0;
                if (src < 0) { src += N; neg = ~neg; }
                if (src < 0) { src += N; neg = ~neg; }
                uint64_t x = test_vector(src);
                a0[m] = 0;
                a1[m] = 0;
                bl[lane + 64 * m] = (x ^ neg) - neg;
            }
        }
        uint64_t cti = ct[0];
        TXD_STAMP_DECL();
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            const uint32_t rot = modswitch(cti);
            cti = ct[(i + 1 < n) ? i + 1 : i];              // next mask element: a step ahead of its use
            const uint32_t base = ((uint32_t)lane - rot) & (2 * N - 1);
            cplx d[KP][8];
            // mask polynomials: staged through the tile for the rotation (lane permutation)
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 16; m++) w.pl[lane + 64 * m] = a0[m];
            wave_lds_fence();
            br2_digits(w.pl, a0, base, thr, d[0]);
#ifdef BR2_SEQ_FWD
            fft_fwd(d[0], w.tw, w.tile, lane);
#endif
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 16; m++) w.pl[lane + 64 * m] = a1[m];
            wave_lds_fence();
            br2_digits(w.pl, a1, base, thr, d[1]);
#ifdef BR2_SEQ_FWD
            fft_fwd(d[1], w.tw, w.tile, lane);
#endif
            {
                uint64_t own[16];
#pragma unroll
                for (int m = 0; m < 16; m++) own[m] = bl[lane + 64 * m];
                br2_digits(bl, own, base, thr, d[2]);
            }
            wave_lds_fence();
            TXD_STAMP(0);
#ifdef BR2_SEQ_FWD
            fft_fwd(d[2], w.tw, w.tile, lane);
#else
            fft_fwd_n<KP>(d, w.tw, w.tile, lane);
#endif
            TXD_STAMP(1);
            const cplx *row = bsk_f + (size_t)i * KP * KP * H;  // [row q][col c][512]
#pragma unroll
            for (int c = 0; c < KP; c++) {
                cplx F[8];
#pragma unroll
                for (int r = 0; r < 8; r++) F[r] = make_double2(0.0, 0.0);
                cplx g[8];
                load_key8(g, row + (size_t)c * H, lane);
#pragma unroll
                for (int q = 0; q < KP; q++) {
#ifdef BR2_NO_PREFETCH
                    fmac8r(F, d[q], g);
                    if (q + 1 < KP) load_key8(g, row + ((size_t)(q + 1) * KP + c) * H, lane);
#else
                    cplx gn[8];
                    if (q + 1 < KP) load_key8(gn, row + ((size_t)(q + 1) * KP + c) * H, lane);
                    fmac8r(F, d[q], g);
                    if (q + 1 < KP) {
#pragma unroll
                        for (int r = 0; r < 8; r++) g[r] = gn[r];
                    }
#endif
                }
                TXD_STAMP(2);
                fft_inv(F, w.tw, w.tile, lane);
                TXD_STAMP(3);
                if (c == 0) {
#pragma unroll
                    for (int m = 0; m < 8; m++) { a0[m] = torus_add_f64(a0[m], F[m].x); a0[m + 8] = torus_add_f64(a0[m + 8], F[m].y); }
                } else if (c == 1) {
#pragma unroll
                    for (int m = 0; m < 8; m++) { a1[m] = torus_add_f64(a1[m], F[m].x); a1[m + 8] = torus_add_f64(a1[m + 8], F[m].y); }
                } else {
                    uint64_t cur[16];
                    wave_lds_fence();
#pragma unroll
                    for (int m = 0; m < 16; m++) cur[m] = bl[lane + 64 * m];
                    issue_fence();
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        bl[lane + 64 * m] = torus_add_f64(cur[m], F[m].x);
                        bl[lane + 64 * (m + 8)] = torus_add_f64(cur[m + 8], F[m].y);
                    }
                }
                TXD_STAMP(4);
            }
        }
        TXD_STAMP_FLUSH(n);
        wave_lds_fence();
        uint64_t *o = acc_out + (size_t)u * GL;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            o[lane + 64 * m] = a0[m];
            o[N + lane + 64 * m] = a1[m];
            o[2 * N + lane + 64 * m] = bl[lane + 64 * m];
        }
        wave_lds_fence();
    }
}

}  // namespace txd
