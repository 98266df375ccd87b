// tx_br2.hpp -- blind rotation at TWO waves per SIMD (round 2): k_blind_rotate3.
//
// Why: one wave alone on a SIMD issues one vector instruction per 4 cycles whatever its width, and every
// LDS / memory round trip of that wave is exposed.  Two co-resident waves interleave: 32-bit instructions
// (a third of the CMux step) issue in 2 cycles, and one wave's tile exchanges and key loads hide behind the
// other's FP64 arithmetic (tools/microbench/occ_bench.hip: 1.5x transform throughput per SIMD).
//
// What has to give: two waves per SIMD means 256 registers per wave and 8 accumulators per CU, and eight
// 24 KiB accumulators do not fit the 160 KiB of LDS.  So the two MASK polynomials of the accumulator live in
// registers (64) and only the BODY polynomial stays in LDS next to the wave's transpose tile:
//   VGPR = mask 64 + three digit transforms 96 + ONE Fourier accumulator 32 + key 32 + temporaries.
// The multiply-accumulate and the inverse transform therefore run column by column (same products, same
// order, same arithmetic per polynomial as k_blind_rotate: bit-identical results).
//
// Work distribution: within a SIMD the OLDER wave wins issue arbitration (measured: 29 k against 45 k cycles per
// step), so two waves started together finish far apart.  Two 256-thread workgroups per CU instead of one of 512
// threads make that imbalance harmless: every workgroup spends the first half of its life as the younger one and
// the second half, after its neighbour has retired and a new one has arrived, as the older one; all workgroups
// take the same time and every SIMD holds two waves until the grid drains.  That needs the workgroup's LDS image
// in 80 KiB, see below.
#pragma once
#include "tx_kernels.hpp"

namespace txd {

// ---- LDS image of the generated kernels ----
//   pass-A 8 rows of 64 complex (as tx_device.hpp), pass-B 7 rows of EIGHT complex (the pass-B twiddle depends on
//   lane & 7 only): 9 088 B; then per wave the transpose tile of tx_device.hpp (pitch 65 complex, 8 320 B: every
//   b128 store and load of both exchanges is conflict-free without address arithmetic) and the 8 KiB body (blind
//   rotation) or the 24 KiB GLWE (trace).  Two 256-thread workgroups per CU: 9 088 + 4 x 16 512 = 75 136 B <= 81 920 B.
//   (The first versions of this layout carried the 16 256-byte twiddle table and needed an unpadded, XOR-swizzled tile
//   to fit; the split twist freed the space, and the swizzle's 16 v_xor per exchange went with it.)
constexpr int BR3_TWB = 8 * 64;                              // first pass-B entry (complex units)
constexpr int BR3_TW_CPLX = 8 * 64 + 7 * 8;
constexpr size_t BR3_TILE_BYTES = (size_t)SCRATCH_CPLX * sizeof(cplx);          // 8 320
constexpr size_t BR3_WAVE_BYTES = BR3_TILE_BYTES + (size_t)N * sizeof(uint64_t);
constexpr size_t LDS_BYTES_BR3 = BR3_TW_CPLX * sizeof(cplx) + WAVES * BR3_WAVE_BYTES;
static_assert(LDS_BYTES_BR3 <= 80 * 1024, "two workgroups per CU");

#ifdef TXD_WITH_BR3   // the hipcc version of this layout: readable specification of the generated loop, not built by default
struct Br3Ctx {
    const cplx *tw;     // pass-A rows 0..7, pass-B compact at BR3_TWB
    cplx *tile;         // transpose tile, pitch XPITCH (also staged as 1024 torus words)
    uint64_t *pl;       // the tile as 1024 torus words
    uint64_t *bl;       // body polynomial
    int lane;
};

// same arithmetic per polynomial as fft_fwd / fft_inv (tx_device.hpp); twiddles are read where they are used
__device__ __forceinline__ void fft_fwd_s(cplx (&v)[8], const Br3Ctx &w)
{
    const int lane = w.lane;
#pragma unroll
    for (int m = 1; m < 8; m++) v[m] = cmul(v[m], c16(m));
    dft8<+1>(v);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = cmul(v[r], w.tw[r * 64 + lane]);
    xchg_hi(v, w.tile, lane);
    dft8<+1>(v);
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmul(v[r], w.tw[BR3_TWB + (r - 1) * 8 + (lane & 7)]);
    xchg_lo(v, w.tile, lane);
    dft8<+1>(v);
}
__device__ __forceinline__ void fft_inv_s(cplx (&v)[8], const Br3Ctx &w)
{
    const int lane = w.lane;
    dft8<-1>(v);
    xchg_lo(v, w.tile, lane);
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmulc(v[r], w.tw[BR3_TWB + (r - 1) * 8 + (lane & 7)]);
    dft8<-1>(v);
    xchg_hi(v, w.tile, lane);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = cmulc(v[r], w.tw[r * 64 + lane]);
    dft8<-1>(v);
#pragma unroll
    for (int m = 1; m < 8; m++) v[m] = cmulc(v[m], c16(m));
}
// three polynomials together (software pipeline over the polynomials as fft_fwd_n)
__device__ __forceinline__ void fft_fwd3_s(cplx (&v)[KP][8], const Br3Ctx &w)
{
    const int lane = w.lane;
    {
        cplx t[8];
#pragma unroll
        for (int r = 0; r < 8; r++) t[r] = w.tw[r * 64 + lane];
        issue_fence();
#pragma unroll
        for (int p = 0; p < KP; p++) {
#pragma unroll
            for (int m = 1; m < 8; m++) v[p][m] = cmul(v[p][m], c16(m));
            dft8<+1>(v[p]);
#pragma unroll
            for (int r = 0; r < 8; r++) v[p][r] = cmul(v[p][r], t[r]);
            xchg_hi(v[p], w.tile, lane);
        }
    }
    {
        cplx t[7];
#pragma unroll
        for (int r = 1; r < 8; r++) t[r - 1] = w.tw[BR3_TWB + (r - 1) * 8 + (lane & 7)];
        issue_fence();
#pragma unroll
        for (int p = 0; p < KP; p++) {
            dft8<+1>(v[p]);
#pragma unroll
            for (int r = 1; r < 8; r++) v[p][r] = cmul(v[p][r], t[r - 1]);
            xchg_lo(v[p], w.tile, lane);
        }
    }
#pragma unroll
    for (int p = 0; p < KP; p++) dft8<+1>(v[p]);
}

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

#ifndef BR3_NUM_VGPR
#define BR3_NUM_VGPR 256
#endif
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(BR3_NUM_VGPR)))
void k_blind_rotate3(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g)
{
    extern __shared__ __align__(16) unsigned char lds[];
    cplx *tw = reinterpret_cast<cplx *>(lds);
    // tw_g is the [15][64] table of tx_api.hip: rows 0..7 as they are, rows 8..14 (pass B) depend on lane & 7 only
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * WAVES + wave;
    if (unit >= count) return;
    Br3Ctx w;
    w.tw = tw;
    w.tile = reinterpret_cast<cplx *>(lds + BR3_TW_CPLX * sizeof(cplx) + (size_t)wave * BR3_WAVE_BYTES);
    w.pl = reinterpret_cast<uint64_t *>(w.tile);
    w.bl = reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(w.tile) + BR3_TILE_BYTES);
    w.lane = lane;
    uint64_t *bl = w.bl;
    const uint32_t thr = 0x400000u - (uint32_t)(lane & 1);
    const uint64_t *ct = lwe + (size_t)unit * (n + 1);
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
        // mask polynomials: staged through the tile for the rotation (a lane permutation)
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < 16; m++) w.pl[lane + 64 * m] = a0[m];
        wave_lds_fence();
        br2_digits(w.pl, a0, base, thr, d[0]);
#ifdef BR3_SEQ
        wave_lds_fence();
        fft_fwd_s(d[0], w);
        issue_fence();
#endif
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < 16; m++) w.pl[lane + 64 * m] = a1[m];
        wave_lds_fence();
        br2_digits(w.pl, a1, base, thr, d[1]);
#ifdef BR3_SEQ
        wave_lds_fence();
        fft_fwd_s(d[1], w);
        issue_fence();
#endif
        {
            uint64_t own[16];
#pragma unroll
            for (int m = 0; m < 16; m++) own[m] = bl[lane + 64 * m];
            br2_digits(bl, own, base, thr, d[2]);
        }
        wave_lds_fence();
        TXD_STAMP(0);
#ifdef BR3_SEQ
        fft_fwd_s(d[2], w);
        issue_fence();
#else
        fft_fwd3_s(d, w);
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
#ifdef BR3_SEQ
                fmac8r(F, d[q], g);
                issue_fence();
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
            fft_inv_s(F, w);
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
    uint64_t *o = acc_out + (size_t)unit * GL;
#pragma unroll
    for (int m = 0; m < 16; m++) {
        o[lane + 64 * m] = a0[m];
        o[N + lane + 64 * m] = a1[m];
        o[2 * N + lane + 64 * m] = bl[lane + 64 * m];
    }
}

#endif  // TXD_WITH_BR3

// ------------------------------------------------------------------------------------------
// k_blind_rotate4: the same kernel with the CMux loop as generated assembly (tools/asmgen/brgen.py ->
// tx_br_asm.inc): fixed register map (256 VGPRs, nothing in scratch), counted waits, key ring of 12 complex
// registers.  The HIP part loads the twiddles, builds the rotated test vector in the wave's LDS slice and hands
// five scalars to the loop; the loop stores the accumulator itself.  Same LDS image as k_blind_rotate3.
// ------------------------------------------------------------------------------------------
#ifdef TXD_EXPERIMENTS   // superseded / reference form: only in libtfhe_mi355x_exp.so (tests, A/B timing)
__global__ __launch_bounds__(256, 2)
void k_blind_rotate4(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * WAVES + wave;
    if (unit >= count) return;
    unsigned char *tile = lds + BR3_TW_CPLX * sizeof(cplx) + (size_t)wave * BR3_WAVE_BYTES;
    uint64_t *bl = reinterpret_cast<uint64_t *>(tile + BR3_TILE_BYTES);
    const uint64_t *ct = lwe + (size_t)unit * (n + 1);
    {   // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101
        const int rot = 2 * N - (int)modswitch(ct[n]);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int src = lane + 64 * m - rot;
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = test_vector(src);
            bl[lane + 64 * m] = (x ^ neg) - neg;
        }
    }
    const uint32_t tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)tile);
    uint64_t *out = acc_out + (size_t)unit * GL;
#ifdef TXD_STAMPS   // per workgroup (its wave 0): s_memrealtime at the start and the end of the unit
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const int rt_slot = (wave == 0 && blockIdx.x < 4096) ? (int)blockIdx.x : -1;
#endif
#ifdef TXD_STAMPS   // diagnostic build: per-phase cycle sums of one wave (tools/br_only.py); every other wave writes to a sink
    unsigned long long *dbg = (blockIdx.x == TXD_STAMP_BLOCK && wave == TXD_STAMP_THREAD / 64) ? txd_stamps : txd_stamps_sink;
    if (lane == 0 && dbg == txd_stamps) txd_stamps[6] = (unsigned long long)n;
    asm volatile(
#include "tx_br_asm_stamps.inc"
        :
        : "s"(ct), "s"(bsk_f), "s"(out), "s"(n), "s"(tile_addr), "s"(tie_on), "s"(dbg)
        :
#include "tx_br_asm_clobbers.inc"
    );
    if (rt_slot >= 0 && (threadIdx.x & 63) == 0) { txd_wave_rt[2 * rt_slot] = rt0; txd_wave_rt[2 * rt_slot + 1] = __builtin_amdgcn_s_memrealtime(); }
#else
    asm volatile(
#include "tx_br_asm.inc"
        :
        : "s"(ct), "s"(bsk_f), "s"(out), "s"(n), "s"(tile_addr), "s"(tie_on)
        :
#include "tx_br_asm_clobbers.inc"
    );
#endif
}
#endif  // TXD_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// k_blind_rotate5: the generated loop with FAIR arbitration.  With two independent workgroups per CU the older wave of
// every SIMD wins each arbitration (25 k against 40 k cycles per step): the workgroups of a CU finish far apart and the
// last one runs unpaired -- about 4.5 ms of every launch, whatever its size.  Here the two waves of a SIMD belong to ONE
// 512-thread workgroup (waves w and w + 4: a workgroup's waves go to the four SIMDs cyclically), publish their step
// counters in LDS every step, and the one that is behind runs at a higher priority (s_setprio 2/3 against 0/1): both
// advance at the same rate and the eight units of a workgroup finish together.  LDS: twiddles + 8 x 16 KiB + counters.
// ------------------------------------------------------------------------------------------
constexpr int BR5_WAVES = 8;
// Scratch: the generated loop clobbers all 256 vector registers, so hipcc keeps what is live across it (threadIdx.x,
// and the vector register it needs for scalar spills) in a small scratch frame (private_segment_fixed_size = 92 bytes
// per lane), stored once before and reloaded once after each UNIT (768 CMux steps).  The loop itself issues no scratch
// instruction and its prologue drains vmcnt(0), so that traffic never sits next to the counted waits.  Taking the
// lane id from v_mbcnt instead (nothing live in a VGPR) does not build: "cannot find enough VGPRs for wwm-regalloc".
// tests/test_asmgen.py bounds the frame in the built code object and checks that the .inc files hold no scratch op.
constexpr size_t LDS_BYTES_BR5 = BR3_TW_CPLX * sizeof(cplx) + BR5_WAVES * BR3_WAVE_BYTES + BR5_WAVES * sizeof(uint32_t);

// Persistent: G = gridDim.x workgroups (one per CU), wave w of workgroup b takes units w * G + b + k * 8 G, k = 0, 1, ...
// -- a static share, so that no wave picks up an extra 11 ms unit at the end, and wave-major, so that a launch that
// does not fill every wave still spreads over every CU (the first G units go to wave 0 of every workgroup, the next G
// to wave 1, ...; waves 0-3 sit on the four SIMDs, waves 4-7 double them up).  The step counters keep counting
// across units.  A wave leaves when its share is done (it then reads as "finished" to its partner).
__global__ __launch_bounds__(512, 2)
void k_blind_rotate5(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    cplx *tw = reinterpret_cast<cplx *>(lds);
    uint32_t *prog = reinterpret_cast<uint32_t *>(lds + BR3_TW_CPLX * sizeof(cplx) + BR5_WAVES * BR3_WAVE_BYTES);
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int first = wave * (int)gridDim.x + (int)blockIdx.x, stride = (int)gridDim.x * BR5_WAVES;
    // a wave without a unit is "finished" from the start: its partner is never behind it
    if ((threadIdx.x & 63) == 0) prog[wave] = (first < count) ? 0u : 0xFFFFFFFFu;
    __syncthreads();
    unsigned char *tile = lds + BR3_TW_CPLX * sizeof(cplx) + (size_t)wave * BR3_WAVE_BYTES;
    uint64_t *bl = reinterpret_cast<uint64_t *>(tile + BR3_TILE_BYTES);
    const uint32_t tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)tile);
    const uint32_t prog_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)prog);
    const uint32_t prog_a = prog_addr + 4u * (uint32_t)wave, prog_b = prog_addr + 4u * (uint32_t)(wave ^ 4);
    uint32_t done_steps = 0;
#ifdef BR5_STAGGER      // experiment: the second wave of every SIMD starts half a step late (waves of one program can lock in phase)
    if (wave >= 4) for (int k = 0; k < BR5_STAGGER; k++) __builtin_amdgcn_s_sleep(127);
#endif
#ifdef TXD_STAMPS
    if ((threadIdx.x & 63) == 0 && first < 4096) txd_wave_rt[2 * first] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int unit = first; unit < count; unit += stride) {
        const int lane = threadIdx.x & 63;
        const uint64_t *ct = lwe + (size_t)unit * (n + 1);
        {   // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101
            const int rot = 2 * N - (int)modswitch(ct[n]);
#pragma unroll
            for (int m = 0; m < 16; m++) {
                int src = lane + 64 * m - rot;
                uint64_t neg = 0;
                if (src < 0) { src += N; neg = ~neg; }
                if (src < 0) { src += N; neg = ~neg; }
                uint64_t x = test_vector(src);
                bl[lane + 64 * m] = (x ^ neg) - neg;
            }
        }
        uint64_t *out = acc_out + (size_t)unit * GL;
#ifdef TXD_STAMPS
        unsigned long long *dbg = (unit == TXD_STAMP_BLOCK * 4 + TXD_STAMP_THREAD / 64) ? txd_stamps : txd_stamps_sink;
        if (lane == 0 && dbg == txd_stamps) txd_stamps[6] = (unsigned long long)n;
        asm volatile(
#include "tx_br_asm_fair_stamps.inc"
            :
            : "s"(ct), "s"(bsk_f), "s"(out), "s"(n), "s"(tile_addr), "s"(prog_a), "s"(prog_b), "s"(done_steps), "s"(tie_on), "s"(dbg)
            :
#include "tx_br_asm_clobbers.inc"
        );
#else
        asm volatile(
#include "tx_br_asm_fair.inc"
            :
            : "s"(ct), "s"(bsk_f), "s"(out), "s"(n), "s"(tile_addr), "s"(prog_a), "s"(prog_b), "s"(done_steps), "s"(tie_on)
            :
#include "tx_br_asm_clobbers.inc"
        );
#endif
        done_steps += (uint32_t)n;
    }
    if ((threadIdx.x & 63) == 0) prog[wave] = 0xFFFFFFFFu;      // finished: the partner is never "behind" this wave again
#ifdef TXD_STAMPS
    if ((threadIdx.x & 63) == 0 && first < 4096) txd_wave_rt[2 * first + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------
// k_blind_rotate6: the COOPERATIVE blind rotation for launches of at most one unit per SIMD (the 8-blocks-per-GPU share
// of the medium instance on an 8-GPU node: 1024 units on 256 CUs; a single AES block: 128 units).  A CMux chain is 768
// dependent steps, so with one unit per SIMD the step time is the launch time, and no schedule of ONE wave shortens it
// (a generated one-wave loop measured equal to k_blind_rotate).  Here a workgroup of THREE waves works on one unit, wave
// c owning polynomial c of the accumulator for the whole chain: it rotates and decomposes its polynomial, transforms the
// digits, publishes the transform through its transpose tile (idle at that point), multiply-accumulates column c
// against the three published transforms, inverts and updates its polynomial -- one transform each way per wave and
// step instead of three, the accumulator never moves, 8 KiB per wave and step cross through LDS, two s_barrier per
// step.  The loop is generated (tools/asmgen/brgen3.py -> tx_br3_asm.inc, 168 registers: three waves per SIMD when the
// launch fills the chip, four workgroups of 34 048 B LDS per CU); same products in the same order: bit-identical to
// k_blind_rotate and the oracle.
// Fair arbitration between the workgroups of a CU (round 4): the SIMD arbiter serves the oldest wave first, so the four
// workgroups of a CU used to finish at 3.9, 5.0, 6.5 and 8.1 ms of a 1024-unit launch, the last one with ever fewer
// neighbours to fill its stalls.  Every workgroup now publishes its step counter in its CU's entry of `fair_tab` (global
// memory; the entry is found by the hardware's own CU number: XCC_ID and the CU / SH / SE fields of HW_ID; the word in the
// entry by a counter the workgroups of the CU draw from), and the generated loop sets s_setprio by the lead over the slowest
// workgroup of the same launch on the same CU (tools/asmgen/brgen3.py, fair_step).  Priorities only: no result depends on it.
// ------------------------------------------------------------------------------------------
constexpr int BR6_WAVES = 3;
constexpr int FAIR_TAB_ENTRIES = 4096, FAIR_TAB_WORDS = 8;      // [xcc 4 bits | se, sh, cu 8 bits] x (4 progress words, 1 slot counter, 3 unused)
#include "tx_br3_asm_consts.inc"
// + 1 KiB behind the tiles where the key prefetch drops what it touched, + the workgroup's word index in the fairness table
constexpr size_t LDS_BYTES_BR6 = BR3_TW_CPLX * sizeof(cplx) + BR6_WAVES * BR3_TILE_BYTES + 1024 + 16;
static_assert(4 * LDS_BYTES_BR6 <= 160 * 1024, "four workgroups per CU");

__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_blind_rotate6(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g,
                     uint32_t *fair_tab, uint32_t epoch_shl16, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];         // (no static __shared__ here: the generated loop reads the twiddles at LDS address 0)
    uint32_t &fair_slot = *reinterpret_cast<uint32_t *>(lds + LDS_BYTES_BR6 - 16);
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int unit = (int)blockIdx.x;                   // the grid is `count` workgroups
    if (unit >= count) return;                          // (uniform per workgroup: no wave is left alone at a barrier)
    // this CU's entry of the fairness table: s_getreg_b32 HW_REG_XCC_ID (register 20, 4 bits) and HW_REG_HW_ID (register 4:
    // cu_id [11:8], sh_id [12], se_id [15:13]); simm16 = (size - 1) << 11 | offset << 6 | register
    const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    const uint32_t cu8 = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 8 << 6 | 4);
    uint32_t *fair_entry = fair_tab + (size_t)(((xcc & 15u) << 8) | (cu8 & 255u)) * FAIR_TAB_WORDS;
    // consecutive draws: distinct words for the FIRST four workgroups a CU hosts in a launch.  A later arrival (a CU hosts 5-7
    // workgroups in turn when the placement is late) may draw the word of a workgroup that is still running; the two then
    // overwrite each other's step counters and a finisher's 0xFFFF marker may briefly stand in a runner's word.  Only
    // s_setprio is affected -- never a result -- and the arbitration is noisier in exactly that case.
    if (threadIdx.x == 0) fair_slot = atomicAdd(fair_entry + 4, 1u) & 3u;
    unsigned char *tile0 = lds + BR3_TW_CPLX * sizeof(cplx);
    unsigned char *tile = tile0 + (size_t)wave * BR3_TILE_BYTES;
    uint64_t *pl = reinterpret_cast<uint64_t *>(tile);
    const uint64_t *ct = lwe + (size_t)unit * (n + 1);
    if (wave == KP - 1) {   // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101: the body polynomial belongs to the last wave
        const int rot = 2 * N - (int)modswitch(ct[n]);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int src = lane + 64 * m - rot;
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = test_vector(src);
            pl[lane + 64 * m] = (x ^ neg) - neg;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) pl[lane + 64 * m] = 0;
    }
    __syncthreads();
    const uint32_t tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)tile);
    const uint32_t tile0_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)tile0);
    const cplx *col = bsk_f + (size_t)wave * H;                         // key[q][c] of row i: col + (i * 9 + q * 3) * H
    uint64_t *out = acc_out + (size_t)unit * GL + (size_t)wave * N;
    // key prefetch: every step this wave touches one 1 KiB piece (of 72) of the key row BR6_PF_ROWS steps ahead, so that the row is
    // in L2 when it is needed; workgroups go to the XCDs round-robin, so (blockIdx.x >> 3) counts the workgroups of an XCD, and
    // consecutive ones cover consecutive triples of pieces.  Relative to `col` (this wave's column of the current row).
    const uint32_t piece = (3u * (blockIdx.x >> 3) + (uint32_t)wave) % (uint32_t)(KP * KP * 8);
    const uint32_t pf_off = (uint32_t)BR6_PF_ROWS * (uint32_t)(KP * KP * H * sizeof(cplx)) + piece * 1024u - (uint32_t)wave * (uint32_t)(H * sizeof(cplx));
    const uint32_t fair_off = __builtin_amdgcn_readfirstlane(fair_slot * 4u | pf_off << 4);      // (fair_slot: written before the __syncthreads above)
#ifdef TXD_STAMPS_BR6   // diagnostic build: per-phase cycle sums of wave TXD_STAMP_THREAD / 64 of workgroup TXD_STAMP_BLOCK
    unsigned long long *dbg = (blockIdx.x == TXD_STAMP_BLOCK && wave == (TXD_STAMP_THREAD / 64) % 3) ? txd_stamps : txd_stamps_sink;
    if (lane == 0 && dbg == txd_stamps) txd_stamps[6] = (unsigned long long)n;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const int rt_slot = (wave == 0 && unit < 4096) ? unit : -1;
    asm volatile(
#include "tx_br3_asm_stamps.inc"
        :
        : "s"(ct), "s"(col), "s"(out), "s"(n), "s"(tile_addr), "s"(tile0_addr), "s"(fair_entry), "s"(fair_off), "s"(epoch_shl16), "s"(tie_on), "s"(dbg)
        :
#include "tx_br3_asm_clobbers.inc"
    );
    if (rt_slot >= 0 && (threadIdx.x & 63) == 0) { txd_wave_rt[2 * rt_slot] = rt0; txd_wave_rt[2 * rt_slot + 1] = __builtin_amdgcn_s_memrealtime(); }
#else
    asm volatile(
#include "tx_br3_asm.inc"
        :
        : "s"(ct), "s"(col), "s"(out), "s"(n), "s"(tile_addr), "s"(tile0_addr), "s"(fair_entry), "s"(fair_off), "s"(epoch_shl16), "s"(tie_on)
        :
#include "tx_br3_asm_clobbers.inc"
    );
#endif
}

// (k_blind_rotate7 itself is superseded by k_blind_rotate8 below and lives in the experiment build only; its constants serve both)
// ------------------------------------------------------------------------------------------
// k_blind_rotate7: the cooperative loop of k_blind_rotate6 with FOUR units in one workgroup of twelve waves (768 threads, one
// workgroup per CU), for launches of exactly four units per CU -- the 8-blocks-per-GPU share of the medium instance: 1024 units.
// Why a kernel of its own: four three-wave workgroups per CU need every SIMD to take exactly three waves of 168 registers
// (504 of 512), and the hardware's placement of independent workgroups does not always find that packing: of 1024 workgroups
// 4 to 36 start only when the first ones have finished (one CU then hosts 5 to 7 workgroups in turn while others sit at 3,
// profiles/r4h_coop_blind_rotation_placement.txt), which turns 6.4-6.7 ms into 9-10 ms.  The waves of ONE workgroup go to the
// SIMDs in cyclic order, so twelve waves are three per SIMD by construction, and unit u = waves 3u, 3u + 1, 3u + 2 sits on
// three different SIMDs.  s_barrier then spans all twelve waves: the four units advance in lockstep, which is also what
// the fair arbitration of k_blind_rotate6 drives its workgroups to, without the table (launch number 0 switches it off);
// running one half of the workgroup a barrier behind the other was measured slower (profiles/r4n_*).
// Same generated loop, same results.  Units past `count` in the last workgroup shadow the last unit (every wave must reach
// every barrier): they compute and store the same words.
// ------------------------------------------------------------------------------------------
constexpr int BR7_UNITS = 4;
// The generated loop derives its key-prefetch dump area from TILE0 + 3 tiles; here that is the NEXT unit's first tile.  The
// committed stream has no prefetch (BR3_PF_ROWS = 0): regenerating with one would corrupt three of every four units.
static_assert(BR6_PF_ROWS == 0, "k_blind_rotate7 passes the unit's own tiles as TILE0: a key prefetch would dump into the next unit's tile");
constexpr size_t LDS_BYTES_BR7 = BR3_TW_CPLX * sizeof(cplx) + BR7_UNITS * BR6_WAVES * BR3_TILE_BYTES + 1024;

#ifdef TXD_EXPERIMENTS   // superseded by k_blind_rotate8 (lockstep: 6.44 against 5.9 ms in the pipeline)
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_blind_rotate7(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g, uint32_t *fair_tab, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int u = wave / BR6_WAVES, c = wave % BR6_WAVES;
    const int unit = min((int)blockIdx.x * BR7_UNITS + u, count - 1);
    unsigned char *tile0 = lds + BR3_TW_CPLX * sizeof(cplx) + (size_t)(u * BR6_WAVES) * BR3_TILE_BYTES;      // the unit's three tiles
    unsigned char *tile = tile0 + (size_t)c * BR3_TILE_BYTES;
    uint64_t *pl = reinterpret_cast<uint64_t *>(tile);
    const uint64_t *ct = lwe + (size_t)unit * (n + 1);
    if (c == KP - 1) {      // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101
        const int rot = 2 * N - (int)modswitch(ct[n]);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int src = lane + 64 * m - rot;
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = test_vector(src);
            pl[lane + 64 * m] = (x ^ neg) - neg;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) pl[lane + 64 * m] = 0;
    }
    __syncthreads();
    const uint32_t tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)tile);
    const uint32_t tile0_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)tile0);
    const cplx *col = bsk_f + (size_t)c * H;
    uint64_t *out = acc_out + (size_t)unit * GL + (size_t)c * N;
    const uint32_t piece = (3u * (blockIdx.x >> 3) * BR7_UNITS + (uint32_t)wave) % (uint32_t)(KP * KP * 8);
    const uint32_t pf_off = (uint32_t)BR6_PF_ROWS * (uint32_t)(KP * KP * H * sizeof(cplx)) + piece * 1024u - (uint32_t)c * (uint32_t)(H * sizeof(cplx));
    const uint32_t fair_off = __builtin_amdgcn_readfirstlane(pf_off << 4);
    const uint32_t no_epoch = 0;        // nobody to be fair to: the barriers keep the four units together
#ifdef TXD_STAMPS_BR6
    unsigned long long *dbg = (blockIdx.x == TXD_STAMP_BLOCK && wave == (TXD_STAMP_THREAD / 64) % 3) ? txd_stamps : txd_stamps_sink;
    if (lane == 0 && dbg == txd_stamps) txd_stamps[6] = (unsigned long long)n;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const int rt_slot = (c == 0 && unit < 4096 && (int)blockIdx.x * BR7_UNITS + u < count) ? unit : -1;
    asm volatile(
#include "tx_br3_asm_stamps.inc"
        :
        : "s"(ct), "s"(col), "s"(out), "s"(n), "s"(tile_addr), "s"(tile0_addr), "s"(fair_tab), "s"(fair_off), "s"(no_epoch), "s"(tie_on), "s"(dbg)
        :
#include "tx_br3_asm_clobbers.inc"
    );
    if (rt_slot >= 0 && (threadIdx.x & 63) == 0) { txd_wave_rt[2 * rt_slot] = rt0; txd_wave_rt[2 * rt_slot + 1] = __builtin_amdgcn_s_memrealtime(); }
#else
    asm volatile(
#include "tx_br3_asm.inc"
        :
        : "s"(ct), "s"(col), "s"(out), "s"(n), "s"(tile_addr), "s"(tile0_addr), "s"(fair_tab), "s"(fair_off), "s"(no_epoch), "s"(tie_on)
        :
#include "tx_br3_asm_clobbers.inc"
    );
#endif
}
#endif  // TXD_EXPERIMENTS

__device__ __forceinline__ uint32_t lds_addr(unsigned char *p)
{
    return __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)p);
}

// ------------------------------------------------------------------------------------------
// k_blind_rotate8 (round 5; what run_br launches between three and four units per CU -- the 8-blocks-per-GPU share of the medium
// instance: 1024 units): four units in ONE twelve-wave workgroup per CU, three waves per SIMD by construction (k_blind_rotate7's
// shape, section "Placement" of DESIGN_HISTORY.md) -- WITHOUT k_blind_rotate7's lockstep: the generated loop's soft-barrier form (tools/asmgen/brgen3.py, SOFT: tx_br3s_asm.inc).
// The three waves of a unit meet at arrival words in LDS that they poll (s_sleep between polls), the four units drift apart and
// fill each other's stalls, and the fair arbitration between them reads their step counters from LDS.  No s_barrier behind the
// __syncthreads below; a wave gives up polling after 2^18 tries and stores 1 to the context's fault word (the launch ENDS, its
// results are invalid, and the host fails the call: tx_api.hip timing_collect).  Same
// arithmetic, same results.  Inside the round pipeline 5.85-5.95 ms per 1024 units against 6.44 for k_blind_rotate7 (same box,
// alternating, profiles/r5g_*); without the fair arbitration 6.8 ms.  The spare kilobyte behind the twelve tiles holds the arrival words (16 bytes per unit) and the four
// step counters.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_blind_rotate8(const uint64_t *lwe, uint64_t *acc_out, int count, int n, const cplx *bsk_f, const cplx *tw_g, int tie_on, uint32_t *fault)
{
    extern __shared__ __align__(16) unsigned char lds[];
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    unsigned char *sync = lds + BR3_TW_CPLX * sizeof(cplx) + (size_t)(BR7_UNITS * BR6_WAVES) * BR3_TILE_BYTES;      // the spare kilobyte
    if (threadIdx.x < 5 * BR7_UNITS) reinterpret_cast<uint32_t *>(sync)[threadIdx.x] = 0;                            // arrival words + step counters
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int u = wave / BR6_WAVES, c = wave % BR6_WAVES;
    const int unit = min((int)blockIdx.x * BR7_UNITS + u, count - 1);
    unsigned char *tile0 = lds + BR3_TW_CPLX * sizeof(cplx) + (size_t)(u * BR6_WAVES) * BR3_TILE_BYTES;
    unsigned char *tile = tile0 + (size_t)c * BR3_TILE_BYTES;
    uint64_t *pl = reinterpret_cast<uint64_t *>(tile);
    const uint64_t *ct = lwe + (size_t)unit * (n + 1);
    if (c == KP - 1) {      // acc = (0, 0, TV * X^(-b~))      pbs.rs:84-101
        const int rot = 2 * N - (int)modswitch(ct[n]);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int src = lane + 64 * m - rot;
            uint64_t neg = 0;
            if (src < 0) { src += N; neg = ~neg; }
            if (src < 0) { src += N; neg = ~neg; }
            uint64_t x = test_vector(src);
            pl[lane + 64 * m] = (x ^ neg) - neg;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) pl[lane + 64 * m] = 0;
    }
    __syncthreads();
    const uint32_t tile_addr = lds_addr(tile), tile0_addr = lds_addr(tile0);
    const uint32_t flags = lds_addr(sync + 16 * u), myflag = flags + 4u * (uint32_t)c;
    const uint32_t prog = lds_addr(sync + 16 * BR7_UNITS), myprog = prog + 4u * (uint32_t)u;
    const cplx *col = bsk_f + (size_t)c * H;
    uint64_t *out = acc_out + (size_t)unit * GL + (size_t)c * N;
    asm volatile(
#include "tx_br3s_asm.inc"
        :
        : "s"(ct), "s"(col), "s"(out), "s"(n), "s"(tile_addr), "s"(tile0_addr), "s"(flags), "s"(myflag), "s"(prog), "s"(myprog), "s"(tie_on), "s"(fault)
        :
#include "tx_br3_asm_clobbers.inc"
    );
}

// ------------------------------------------------------------------------------------------
// k_extract_trace4: k_extract_trace with trace_lds() replaced by the generated loop of tools/asmgen/trgen.py
// (tx_trace_asm.inc): one wave per SIMD as before, but the six digit transforms of a step are parked in accumulation
// registers, which leaves room for two key register sets (the LDS read of the next key polynomial is in flight
// while the current one is multiplied) and three-way interleaved transforms in both directions.  LDS image: the
// compact twiddles of k_blind_rotate3/4, then per wave the 8 320-byte tile (its first 8 KiB = that wave's ring slot) and the
// 24 KiB GLWE.  Same ring protocol (KeyRing, tx_kernels.hpp): every wave takes part in every barrier, so idle
// waves of the last workgroup shadow the last unit.  Bit-identical to k_extract_trace and the oracle.
// ------------------------------------------------------------------------------------------
constexpr size_t TR4_WAVE_BYTES = BR3_TILE_BYTES + (size_t)GL * sizeof(uint64_t);
#ifndef TR4_EXTRA_SLOTS
#define TR4_EXTRA_SLOTS 0      // ring mode experiments (TR_SLOTS of trgen.py): 8 KiB slots behind the last wave's GLWE
#endif
constexpr size_t LDS_BYTES_TR4 = BR3_TW_CPLX * sizeof(cplx) + WAVES * TR4_WAVE_BYTES + TR4_EXTRA_SLOTS * 8192;

// a6: buf = acc * X^(-lvl); body[0] += 2^(63 - 2 level); extract@0; >> 10; LWE -> GLWE const  (as k_extract_trace),
// written into the wave's GLWE region of LDS
__device__ __forceinline__ void tr4_extract_to_lds(const uint64_t *a, int lvl, uint64_t *gl, int lane)
{
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            int j = lane + 64 * m, s = j + lvl;
            uint64_t x = (s < N) ? a[q * N + s] : (uint64_t)0 - a[q * N + s - N];
            gl[q * N + j] = (j == 0) ? (x >> LOGN) : (uint64_t)0 - (((uint64_t)0 - x) >> LOGN);
        }
    const uint64_t b0 = (a[2 * N + lvl] + ((uint64_t)1 << (63 - CBS_B * (lvl + 1)))) >> LOGN;
#pragma unroll
    for (int m = 0; m < 16; m++) gl[2 * N + lane + 64 * m] = (lane + 64 * m == 0) ? b0 : 0;
}
// a7: the generated trace loop on the GLWE behind `tile` (tools/asmgen/trgen.py)
__device__ __forceinline__ void tr4_trace_asm(const cplx *auto_f, uint32_t split_bytes, int wave, uint32_t slot0_addr, uint32_t tile_addr, int tie_on)
{
#ifdef TXD_STAMPS_TRACE
    unsigned long long *dbg = (blockIdx.x == TXD_STAMP_BLOCK && wave == TXD_STAMP_THREAD / 64) ? txd_stamps : txd_stamps_sink;
    if ((threadIdx.x & 63) == 0 && dbg == txd_stamps) txd_stamps[6] = (unsigned long long)LOGN;
    asm volatile(
#include "tx_trace_asm_stamps.inc"
        :
        : "s"(auto_f), "s"(split_bytes), "s"(wave), "s"(slot0_addr), "s"(tile_addr), "s"(tie_on), "s"(dbg)
        :
#include "tx_trace_asm_clobbers.inc"
    );
#else
    asm volatile(
#include "tx_trace_asm.inc"
        :
        : "s"(auto_f), "s"(split_bytes), "s"(wave), "s"(slot0_addr), "s"(tile_addr), "s"(tie_on)
        :
#include "tx_trace_asm_clobbers.inc"
    );
#endif
}
// a8 + a9: the generated scheme-switching body on the GLev element behind `tile` (tools/asmgen/ssgen.py)
__device__ __forceinline__ void tr4_ss_asm(const cplx *ss_f, cplx *o, int wave, uint32_t slot0_addr, uint32_t tile_addr, int tie_on)
{
#ifdef TXD_STAMPS_SS
    unsigned long long *dbg = (blockIdx.x == TXD_STAMP_BLOCK && wave == TXD_STAMP_THREAD / 64) ? txd_stamps : txd_stamps_sink;
    if ((threadIdx.x & 63) == 0 && dbg == txd_stamps) txd_stamps[6] = 1ull;
    asm volatile(
#include "tx_ss_asm_stamps.inc"
        :
        : "s"(ss_f), "s"(o), "s"(wave), "s"(slot0_addr), "s"(tile_addr), "s"(tie_on), "s"(dbg)
        :
#include "tx_trace_asm_clobbers.inc"
    );
#else
    asm volatile(
#include "tx_ss_asm.inc"
        :
        : "s"(ss_f), "s"(o), "s"(wave), "s"(slot0_addr), "s"(tile_addr), "s"(tie_on)
        :
#include "tx_trace_asm_clobbers.inc"
    );
#endif
}
// twiddles into LDS (compact image), then this wave's slice: returns the tile; slot0 = wave 0's tile
__device__ __forceinline__ unsigned char *tr4_setup(unsigned char *lds, const cplx *tw_g, int &wave, unsigned char *&slot0)
{
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) tw[i] = tw_g[i];
    if (threadIdx.x < 7 * 8) tw[BR3_TWB + threadIdx.x] = tw_g[(8 + (threadIdx.x >> 3)) * 64 + (threadIdx.x & 7)];
    __syncthreads();
    wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    slot0 = lds + BR3_TW_CPLX * sizeof(cplx);
    return slot0 + (size_t)wave * TR4_WAVE_BYTES;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_extract_trace4(const uint64_t *acc, uint64_t *glev, int count, const cplx *auto_f, size_t split_stride, const cplx *tw_g, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    int wave; unsigned char *slot0;
    unsigned char *tile = tr4_setup(lds, tw_g, wave, slot0);
    const int lane = threadIdx.x & 63;
    int unit = blockIdx.x * WAVES + wave;
    const bool active = unit < count * CBS_L;       // idle waves of the last workgroup redo the last unit (in their own LDS):
    if (!active) unit = count * CBS_L - 1;          // needed by the ring form of the loop (barriers); harmless with streamed keys
    uint64_t *gl = reinterpret_cast<uint64_t *>(tile + BR3_TILE_BYTES);
    tr4_extract_to_lds(acc + (size_t)(unit / CBS_L) * GL, unit % CBS_L, gl, lane);
    tr4_trace_asm(auto_f, (uint32_t)(split_stride * sizeof(cplx)), wave, lds_addr(slot0), lds_addr(tile), tie_on);
    if (!active) return;
    uint64_t *o = glev + (size_t)unit * GL;
    const int lane2 = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) o[q * N + lane2 + 64 * m] = gl[q * N + lane2 + 64 * m];
}

// ------------------------------------------------------------------------------------------
// a8 + a9 with the generated body (tools/asmgen/ssgen.py -> tx_ss_asm.inc): k_scheme_switch with the explicit
// schedule of the trace loop.  Same LDS image as k_extract_trace4 (the GLev element sits where that kernel keeps its
// GLWE), same ring protocol: every wave takes part in every barrier, so idle waves of the last workgroup shadow the
// last unit -- and store the same values to the same place as the wave that owns it.  Bit-identical to
// k_scheme_switch and the oracle.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_scheme_switch4(const uint64_t *glev, cplx *ggsw_f, int units, const cplx *ss_f, const cplx *tw_g, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    int wave; unsigned char *slot0;
    unsigned char *tile = tr4_setup(lds, tw_g, wave, slot0);
    const int lane = threadIdx.x & 63;
    int unit = blockIdx.x * WAVES + wave;
    if (unit >= units) unit = units - 1;
    uint64_t *gl = reinterpret_cast<uint64_t *>(tile + BR3_TILE_BYTES);
    const uint64_t *gp = glev + (size_t)unit * GL;
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) gl[q * N + lane + 64 * m] = gp[q * N + lane + 64 * m];
    tr4_ss_asm(ss_f, ggsw_f + (size_t)unit * KP * KP * H, wave, lds_addr(slot0), lds_addr(tile), tie_on);
}

// ------------------------------------------------------------------------------------------
// a10 with the generated CMux tree (tools/asmgen/lutgen.py -> tx_lut_asm.inc): k_lut_eval with the explicit schedule of the
// trace loop.  One wave per SIMD as before; the accumulator GLWE sits in the wave's LDS slice (the LDS image of the trace
// kernels), the eight selector GGSWs of the byte stream from L2 through a twelve-quad register ring filled one and a half key
// polynomials ahead, transforms three at a time with their exchange stores spread through the butterflies.  The wrapper
// loads the keyed table, and extracts the four LWEs (coefficients 0, 256, 512, 768: bits 4 acc + i of the table entry) from
// LDS afterwards.  Unit order as k_lut_eval: XCD-aware, the 2 nlut units of a byte on one XCD.  No barrier inside the
// loop (every wave streams its own selector): a wave past the end leaves.  Bit-identical to k_lut_eval and the oracle.
// NOT the default (TX_LUT_VARIANT=4 selects it): measured 2.30 ms against 2.09 ms for the hipcc kernel on the medium
// instance.  In-kernel stamps, cycles per CMux step: difference + digits 18.0 k, 21 forward transforms 41.1 k, the 63 key
// polynomials 55.7 k (880 cycles each: twelve quads in flight are one and a half polynomials, far less than the L2 round
// trip), three inverses 5.9 k, update 3.7 k.  hipcc's schedule spends 480 registers on the same work and keeps more of the
// selector in flight; parking the digit transforms in accumulation registers to widen the ring would bring the key pass to
// ~20 k and the step to ~89 k against hipcc's ~95 k -- 0.3 % of a round -- and was not built (profiles/r4t_*).
// ------------------------------------------------------------------------------------------
#ifdef TXD_EXPERIMENTS   // superseded / reference form: only in libtfhe_mi355x_exp.so (tests, A/B timing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_lut_eval4(const cplx *ggsw_f, const uint64_t *luts, size_t lut_stride, size_t byte_stride, uint64_t *out, int nbytes, int nlut,
                 const cplx *tw_g, const uint8_t *pos)
{
    extern __shared__ __align__(16) unsigned char lds[];
    int wave; unsigned char *slot0;
    unsigned char *tile = tr4_setup(lds, tw_g, wave, slot0);
    const int lane = threadIdx.x & 63;
    const int unit = (int)(((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * WAVES) + wave;
    if (unit >= nbytes * nlut * 2) return;
    const int byte = unit / (nlut * 2), lut = (unit >> 1) % nlut, acc_idx = unit & 1;
    const int table = pos ? (int)pos[byte] : (byte & 15);
    const uint64_t *lp = luts + (size_t)lut * lut_stride + (size_t)table * byte_stride + (size_t)acc_idx * GL;
    uint64_t *gl = reinterpret_cast<uint64_t *>(tile + BR3_TILE_BYTES);
#pragma unroll
    for (int q = 0; q < KP; q++)
#pragma unroll
        for (int m = 0; m < 16; m++) gl[q * N + lane + 64 * m] = lp[q * N + lane + 64 * m];
    const cplx *gg = ggsw_f + (size_t)byte * 8 * CBS_L * KP * KP * H;
    const uint32_t tile_addr = lds_addr(tile);
#ifdef TXD_STAMPS_LUT
    unsigned long long *dbg = (blockIdx.x == TXD_STAMP_BLOCK && wave == TXD_STAMP_THREAD / 64) ? txd_stamps : txd_stamps_sink;
    if (lane == 0 && dbg == txd_stamps) txd_stamps[6] = 8ull;
    asm volatile(
#include "tx_lut_asm_stamps.inc"
        :
        : "s"(gg), "s"(tile_addr), "s"(dbg)
        :
#include "tx_trace_asm_clobbers.inc"
    );
#else
    asm volatile(
#include "tx_lut_asm.inc"
        :
        : "s"(gg), "s"(tile_addr)
        :
#include "tx_trace_asm_clobbers.inc"
    );
#endif
    // the four LWEs of this accumulator: lwe[q N + j] = a_q[nth - j] (j <= nth), -a_q[N + nth - j] otherwise; body = b[nth]
    const int lane2 = threadIdx.x & 63;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        const int nth = 256 * i;
        uint64_t *o = out + (((size_t)lut * nbytes + byte) * 8 + acc_idx * 4 + i) * (size_t)BIG;
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const int j = lane2 + 64 * m;
                int src = nth - j;
                uint64_t neg = 0;
                if (src < 0) { src += N; neg = ~neg; }
                const uint64_t x = gl[q * N + src];
                o[q * N + j] = (x ^ neg) - neg;
            }
        if (lane2 == 0) o[2 * N] = gl[2 * N + nth];
    }
}
#endif  // TXD_EXPERIMENTS

// a6 + a7 + a8 + a9 in one kernel: the traced GLev element stays in the wave's LDS slice and goes straight into the
// scheme switch -- no GLev round trip through HBM (24 KiB out and in per unit) and one workgroup start (twiddle image,
// first loads) instead of two.  What run_cbs launches; the stage entry points keep the two kernels apart.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_trace_scheme_switch4(const uint64_t *acc, cplx *ggsw_f, int count, const cplx *auto_f, size_t split_stride,
                            const cplx *ss_f, const cplx *tw_g, int tie_on)
{
    extern __shared__ __align__(16) unsigned char lds[];
    int wave; unsigned char *slot0;
    unsigned char *tile = tr4_setup(lds, tw_g, wave, slot0);
    const int lane = threadIdx.x & 63;
    uint64_t *gl = reinterpret_cast<uint64_t *>(tile + BR3_TILE_BYTES);
    const uint32_t slot0_addr = lds_addr(slot0), tile_addr = lds_addr(tile);
    const uint32_t split_bytes = (uint32_t)(split_stride * sizeof(cplx));
    // persistent workgroups: the twiddle image is built once; every wave of the workgroup makes the same number of
    // passes (the ring's barriers), a wave past the end redoes the last unit and stores what its owner stores
    const int total = count * CBS_L, per_pass = (int)gridDim.x * WAVES;
#pragma unroll 1
    for (int base = 0; base < total; base += per_pass) {
        int unit = base + (int)blockIdx.x * WAVES + wave;
        if (unit >= total) unit = total - 1;
        tr4_extract_to_lds(acc + (size_t)(unit / CBS_L) * GL, unit % CBS_L, gl, lane);
        tr4_trace_asm(auto_f, split_bytes, wave, slot0_addr, tile_addr, tie_on);
        tr4_ss_asm(ss_f, ggsw_f + (size_t)unit * KP * KP * H, wave, slot0_addr, tile_addr, tie_on);
    }
}

}  // namespace txd
