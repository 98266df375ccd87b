// exploratory microbenchmark (round 2): does a SECOND wave per SIMD pay for transform-shaped work?
// A single wave issues one VALU instruction per 4 cycles whatever its width; two waves on a SIMD
// interleave, so 32-bit instructions cost 2 cycles and one wave's LDS round trips hide behind the
// other's arithmetic.  Work per wave and iteration: 3 forward + 3 inverse transforms (+ INT_OPS rounds
// of 48 32-bit integer instructions standing in for decomposition / conversion work).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I temp-fhe-transciphering_amd/csrc \
//         -DTHREADS=512 -DPOLYS=1 -DINT_OPS=20 -o build_exp/occ_bench tools/microbench/occ_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tx_kernels.hpp"
using namespace txd;
#ifndef THREADS
#define THREADS 512
#endif
#ifndef POLYS
#define POLYS 1
#endif
#ifndef INT_OPS
#define INT_OPS 0
#endif
constexpr int WPE = THREADS / 256;
constexpr int NW = THREADS / 64;
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_occ_bench(const cplx *tw_g, cplx *out, int iters, unsigned long long *cyc)
{
    extern __shared__ __align__(16) unsigned char lds[];
    cplx *tw = reinterpret_cast<cplx *>(lds);
    for (int i = threadIdx.x; i < TW_ROWS * 64; i += blockDim.x) tw[i] = tw_g[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    cplx *tile = reinterpret_cast<cplx *>(lds + TW_ROWS * 64 * sizeof(cplx) + wave * SCRATCH_CPLX * sizeof(cplx));
    cplx v[KP][8];
    uint32_t z[16];
#pragma unroll
    for (int p = 0; p < KP; p++)
#pragma unroll
        for (int m = 0; m < 8; m++) v[p][m] = make_double2((double)(lane + m + p), (double)(lane * 3 - m));
#pragma unroll
    for (int k = 0; k < 16; k++) z[k] = lane * 17 + k;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#if POLYS == 3
        fft_fwd_n<KP>(v, tw, tile, lane);
#else
#pragma unroll
        for (int p = 0; p < KP; p++) fft_fwd(v[p], tw, tile, lane);
#endif
#pragma unroll
        for (int p = 0; p < KP; p++)
#pragma unroll
            for (int m = 0; m < 8; m++) v[p][m] = make_double2(v[p][m].x * 0x1p-9, v[p][m].y * 0x1p-9);
#if POLYS == 3
        fft_inv_n<KP>(v, tw, tile, lane);
#else
#pragma unroll
        for (int p = 0; p < KP; p++) fft_inv(v[p], tw, tile, lane);
#endif
#pragma unroll 1
        for (int q = 0; q < INT_OPS; q++) {
#pragma unroll
            for (int k = 0; k < 16; k++) { z[k] = (z[k] ^ (z[k] >> 7)) + (uint32_t)(q + k); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x == 7) { cyc[2 * wave] = t1 - t0; cyc[2 * wave + 1] = r1 - r0; }
    uint32_t zz = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) zz ^= z[k];
    const int unit = blockIdx.x * NW + wave;
#pragma unroll
    for (int p = 0; p < KP; p++)
#pragma unroll
        for (int m = 0; m < 8; m++) out[((size_t)unit * KP + p) * 512 + m * 64 + lane] = make_double2(v[p][m].x + (double)zz, v[p][m].y);
}
int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000, wgs = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<cplx> tw(TW_ROWS * 64);
    for (int r = 0; r < TW_ROWS; r++)
        for (int l = 0; l < 64; l++) { double a = 0.001 * (r * 64 + l); tw[r * 64 + l] = make_double2(cos(a), sin(a)); }
    cplx *d_tw, *d_out; unsigned long long *d_c, c[16] = {0};
    hipMalloc(&d_tw, tw.size() * sizeof(cplx)); hipMalloc(&d_out, (size_t)wgs * NW * KP * 512 * sizeof(cplx)); hipMalloc(&d_c, 128);
    hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)(TW_ROWS * 64 + NW * SCRATCH_CPLX) * sizeof(cplx);
    hipFuncSetAttribute((const void *)k_occ_bench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_occ_bench, dim3(wgs), dim3(THREADS), ldsb, 0, d_tw, d_out, iters, d_c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(c, d_c, 128, hipMemcpyDeviceToHost);
        printf("threads %d polys-interleaved %d int_ops %d iters %d: %.3f ms = %.2f us per iteration;", THREADS, POLYS, INT_OPS, iters, ms, ms * 1e3 / iters);
        for (int k = 0; k < NW; k++) printf(" w%d %.0f cyc @%.2f GHz", k, (double)c[2 * k] / iters, (double)c[2 * k] / (double)c[2 * k + 1] * 0.1);
        printf("\n");
    }
    return 0;
}
