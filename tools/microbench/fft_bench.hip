// exploratory microbenchmark: the 3-polynomial interleaved transforms alone (one wave per SIMD, 4 per CU),
// to study the cost of the LDS tile exchanges outside the blind rotation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I temp-fhe-transciphering_amd/csrc \
//         [-DTXD_ABL_NOXCHG] [-DWAVES_ACTIVE=k] -o build_exp/fft_bench tools/microbench/fft_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tx_kernels.hpp"
using namespace txd;
#ifndef WAVES_ACTIVE
#define WAVES_ACTIVE 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fft_bench(const cplx *tw_g, cplx *out, int iters, unsigned long long *cyc)
{
    extern __shared__ __align__(16) unsigned char lds[];
    WaveCtx w = wave_setup(tw_g, lds);
    if ((threadIdx.x >> 6) >= WAVES_ACTIVE) return;
    cplx v[KP][8];
#pragma unroll
    for (int p = 0; p < KP; p++)
#pragma unroll
        for (int m = 0; m < 8; m++) v[p][m] = make_double2((double)(w.lane + m + p), (double)(w.lane * 3 - m));
#ifdef STAGGER
    for (int k = (int)(threadIdx.x >> 6) * STAGGER; k > 0; k--) __builtin_amdgcn_s_sleep(1);
#endif
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        fft_fwd_n<KP>(v, w.tw, w.tile, w.lane);
#pragma unroll
        for (int p = 0; p < KP; p++)
#pragma unroll
            for (int m = 0; m < 8; m++) v[p][m] = make_double2(v[p][m].x * 0x1p-9, v[p][m].y * 0x1p-9);
        fft_inv_n<KP>(v, w.tw, w.tile, w.lane);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
#pragma unroll
    for (int p = 0; p < KP; p++)
#pragma unroll
        for (int m = 0; m < 8; m++) out[((size_t)w.unit * KP + p) * 512 + m * 64 + w.lane] = v[p][m];
}
int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000, wgs = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<cplx> tw(TW_ROWS * 64);
    for (int r = 0; r < TW_ROWS; r++)
        for (int l = 0; l < 64; l++) { double a = 0.001 * (r * 64 + l); tw[r * 64 + l] = make_double2(cos(a), sin(a)); }
    cplx *d_tw, *d_out; unsigned long long *d_c, c = 0;
    hipMalloc(&d_tw, tw.size() * sizeof(cplx)); hipMalloc(&d_out, (size_t)wgs * 4 * KP * 512 * sizeof(cplx)); hipMalloc(&d_c, 8);
    hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fft_bench, dim3(wgs), dim3(256), LDS_BYTES, 0, d_tw, d_out, iters, d_c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
        printf("waves/CU %d  iters %d: %.3f ms, %.0f cycles per (3 fwd + 3 inv) = %.0f per transform\n", WAVES_ACTIVE, iters, ms,
               (double)c / iters, (double)c / iters / 6);
    }
    return 0;
}
