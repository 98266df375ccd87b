// exploratory microbenchmark (round 4): issue cost of a few vector instructions for a lone wave and for two / three waves per SIMD:
// v_lshl_add_u64 against the v_add_co_u32 + v_addc_co_u32 pair it could replace in the conversions, v_permlane32_swap, DPP v_cndmask_b32.
//   hipcc --offload-arch=gfx950 -O3 -o build_exp/insn_rate tools/microbench/insn_rate.hip && build_exp/insn_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ void k_rate(unsigned long long *out, int iters)
{
    unsigned long long a = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, b = a * 3, c = a * 5, d = a * 7;
    unsigned int lo0 = (unsigned int)a, hi0 = (unsigned int)(a >> 32), lo1 = (unsigned int)b, hi1 = (unsigned int)(b >> 32);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {        // four independent 64-bit adds as v_lshl_add_u64
            asm volatile(REP64("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0x123456789ull + threadIdx.x));
        } else if (KIND == 1) { // two independent 64-bit adds as add_co / addc pairs (same instruction count as KIND 0 per REP)
            asm volatile(REP64("v_add_co_u32 %0, s[20:21], %0, %4\n v_addc_co_u32 %1, s[20:21], %1, %5, s[20:21]\n v_add_co_u32 %2, s[22:23], %2, %4\n v_addc_co_u32 %3, s[22:23], %3, %5, s[22:23]\n")
                         : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1) : "v"(threadIdx.x * 77u + 1u), "v"(threadIdx.x + 5u) : "s20", "s21", "s22", "s23");
        } else if (KIND == 2) { // v_add_f64 (reference: 4 cycles)
            double x = __longlong_as_double(a), y = __longlong_as_double(b), z = __longlong_as_double(c), w = __longlong_as_double(d);
            asm volatile(REP64("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(1.0));
            a = __double_as_longlong(x); b = __double_as_longlong(y); c = __double_as_longlong(z); d = __double_as_longlong(w);
        } else if (KIND == 3) { // v_permlane32_swap
            asm volatile(REP64("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n")
                         : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
        } else if (KIND == 5) { // v_lshlrev_b64 (variable shift)
            asm volatile(REP64("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x & 1u));
        } else if (KIND == 6) { // v_and_or_b32, v_xad_u32, v_max_u32, v_bfe_u32 (one each)
            asm volatile(REP64("v_and_or_b32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %4, %2\n v_max_u32 %2, %2, %3\n v_bfe_u32 %3, %0, 3, 11\n")
                         : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1) : "v"(threadIdx.x * 77u + 1u));
        } else if (KIND == 7) { // v_fma_f64 x 2 + v_add_u32 x 2: do 32-bit operations hide behind FP64 ones?
            double x = __longlong_as_double(a), y = __longlong_as_double(b);
            asm volatile(REP64("v_fma_f64 %0, %0, %4, %0\n v_add_u32 %2, %2, %5\n v_fma_f64 %1, %1, %4, %1\n v_add_u32 %3, %3, %5\n")
                         : "+v"(x), "+v"(y), "+v"(lo0), "+v"(hi0) : "v"(1.0), "v"(threadIdx.x * 77u + 1u));
            a = __double_as_longlong(x); b = __double_as_longlong(y);
        } else if (KIND == 8) { // v_cvt_f64_i32 / v_rndne_f64 / v_mul_f64 / v_fma_f64 (one each)
            double x = __longlong_as_double(a), y = __longlong_as_double(b), z = __longlong_as_double(c);
            asm volatile(REP64("v_cvt_f64_i32 %0, %3\n v_rndne_f64 %1, %1\n v_mul_f64 %2, %2, %4\n v_fma_f64 %1, %1, %4, %0\n")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(lo0) : "v"(1.0));
            a = __double_as_longlong(x); b = __double_as_longlong(y); c = __double_as_longlong(z);
        } else if (KIND == 4) { // v_add_u32 (32-bit reference)
            asm volatile(REP64("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1) : "v"(threadIdx.x * 77u + 1u));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a + b + c + d + lo0 + hi0 + lo1 + hi1 == 12345) out[0] = 1;
}
template <int KIND> void run(const char *name, int threads)
{
    unsigned long long *d; (void)hipMalloc(&d, 8192 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[8192]; (void)hipMemcpy(h, d, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; double sum = 0; const int nw = 256 * (threads / 64);
    for (int i = 0; i < nw; i++) { mx = h[i] > mx ? h[i] : mx; sum += (double)h[i]; }
    const double instr = iters * 64.0 * 4;
    printf("%-24s %d waves per SIMD: per wave %.2f (mean) / %.2f (max) cycles per instruction; launch %.3f ms = %.2f ns per instruction and SIMD\n", name, threads / 256,
           sum / nw / instr, (double)mx / instr, ms, ms * 1e6 / (instr * (threads / 256)));
    (void)hipFree(d);
}
int main()
{
    for (int threads : {256, 512, 768}) {
        if (threads == 256) { run<5>("v_lshlrev_b64", 256); run<6>("and_or/xad/max/bfe", 256); run<7>("fma_f64 + add_u32 mix", 256); run<8>("cvt/rndne/mul/fma f64", 256); }
        if (threads == 512) { run<5>("v_lshlrev_b64", 512); run<6>("and_or/xad/max/bfe", 512); run<7>("fma_f64 + add_u32 mix", 512); run<8>("cvt/rndne/mul/fma f64", 512); }
        if (threads == 768) { run<5>("v_lshlrev_b64", 768); run<6>("and_or/xad/max/bfe", 768); run<7>("fma_f64 + add_u32 mix", 768); run<8>("cvt/rndne/mul/fma f64", 768); }
    }
    for (int threads : {256, 512, 768, 1024}) {
        if (threads == 256) { run<2>("v_add_f64", 256); run<4>("v_add_u32", 256); run<0>("v_lshl_add_u64", 256); run<1>("v_add_co + v_addc_co", 256); run<3>("v_permlane32_swap_b32", 256); }
        if (threads == 512) { run<2>("v_add_f64", 512); run<4>("v_add_u32", 512); run<0>("v_lshl_add_u64", 512); run<1>("v_add_co + v_addc_co", 512); run<3>("v_permlane32_swap_b32", 512); }
        if (threads == 1024) { run<2>("v_add_f64", 1024); run<4>("v_add_u32", 1024); run<0>("v_lshl_add_u64", 1024); run<1>("v_add_co + v_addc_co", 1024); run<3>("v_permlane32_swap_b32", 1024); }
        if (threads == 768) { run<2>("v_add_f64", 768); run<4>("v_add_u32", 768); run<0>("v_lshl_add_u64", 768); run<1>("v_add_co + v_addc_co", 768); run<3>("v_permlane32_swap_b32", 768); }
    }
    return 0;
}
