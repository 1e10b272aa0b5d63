// VALU issue-cost probe: cycles per wave-instruction per SIMD for the ops of the key tracker.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return; } } while (0)
template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned *out, int iters, unsigned m) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u;
    unsigned x = threadIdx.x, y = threadIdx.x * 3 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_and_or_b32 %0, %1, %2, 17" : "=v"(a[i]) : "v"(a[i]), "s"(m));
                if (OP == 1) asm volatile("v_max3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(x), "v"(y));
                if (OP == 2) asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(x), "v"(y));
                if (OP == 3) asm volatile("v_max_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == 5) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 6) asm volatile("v_and_or_b32 %0, %1, %2, 17" : "=v"(a[i]) : "v"(a[i]), "v"(y));
                if (OP == 7) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(x), "v"(y));
                if (OP == 8) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(x), "v"(y));
                if (OP == 9) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 10) asm volatile("v_bfi_b32 %0, %1, %2, 17" : "=v"(a[i]) : "s"(m), "v"(a[i]));
                if (OP == 11) asm volatile("v_and_b32 %0, 0x7fffffc0, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 12) asm volatile("v_or_b32 %0, 17, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 13) asm volatile("v_lshl_or_b32 %0, %1, 6, 17" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 14) asm volatile("v_add3_u32 %0, %1, %2, 17" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 15) asm volatile("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 16) asm volatile("v_max_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 17) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(x));
                if (OP == 18) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(x), "v"(y));
                if (OP == 19) asm volatile("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(a[i]) : "v"(a[i]), "v"(x), "v"(y));
                if (OP == 20) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(unsigned long long*)&a[i&6]) : "v"(*(unsigned long long*)&a[i&6]), "v"(*(unsigned long long*)&a[(i+2)&6]));
            }
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int bpc) {
    unsigned *out;
    int blocks = 256 * bpc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<OP><<<blocks, 256>>>(out, 100, 0x7fffffc0u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    probe<OP><<<blocks, 256>>>(out, iters, 0x7fffffc0u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double insts_per_simd = (double)iters * 32 * bpc;   // waves per SIMD = bpc
    printf("%-18s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, bpc, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    CK(hipFree(out));
}
int main() {
    for (int bpc = 2; bpc <= 4; bpc += 2) {
        run<0>("v_and_or_b32 sgpr", bpc); run<6>("v_and_or_b32 vgpr", bpc); run<1>("v_max3_u32", bpc); run<2>("v_med3_u32", bpc); run<3>("v_max_u32", bpc);
        run<4>("v_fma_f32", bpc); run<5>("v_add_u32", bpc);
        run<7>("v_max3_f32", bpc); run<8>("v_med3_f32", bpc); run<9>("v_max_f32", bpc); run<10>("v_bfi_b32", bpc);
        run<11>("v_and_b32 lit", bpc); run<12>("v_or_b32", bpc); run<13>("v_lshl_or_b32", bpc); run<14>("v_add3_u32", bpc);
        run<15>("v_lshl_add_u32", bpc); run<16>("v_max_i32", bpc); run<17>("v_pk_max_u16", bpc); run<18>("v_perm_b32", bpc);
        run<19>("v_max3_f32 abs", bpc); run<20>("v_pk_add_f32", bpc);
    }
    return 0;
}
