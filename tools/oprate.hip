// Issue cost of single VALU operations on gfx950, per wave-instruction, at 1 / 2 / 4 waves per SIMD:
//     hipcc --offload-arch=gfx950 -O3 tools/oprate.hip -o tools/exp/oprate && tools/exp/oprate [waves per SIMD = 4]
// Every wave runs ITER x 8 operations of one kind on 8 independent registers.  (The v_cndmask line measures its
// vcc hazard in this form, not the operation.)  Results: profiles/r02_b_valu_op_rates.txt, DESIGN.md section 7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
template <int OP> __global__ __launch_bounds__(256) void k(float *out, float seedf) {
    float f[8]; double d[8]; int c = 0;
    for (int i = 0; i < 8; ++i) { f[i] = seedf + threadIdx.x * 1e-3f + i; d[i] = (double)f[i]; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[u]));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[u]));
            if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[u]) : "v"(f[u]));
            if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[u]) : "v"(d[u]));
            if (OP == 4) asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(d[u]), "v"(d[(u + 1) & 7]) : "vcc");
            if (OP == 5) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[u]));
            if (OP == 6) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[u]));
            if (OP == 7) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(f[u]) : : "a0");
            if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[u]));
            if (OP == 9) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(f[u]), "v"(f[(u + 1) & 7]) : "vcc");
            if (OP == 10) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(c) : "v"(u), "v"(c), "v"(c): "vcc");
            if (OP == 11) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(d[u]));
            if (OP == 12) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(f[u]), "+v"(f[(u+1)&7]));
            if (OP == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 14) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 16) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 17) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[u]) : "v"(f[(u+1)&7]) : "vcc");
            if (OP == 19) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 21) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 22) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 23) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(f[u]) : "s20");
            if (OP == 24) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 25) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 26) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 27) asm volatile("v_max_u32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 28) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 29) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[u]));
            if (OP == 30) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(f[u]), "v"(f[(u+1)&7]) : "vcc");
            if (OP == 31) asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(f[u]), "v"(f[(u+1)&7]) : "s20", "s21");
            if (OP == 32) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(f[u]));
            if (OP == 33) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 34) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 35) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "s"(seedf), "v"(f[(u+2)&7]));
            if (OP == 36) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 37) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(f[u]));
            if (OP == 38) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 39) asm volatile("v_max_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 40) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 41) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 42) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 43) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 44) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
            if (OP == 45) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(f[u]));
            if (OP == 46) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(f[(u+1)&7]), "v"(f[(u+2)&7]));
            if (OP == 47) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u+1)&7]));
        }
    }
    float s = c;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static int g_wps = 4;
template <int OP> void run(const char *name, int per) {
    float *o; hipMalloc(&o, 256 * 4 * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * g_wps;   // g_wps blocks of 4 waves per CU: g_wps waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 1.0f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, o, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: 4 waves x ITER x 8 x per ops
    const double ops = (double)g_wps * ITER * 8 * per;
    printf("%-22s %.2f ns per wave-op  (%.1f cycles at 2.4 GHz)\n", name, ms * 1e6 / ops, ms * 1e6 / ops * 2.4);
    hipFree(o);
}
int main(int argc, char **argv) {
    if (argc > 1) g_wps = atoi(argv[1]);
    printf("%d waves per SIMD\n", g_wps);
    run<0>("v_fma_f32", 1); run<1>("v_add_f64", 1); run<2>("v_cvt_f64_f32", 1); run<3>("v_cvt_f32_f64", 1);
    run<4>("v_cmp_ge_f64", 1); run<5>("v_fma_f64", 1); run<6>("v_mul_f64", 1); run<7>("accvgpr write+read", 2);
    run<8>("v_rcp_f32", 1); run<9>("v_cmp_ge_f32", 1); run<10>("add_co+addc", 2); run<11>("v_lshlrev_b64", 1);
    run<12>("v_permlane32_swap", 1);
    run<13>("v_max_f32", 1); run<14>("v_max3_f32", 1); run<15>("v_and_or_b32", 1); run<16>("v_max3_u32", 1);
    run<17>("v_med3_u32", 1); run<18>("v_cndmask_b32 vcc", 1); run<19>("v_and_b32", 1); run<20>("v_mov_b32", 1);
    run<21>("v_cvt_pk_bf16_f32", 1); run<22>("v_sub_f32", 1); run<23>("v_readlane_b32", 1); run<24>("v_fmac_f32", 1);
    run<25>("v_mul_f32", 1); run<26>("v_add_u32", 1); run<27>("v_max_u32", 1); run<28>("v_max_f32 |a|,|b|", 1);
    run<29>("v_pk_fma_f32", 1); run<30>("v_cmp_gt_u32 vcc", 1); run<31>("v_cmp_gt_u32 sgpr", 1); run<32>("v_lshlrev_b32", 1);
    run<33>("v_perm_b32", 1); run<34>("v_maximum3_f32", 1); run<35>("v_fma_f32 sgpr src", 1); run<36>("v_min_f32", 1);
    run<37>("v_bfe_u32", 1); run<38>("v_mov_b32_dpp", 1); run<39>("v_max_f32_dpp", 1);
    run<40>("v_mul_lo_u32", 1); run<41>("v_mul_u32_u24", 1); run<42>("v_mad_u32_u24", 1); run<43>("v_mul_hi_u32", 1);
    run<44>("v_xor_b32", 1); run<45>("v_alignbit_b32", 1); run<46>("v_xad_u32", 1); run<47>("v_mul_hi_u32_u24", 1);
    return 0;
}
