// Micro-probe: f32 MFMA issue rates on gfx950 in the shapes the encode kernel uses.
// hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float *in, float *out, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = in[threadIdx.x + 64 * i];
        b[i] = in[threadIdx.x + 64 * i + 512];
    }
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float keep = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // one dependent chain of 8 (32x32x2), reset each chain
            f32x16 acc = {0};
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
            keep += acc[0] + acc[15];
        } else if (MODE == 1) {  // two independent chains interleaved
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[k], a[k], acc1, 0, 0, 0);
            }
        } else if (MODE == 2) {  // 16x16x4, four independent accumulators, 16 MFMAs = same MACs as 8 32x32x2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k + 4], b[k], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k + 4], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k + 4], b[k + 4], c3, 0, 0, 0);
            }
        } else if (MODE == 3) {  // single accumulating chain, never reset
#pragma unroll
            for (int k = 0; k < 8; ++k) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc0, 0, 0, 0);
        }
    }
    float r = keep + acc0[0] + acc1[3] + c0[0] + c1[1] + c2[2] + c3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int blocks, int threads, double macs_per_iter_per_wave) {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMemset(in, 0, 4096 * 4);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8) * 1e-7f - 0.8f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, threads>>>(in, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, threads>>>(in, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * threads / 64;
    double flops = 2.0 * macs_per_iter_per_wave * iters * waves;
    printf("%-44s blocks=%4d thr=%3d  %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, ms, flops / ms / 1e9);
    hipFree(in);
    hipFree(out);
}

int main() {
    const double m8 = 8.0 * 32 * 32 * 2;
    for (int wpc = 1; wpc <= 3; ++wpc) {
        int blocks = 256 * wpc;  // wpc blocks of 256 threads per CU -> wpc waves per SIMD
        run<0>("32x32x2 dependent chain of 8, reset", blocks, 256, m8);
        run<3>("32x32x2 one long dependent chain", blocks, 256, m8);
        run<1>("32x32x2 two independent chains", blocks, 256, 2 * m8);
        run<2>("16x16x4 four independent accumulators", blocks, 256, 16.0 * 16 * 16 * 4);
    }
    return 0;
}
