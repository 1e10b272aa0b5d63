// Probe of the encode inner loop (registers only, no memory): which part limits it?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o enc_probe enc_probe.hip 2>/dev/null
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return; } } while (0)

__device__ __forceinline__ void take(float &bv, int &bi, float v, int idx) {
    const bool gt = fabsf(v) > fabsf(bv);
    bv = gt ? v : bv;
    bi = gt ? idx : bi;
}
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }
__device__ __forceinline__ f32x16 chain(const float (&a)[8], const float (&b)[8]) {
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[ks], acc, 0, 0, 0);
    return acc;
}

// MODE 0: MFMA only (acc kept alive); 1: full argmax (cmp + 2 cndmask); 2: max3 only (value, no index)
// 3: full argmax, no sched_group_barrier; 4: argmax only (no MFMA; acc from registers)
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float *in, float *out, int iters) {
    float a[8][8], b[2][8];
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 8; ++k) a[i][k] = in[(threadIdx.x + 64 * i + 7 * k) & 1023];
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 8; ++k) b[i][k] = in[(threadIdx.x * 3 + 64 * i + 5 * k) & 1023];
    float bv[2] = {0, 0};
    int bi[2] = {0, 0};
    float vm = 0;
    f32x16 fake;
    for (int r = 0; r < 16; ++r) fake[r] = in[(threadIdx.x + r) & 1023];
    for (int it = 0; it < iters; ++it) {
        f32x16 acc = (MODE == 4) ? fake : chain(a[0], b[0]);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            f32x16 nacc;
            if (c + 1 < 16) nacc = (MODE == 4) ? fake : chain(a[(c + 1) & 7], b[(c + 1) >> 3]);
            if (MODE == 0) {
                asm volatile("" ::"v"(acc));
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) vm = fmaxf(fmaxf(fabsf(acc[r]), fabsf(acc[r + 1])), vm);
            } else {
                float lv = acc[0], hv = acc[8];
                int li = acc_row(0), hi = acc_row(8);
#pragma unroll
                for (int r = 1; r < 8; ++r) {
                    take(lv, li, acc[r], acc_row(r));
                    take(hv, hi, acc[r + 8], acc_row(r + 8));
                }
                take(lv, li, hv, hi);
                take(bv[c >> 3], bi[c >> 3], lv, li + (c & 7) * 32);
            }
            if (c + 1 < 16) {
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    }
                }
                if (MODE == 4) fake[c] += bv[c >> 3];
                acc = nacc;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bv[0] + bv[1] + (float)(bi[0] + bi[1]) + vm;
}

template <int MODE>
void run(const char *name, int blocks) {
    float *in, *out;
    CK(hipMalloc(&in, 4096 * 4));
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8) * 1e-7f - 0.8f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    const int iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    probe<MODE><<<blocks, 256>>>(in, out, 20);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    probe<MODE><<<blocks, 256>>>(in, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * 4;
    double flops = 2.0 * 128 * 2048.0 * iters * waves;  // 128 MFMAs per tile-iteration
    printf("%-40s blocks=%4d  %.3f ms  %.1f TFLOP/s-equivalent  (%.0f cycles@2.4GHz per tile per wave)\n", name, blocks,
           ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / iters);
    CK(hipFree(in));
    CK(hipFree(out));
}

int main() {
    for (int bpc = 1; bpc <= 3; ++bpc) {
        int blocks = 256 * bpc;
        run<0>("MFMA only", blocks);
        run<1>("MFMA + argmax (sched groups)", blocks);
        run<3>("MFMA + argmax (compiler order)", blocks);
        run<2>("MFMA + max3 value only", blocks);
        run<4>("argmax only", blocks);
    }
    return 0;
}
