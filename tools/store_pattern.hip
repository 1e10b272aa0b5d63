// Does the SHAPE of a wave's stores matter for a kernel that writes 100 MB once?  The decode-mean kernels write a subvector's
// 64 bytes by the four lanes of a team: one store instruction of a wave = 16 pieces of 64 B, 256 B apart, and four instructions
// complete the wave's 4 KB.  A fill kernel writes 1 KB contiguous per instruction.
//     hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o tools/exp/store_pattern && tools/exp/store_pattern
//   pattern 0: lane L writes 16 B at 16 L of a 1 KB block, four blocks per item      (contiguous)
//   pattern 1: lane (team t, quarter q) writes 16 B at (4 t + k) 64 + 16 q, k = 0..3  (the decode kernels' d = 16 form)
//   pattern 2: lane L writes 16 B at 32 L and at 32 L + 16                            (the QSGD decode's form: 32 B per lane)
//   pattern 3: lanes 2 j, 2 j + 1 write 32 B contiguous at 64 j, then at 64 j + 32       (pattern 2 after a swap between lane pairs)
// Every pattern: 1024-thread workgroups, one resident wave of workgroups, grid-stride over 4 KB items per wave, values from a
// few VALU operations (nothing is read); 25 M floats per launch, 200 launches on rotating buffers (3 x 100 MB: not cache resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int P> __global__ __launch_bounds__(1024) void fillk(float *out, long nitems, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 16;
    for (long it = (long)blockIdx.x * 16 + wave; it < nitems; it += nw) {
        float *base = out + it * 1024;      // 4 KB per wave and item
        const float v = seed + (float)it;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 x = {v + k, v * 2.0f, v - k, v + lane};
            long off;
            if (P == 0) off = k * 256 + 4 * lane;
            else if (P == 1) off = ((lane >> 2) * 4 + k) * 16 + 4 * (lane & 3);
            else if (P == 2) off = (k >> 1) * 512 + 8 * lane + 4 * (k & 1);
            else off = (k >> 1) * 512 + 16 * (lane >> 1) + 8 * (k & 1) + 4 * (lane & 1);
            *reinterpret_cast<f32x4 *>(base + off) = x;
        }
    }
}
template <int P> static float run(float **bufs, long n, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(fillk<P>, dim3(blocks), dim3(1024), 0, 0, bufs[i % 3], n / 1024, 1.0f);
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(fillk<P>, dim3(blocks), dim3(1024), 0, 0, bufs[i % 3], n / 1024, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 200 * 1e3f;
}
int main() {
    const long n = 25 * 1024 * 1024;
    float *bufs[3];
    for (int i = 0; i < 3; ++i) hipMalloc(&bufs[i], n * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    for (int mult = 1; mult <= 2; ++mult) {
        const int blocks = p.multiProcessorCount * mult;
        const float t0 = run<0>(bufs, n, blocks), t1 = run<1>(bufs, n, blocks), t2 = run<2>(bufs, n, blocks), t3 = run<3>(bufs, n, blocks);
        printf("%d workgroups of 1024: contiguous %.1f us (%.2f TB/s)  team-of-four pieces %.1f us (%.2f TB/s)  32 B per lane %.1f us (%.2f TB/s)  32 B per lane pair %.1f us (%.2f TB/s)\n",
               blocks, t0, n * 4 / t0 * 1e-6, t1, n * 4 / t1 * 1e-6, t2, n * 4 / t2 * 1e-6, t3, n * 4 / t3 * 1e-6);
    }
    return 0;
}
