// Facts the f16 two-MFMA prefilter of hsq_encode_pf.hip relies on, checked on the device (MI355X, gfx950):
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/f16_probe tools/f16_probe.hip && tools/exp/f16_probe
//  1. v_mfma_f32_32x32x16_f16 keeps f16 SUBNORMAL inputs (does not flush them to zero)
//  2. v_fma_mixlo_f16 / v_fma_mixhi_f16 (x * 2^s + 0, f32 sources) == round-to-nearest-even conversion of the exact product
//  3. v_dot2_f32_f16(a, a, acc) against the exact sum of squares: largest relative error
//  4. v_cvt_pk_f16_f32 == RNE
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));

__global__ void mfma_denorm(const _Float16 *a, const _Float16 *b, float *out) {
    const int lane = threadIdx.x;
    half8 A, B;
    for (int i = 0; i < 8; ++i) {   // lane (row j = lane & 31, half h): k = 8h .. 8h+7
        A[i] = a[(lane & 31) * 16 + 8 * (lane >> 5) + i];
        B[i] = b[(lane & 31) * 16 + 8 * (lane >> 5) + i];
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

__global__ void mix_probe(const float *x, float scale, unsigned *packed, float *dot, unsigned *cvtpk, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned r = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(r) : "v"(x[2 * i]), "v"(scale));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r) : "v"(x[2 * i + 1]), "v"(scale));
    packed[i] = r;
    float acc = 0.0f;
    asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(r));
    dot[i] = acc;
    unsigned c;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
    cvtpk[i] = c;
}

int main() {
    // 1. subnormal inputs
    std::vector<_Float16> a(32 * 16, (_Float16)0.0f), b(32 * 16, (_Float16)0.0f);
    a[0] = (_Float16)ldexpf(1.0f, -20);      // row 0, k 0: subnormal f16 (min normal 2^-14)
    b[0] = (_Float16)1024.0f;                // col 0, k 0
    a[16 + 1] = (_Float16)2.0f;              // row 1, k 1
    b[16 + 1] = (_Float16)ldexpf(3.0f, -24); // col 1, k 1: subnormal
    _Float16 *da, *db;
    float *dout;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    mfma_denorm<<<1, 64>>>(da, db, dout);
    std::vector<float> out(64 * 16);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    // D[row][col]: lane = col + 32 * ((row >> 2) & 1), reg = (row & 3) + 4 * (row >> 3)
    auto D = [&](int row, int col) { return out[(col + 32 * ((row >> 2) & 1)) * 16 + (row & 3) + 4 * (row >> 3)]; };
    printf("1. MFMA f16 subnormal inputs: D[0][0] = %g (kept: %g), D[1][1] = %g (kept: %g)\n", D(0, 0), ldexpf(1.0f, -10), D(1, 1),
           ldexpf(6.0f, -24));
    // 2..4
    const int n = 1 << 22;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand();
        float m = 1.0f + (bits & 0x7FFFFF) * ldexpf(1.0f, -23);
        int e = (int)((bits >> 23) % 40) - 30;      // 2^-30 .. 2^9 after the scale below: subnormal, normal f16 results
        x[i] = ((bits >> 31) ? -1.0f : 1.0f) * ldexpf(m, e);
        if (i % 1000 == 0) x[i] = ldexpf((float)(2 * (i % 2048) + 1), -12);   // exact ties of the f16 grid in [1, 2)
    }
    float *dx, *ddot; unsigned *dp, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ddot, n * 2); hipMalloc(&dp, n * 2); hipMalloc(&dc, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    const float scale = 4.0f;
    mix_probe<<<n / 2 / 256, 256>>>(dx, scale, dp, ddot, dc, n);
    std::vector<unsigned> p(n / 2), c(n / 2); std::vector<float> dot(n / 2);
    hipMemcpy(p.data(), dp, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(dot.data(), ddot, n * 2, hipMemcpyDeviceToHost);
    long bad_mix = 0, bad_cvt = 0; double worst = 0;
    for (int i = 0; i < n / 2; ++i) {
        _Float16 lo = (_Float16)(x[2 * i] * scale), hi = (_Float16)(x[2 * i + 1] * scale);   // host: RNE
        unsigned short ulo, uhi; memcpy(&ulo, &lo, 2); memcpy(&uhi, &hi, 2);
        if (p[i] != ((unsigned)uhi << 16 | ulo)) ++bad_mix;
        _Float16 l2 = (_Float16)x[2 * i], h2 = (_Float16)x[2 * i + 1];
        memcpy(&ulo, &l2, 2); memcpy(&uhi, &h2, 2);
        if (c[i] != ((unsigned)uhi << 16 | ulo)) ++bad_cvt;
        const double ex = (double)(float)lo * (double)(float)lo + (double)(float)hi * (double)(float)hi;
        if (ex > 0) { const double r = fabs(dot[i] - ex) / ex; if (r > worst) worst = r; }
    }
    printf("2. v_fma_mixlo/hi_f16(x * 4): %ld of %d pairs differ from the RNE conversion of the product\n", bad_mix, n / 2);
    printf("3. v_dot2_f32_f16(a, a, 0): largest relative error %.3g (2^%.1f)\n", worst, log2(worst > 0 ? worst : 1e-30));
    printf("4. v_cvt_pk_f16_f32: %ld of %d pairs differ from RNE\n", bad_cvt, n / 2);
    return 0;
}
