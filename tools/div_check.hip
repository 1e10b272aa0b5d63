// The mean's quotient by an odd number of users (csrc/gq_common.hpp: odd_quotient -- four operations) against the device's
// own IEEE x / R, for EVERY one of the 2^32 float inputs and every odd R in [3, GQ_ODD_DIV_MAX]; also counts what the
// same four operations get wrong for the even R that are not powers of two (why those keep the IEEE sequence).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Igradient-quantization_amd/csrc tools/div_check.hip -o tools/exp/div_check
//   gpurun -- 'tools/exp/div_check'            (a few seconds on one MI355X)
#include "gq_common.hpp"

#include <stdlib.h>

__global__ void check(int R, unsigned long long *bad, unsigned *example) {
    const gq::MeanDiv md = gq::mean_div_of(R);
    unsigned long long mine = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned n = 0; n < (1ull << 32) / stride; ++n, u += stride) {
        const float x = __uint_as_float(u);
        const float q = gq::odd_quotient(x, md.fR, md.inv);
        const float t = x / md.fR;
        const bool same = __float_as_uint(q) == __float_as_uint(t) || (q != q && t != t);
        if (!same && u != 0x80000000u) {   // -0 is checked below: it has to come out as the +0 of (+0) + sum
            ++mine;
            *example = u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && __float_as_uint(gq::odd_quotient(-0.0f, md.fR, md.inv)) != 0u) {
        ++mine;
        *example = 0x80000000u;
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char **argv) {
    const int r_max = argc > 1 ? atoi(argv[1]) : GQ_ODD_DIV_MAX;
    unsigned long long *bad;
    unsigned *example;
    hipMalloc(&bad, 8);
    hipMalloc(&example, 4);
    unsigned long long total_odd = 0, h;
    unsigned ex;
    int n_odd = 0;
    for (int R = 3; R <= r_max; ++R) {
        if ((R & (R - 1)) == 0) continue;
        if (!(R & 1) && R > 64) continue;   // the even ones are only counted, a few are enough
        hipMemset(bad, 0, 8);
        hipMemset(example, 0, 4);
        check<<<4096, 256>>>(R, bad, example);   // 2^20 threads, 4096 inputs each
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&ex, example, 4, hipMemcpyDeviceToHost);
        if (R & 1) {
            total_odd += h;
            ++n_odd;
            if (h) printf("R = %d (odd): %llu inputs differ, e.g. 0x%08x\n", R, h, ex);
        } else {
            printf("R = %d (even): %llu of 2^32 inputs differ, e.g. 0x%08x\n", R, h, ex);
        }
    }
    printf("%d odd R in [3, %d], all 2^32 inputs each: %llu differences from x / R\n", n_odd, r_max, total_odd);
    return total_odd != 0;
}
