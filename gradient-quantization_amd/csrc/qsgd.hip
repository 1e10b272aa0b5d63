// QSGD bucketed stochastic quantiser for gfx950 (BASELINE config 5).
//
// Replaces qsgd_compressor.py:42-71 of the reference.  Pure HBM-bound byte work:
// 4 B read per element, 2 B (sign + level) written; one wave owns one bucket, the
// max-|v| reduction is a wave shuffle reduction, the second sweep over the bucket
// hits L1/L2.  -ffp-contract=off; IEEE divide.
#include <limits.h>

#include "gq_common.hpp"

namespace gq {

constexpr int QS_THREADS = 256;

template <typename LevelT>
__device__ __forceinline__ LevelT nan_level();
template <>
__device__ __forceinline__ int32_t nan_level<int32_t>() {
    return INT_MIN;  // the reference's x86 float->int32 cast of NaN
}
template <>
__device__ __forceinline__ uint8_t nan_level<uint8_t>() {
    return 0;  // decodes to 0 like INT_MIN does (norm of that bucket is 0)
}

template <typename LevelT>
__device__ __forceinline__ void qsgd_quantise_one(float v, float norm, float s, float smax, int random_mode,
                                                  const float *__restrict__ r, uint64_t seed, int64_t i,
                                                  uint8_t *__restrict__ signs, LevelT *__restrict__ levels) {
    const float q = v / norm;
    const float x = fabsf(q) * s;
    LevelT out;
    if (x != x) {
        out = nan_level<LevelT>();
    } else {
        const float c = fminf(fmaxf(x, 0.0f), smax);
        int l = (int)c;
        if (random_mode != GQ_RANDOM_OFF) {
            const float prob = x - (float)l;
            const float rr = (random_mode == GQ_RANDOM_GIVEN) ? r[i] : uniform01(seed, (uint64_t)i);
            l += (prob > rr) ? 1 : 0;
        }
        out = (LevelT)l;
    }
    levels[i] = out;
    signs[i] = v > 0.0f ? 1 : 0;
}

// One wave per bucket (d <= a few thousand).
template <typename LevelT>
__global__ __launch_bounds__(QS_THREADS) void qsgd_compress_wave_kernel(const float *__restrict__ grad, int64_t Mb,
                                                                       int d, int n_bit, int random_mode,
                                                                       const float *__restrict__ r, uint64_t seed,
                                                                       float *__restrict__ norm,
                                                                       uint8_t *__restrict__ signs,
                                                                       LevelT *__restrict__ levels) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (QS_THREADS / 64);
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    for (int64_t b = (int64_t)blockIdx.x * (QS_THREADS / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); b < Mb; b += nw) {   // (the wave's index: uniform to the compiler too, the bucket's address arithmetic stays scalar)
        const float *v = grad + b * (int64_t)d;
        float mx = 0.0f;
        for (int jj = lane; jj < d; jj += 64) mx = absmax3_nan(mx, v[jj], v[jj]);   // NaN-propagating, like torch.max
        mx = wave_max_nan(mx);
        if (lane == 0) norm[b] = mx;
        for (int jj = lane; jj < d; jj += 64)
            qsgd_quantise_one<LevelT>(v[jj], mx, s, smax, random_mode, r, seed, b * (int64_t)d + jj, signs, levels);
    }
}

// Large buckets (c_dim = 0 -> one bucket spanning the tensor): abs-max by atomicMax on
// the bit pattern (|v| >= 0, so unsigned order == float order), then an elementwise pass.
// A wave folds its 64 values first when they share a bucket and looks at the word before the atomic (it only
// grows): one atomic per element queued 2.4 M of them on one address (~90 per microsecond).
__global__ __launch_bounds__(QS_THREADS) void qsgd_absmax_kernel(const float *__restrict__ grad, int64_t Mb, int d,
                                                                unsigned *__restrict__ norm_bits) {
    const int64_t total = Mb * (int64_t)d;
    const int64_t stride = (int64_t)gridDim.x * QS_THREADS;
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * QS_THREADS + (threadIdx.x & ~63); i0 < total; i0 += stride) {
        const int64_t i = i0 + lane;
        const bool in = i < total;
        const float a = in ? fabsf(grad[i]) : 0.0f;
        const int64_t last = (i0 + 63 < total ? i0 + 63 : total - 1);
        const int64_t b0 = i0 / d;
        if (b0 == last / d) {                       // the wave's elements share a bucket (wave-uniform test)
            const unsigned m = __float_as_uint(wave_max(a));
            if (lane == 0 && m > __hip_atomic_load(&norm_bits[b0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&norm_bits[b0], m);
        } else if (in) {
            atomicMax(&norm_bits[i / d], __float_as_uint(a));
        }
    }
}

template <typename LevelT>
__global__ __launch_bounds__(QS_THREADS) void qsgd_quantise_kernel(const float *__restrict__ grad, int64_t Mb, int d,
                                                                  int n_bit, int random_mode,
                                                                  const float *__restrict__ r, uint64_t seed,
                                                                  const float *__restrict__ norm,
                                                                  uint8_t *__restrict__ signs,
                                                                  LevelT *__restrict__ levels) {
    const int64_t total = Mb * (int64_t)d;
    const int64_t stride = (int64_t)gridDim.x * QS_THREADS;
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * QS_THREADS + threadIdx.x; i < total; i += stride)
        qsgd_quantise_one<LevelT>(grad[i], norm[i / d], s, smax, random_mode, r, seed, i, signs, levels);
}

template <typename LevelT>
__global__ __launch_bounds__(QS_THREADS) void qsgd_decode_sum_kernel(const float *__restrict__ norm,
                                                                    const uint8_t *__restrict__ signs,
                                                                    const LevelT *__restrict__ levels, int R,
                                                                    int64_t Mb, int d, int n_bit,
                                                                    float *__restrict__ out) {
    const int64_t total = Mb * (int64_t)d;
    const int64_t stride = (int64_t)gridDim.x * QS_THREADS;
    const float s = (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R);
    for (int64_t i = (int64_t)blockIdx.x * QS_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t b = i / d;
        float acc = 0.0f;
        for (int r = 0; r < R; ++r) {
            // qsgd_compressor.py:69-70: (l * (2*signs - 1)) * norm / s
            const float sg = 2.0f * (float)signs[(int64_t)r * total + i] - 1.0f;
            float t = (float)levels[(int64_t)r * total + i] * sg;
            t = t * norm[(int64_t)r * Mb + b];
            t = t / s;
            acc = (r == 0) ? t : acc + t;
        }
        if (md.apply) acc = mean_div(acc, md);
        out[i] = acc;
    }
}

static inline int64_t grid_cap(int64_t blocks) {
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

template <typename LevelT>
static int launch_qsgd_compress(const float *grad, int64_t Mb, int d, int n_bit, int random_mode, const float *r,
                                uint64_t seed, float *norm, uint8_t *signs, LevelT *levels, hipStream_t st) {
    if (d <= 8192) {
        const int64_t blocks = grid_cap((Mb + (QS_THREADS / 64) - 1) / (QS_THREADS / 64));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_compress_wave_kernel<LevelT>), dim3((unsigned)blocks),
                           dim3(QS_THREADS), 0, st, grad, Mb, d, n_bit, random_mode, r, seed, norm, signs, levels);
    } else {
        const int64_t total = Mb * (int64_t)d;
        const int64_t blocks = grid_cap((total + QS_THREADS - 1) / QS_THREADS);
        hipError_t e = hipMemsetAsync(norm, 0, (size_t)Mb * sizeof(float), st);
        if (e != hipSuccess) return fail(GQ_ERR_HIP, "gq_qsgd_compress: memset: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(qsgd_absmax_kernel, dim3((unsigned)blocks), dim3(QS_THREADS), 0, st, grad, Mb, d,
                           reinterpret_cast<unsigned *>(norm));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_quantise_kernel<LevelT>), dim3((unsigned)blocks), dim3(QS_THREADS), 0,
                           st, grad, Mb, d, n_bit, random_mode, r, seed, norm, signs, levels);
    }
    GQ_CHECK_LAUNCH("gq_qsgd_compress");
    return GQ_OK;
}

}  // namespace gq

GQ_API int gq_qsgd_compress(const float *grad, int64_t Mb, int d, int n_bit, int random_mode, const float *r,
                            uint64_t seed, float *norm, uint8_t *signs, void *levels, int level_bytes, void *stream) {
    if (Mb < 1 || d < 1 || n_bit < 1 || n_bit > 30)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress: bad sizes Mb=%lld d=%d n_bit=%d", (long long)Mb, d, n_bit);
    if (!grad || !norm || !signs || !levels) return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress: null pointer");
    if (random_mode < GQ_RANDOM_OFF || random_mode > GQ_RANDOM_DEVICE || (random_mode == GQ_RANDOM_GIVEN && !r))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress: random_mode / r");
    hipStream_t st = gq::as_stream(stream);
    if (level_bytes == 1) {
        if (((int64_t)1 << n_bit) > 255) return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress: uint8 levels need n_bit <= 7");
        return gq::launch_qsgd_compress<uint8_t>(grad, Mb, d, n_bit, random_mode, r, seed, norm, signs,
                                                 static_cast<uint8_t *>(levels), st);
    }
    if (level_bytes == 4)
        return gq::launch_qsgd_compress<int32_t>(grad, Mb, d, n_bit, random_mode, r, seed, norm, signs,
                                                 static_cast<int32_t *>(levels), st);
    return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress: level_bytes must be 1 or 4");
}

GQ_API int gq_qsgd_decode_sum(const float *norm, const uint8_t *signs, const void *levels, int level_bytes, int R,
                              int64_t Mb, int d, int n_bit, float *out, void *stream) {
    if (Mb < 1 || d < 1 || R < 1 || n_bit < 1 || n_bit > 30)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum: bad sizes");
    if (!norm || !signs || !levels || !out) return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum: null pointer");
    hipStream_t st = gq::as_stream(stream);
    const int64_t blocks = gq::grid_cap((Mb * (int64_t)d + gq::QS_THREADS - 1) / gq::QS_THREADS);
    if (level_bytes == 1)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gq::qsgd_decode_sum_kernel<uint8_t>), dim3((unsigned)blocks),
                           dim3(gq::QS_THREADS), 0, st, norm, signs, static_cast<const uint8_t *>(levels), R, Mb, d,
                           n_bit, out);
    else if (level_bytes == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gq::qsgd_decode_sum_kernel<int32_t>), dim3((unsigned)blocks),
                           dim3(gq::QS_THREADS), 0, st, norm, signs, static_cast<const int32_t *>(levels), R, Mb, d,
                           n_bit, out);
    else
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum: level_bytes must be 1 or 4");
    GQ_CHECK_LAUNCH("gq_qsgd_decode_sum");
    return GQ_OK;
}
