// HSQ decode + parameter-server aggregate for gfx950.
//
// Replaces, for each of R (codes, levels, lb, ub) payloads,
// probabilistic_scalar_compressor.py:29-33 (level -> norm) and
// nearest_neighbor_compressor.py:80-90 (codebook gather * norm), then
// ps_quantizer.py:48 (stack().mean(0)): payloads are summed in ascending rank order
// and divided by R, so the fp32 result equals the reference's for the same payloads.
// Write-bound: 4 B per output element + 2R/d B of payload reads; the codebook is
// staged in LDS once per workgroup.  -ffp-contract=off keeps mul / div / add unfused.
#include "gq_common.hpp"

namespace gq {

constexpr int DEC_THREADS = 256;

template <typename LevelT>
__device__ __forceinline__ float level_to_norm(LevelT l, float lb, float range, float s) {
    float t = (float)l * range;
    t = t / s;
    return t + lb;
}
template <>
__device__ __forceinline__ float level_to_norm<float>(float l, float, float, float) {
    return l;  // n_bit == 32: the payload already carries the f32 projection
}

// d % 4 == 0: one thread produces 4 consecutive floats (one dwordx4 store).
template <typename CodeT, typename LevelT, bool LDS_CB>
__global__ __launch_bounds__(DEC_THREADS) void hsq_decode_sum_v4_kernel(
    const CodeT *__restrict__ codes, const LevelT *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int R, int64_t M,
    int d, int K, int n_bit, float *__restrict__ out) {
    extern __shared__ float s_cb[];
    if (LDS_CB) {
        for (int i = threadIdx.x; i < K * d; i += DEC_THREADS) s_cb[i] = cb[i];
        __syncthreads();
    }
    const int q_per = d >> 2;
    const int64_t total = M * q_per;
    const float s = (float)(1 << (n_bit & 31));
    const float fR = (float)R;
    const int64_t stride = (int64_t)gridDim.x * DEC_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * DEC_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / q_per;
        const int q = (int)(i - m * q_per);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int r = 0; r < R; ++r) {
            const int code = (int)codes[(int64_t)r * code_stride + m];
            const float lb = lb_ub ? lb_ub[r * lbub_stride] : 0.0f;
            const float range = lb_ub ? (lb_ub[r * lbub_stride + 1] - lb) : 0.0f;
            const float n = level_to_norm<LevelT>(levels[(int64_t)r * level_stride + m], lb, range, s);
            const float *row = (LDS_CB ? s_cb : cb) + (int64_t)code * d + 4 * q;
            const f32x4 c = *reinterpret_cast<const f32x4 *>(row);
            f32x4 dec;
            dec[0] = c[0] * n;
            dec[1] = c[1] * n;
            dec[2] = c[2] * n;
            dec[3] = c[3] * n;
            if (r == 0)
                acc = dec;
            else {
                acc[0] = acc[0] + dec[0];
                acc[1] = acc[1] + dec[1];
                acc[2] = acc[2] + dec[2];
                acc[3] = acc[3] + dec[3];
            }
        }
        if (R > 1) {
            acc[0] = acc[0] / fR;
            acc[1] = acc[1] / fR;
            acc[2] = acc[2] / fR;
            acc[3] = acc[3] / fR;
        }
        *reinterpret_cast<f32x4 *>(out + 4 * i) = acc;
    }
}

// any d: one thread per output float.
template <typename CodeT, typename LevelT>
__global__ __launch_bounds__(DEC_THREADS) void hsq_decode_sum_scalar_kernel(
    const CodeT *__restrict__ codes, const LevelT *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int R, int64_t M,
    int d, int n_bit, float *__restrict__ out) {
    const int64_t total = M * d;
    const float s = (float)(1 << (n_bit & 31));
    const float fR = (float)R;
    const int64_t stride = (int64_t)gridDim.x * DEC_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * DEC_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / d;
        const int jj = (int)(i - m * d);
        float acc = 0.0f;
        for (int r = 0; r < R; ++r) {
            const int code = (int)codes[(int64_t)r * code_stride + m];
            const float lb = lb_ub ? lb_ub[r * lbub_stride] : 0.0f;
            const float range = lb_ub ? (lb_ub[r * lbub_stride + 1] - lb) : 0.0f;
            const float n = level_to_norm<LevelT>(levels[(int64_t)r * level_stride + m], lb, range, s);
            const float dec = cb[(int64_t)code * d + jj] * n;
            acc = (r == 0) ? dec : acc + dec;
        }
        if (R > 1) acc = acc / fR;
        out[i] = acc;
    }
}

template <typename CodeT, typename LevelT>
static int launch_decode(const CodeT *codes, const LevelT *levels, const float *lb_ub, int64_t cs, int64_t ls,
                         int64_t bs, const float *cb, int R, int64_t M, int d, int K, int n_bit, float *out,
                         hipStream_t st) {
    const int64_t cap = (int64_t)cu_count() * 8;
    if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(cb) & 15) == 0) {
        const int64_t total = M * (d >> 2);
        int64_t blocks = (total + DEC_THREADS - 1) / DEC_THREADS;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        const size_t lds = (size_t)K * d * sizeof(float);
        // stage the codebook in LDS when it fits and the launch is big enough to amortise it
        if (lds <= 64 * 1024 && total >= (int64_t)K * d) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_v4_kernel<CodeT, LevelT, true>), dim3((unsigned)blocks),
                               dim3(DEC_THREADS), lds, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, d, K, n_bit, out);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_v4_kernel<CodeT, LevelT, false>), dim3((unsigned)blocks),
                               dim3(DEC_THREADS), 0, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, d, K, n_bit, out);
        }
    } else {
        const int64_t total = M * d;
        int64_t blocks = (total + DEC_THREADS - 1) / DEC_THREADS;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_scalar_kernel<CodeT, LevelT>), dim3((unsigned)blocks),
                           dim3(DEC_THREADS), 0, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, d, n_bit, out);
    }
    GQ_CHECK_LAUNCH("gq_hsq_decode_sum");
    return GQ_OK;
}

template <typename CodeT>
static int dispatch_levels(const CodeT *codes, int64_t cs, const void *levels, int level_bytes, int64_t ls,
                           const float *lb_ub, int64_t bs, const float *cb, int R, int64_t M, int d, int K, int n_bit,
                           float *out, hipStream_t st) {
    switch (level_bytes) {
        case 0:
            return launch_decode<CodeT, float>(codes, static_cast<const float *>(levels), nullptr, cs, ls, 0, cb, R, M,
                                               d, K, 0, out, st);
        case 1:
            return launch_decode<CodeT, uint8_t>(codes, static_cast<const uint8_t *>(levels), lb_ub, cs, ls, bs, cb, R,
                                                 M, d, K, n_bit, out, st);
        case 2:
            return launch_decode<CodeT, uint16_t>(codes, static_cast<const uint16_t *>(levels), lb_ub, cs, ls, bs, cb,
                                                  R, M, d, K, n_bit, out, st);
        case 4:
            return launch_decode<CodeT, int32_t>(codes, static_cast<const int32_t *>(levels), lb_ub, cs, ls, bs, cb, R,
                                                 M, d, K, n_bit, out, st);
        default:
            return fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: level_bytes must be 0, 1, 2 or 4");
    }
}

}  // namespace gq

GQ_API int gq_hsq_decode_sum_strided(const void *codes, int code_bytes, int64_t code_stride_bytes, const void *levels,
                                     int level_bytes, int64_t level_stride_bytes, const float *lb_ub,
                                     int64_t lbub_stride_bytes, const float *codebook, int R, int64_t M, int d, int K,
                                     int n_bit, float *out, void *stream) {
    if (M < 1 || d < 1 || K < 1 || R < 1)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: bad sizes R=%d M=%lld d=%d K=%d", R, (long long)M, d, K);
    if (!codes || !levels || !codebook || !out) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: null pointer");
    if (level_bytes != 0 && (!lb_ub || n_bit < 1 || n_bit > 30))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: lb_ub / n_bit required with integer levels");
    const int lsz = level_bytes == 0 ? 4 : level_bytes;
    if ((code_bytes != 1 && code_bytes != 4) || code_stride_bytes % code_bytes || level_stride_bytes % lsz ||
        lbub_stride_bytes % 4)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: strides must be multiples of the element size");
    hipStream_t st = gq::as_stream(stream);
    const int64_t cs = code_stride_bytes / code_bytes, ls = level_stride_bytes / lsz, bs = lbub_stride_bytes / 4;
    if (code_bytes == 1)
        return gq::dispatch_levels<uint8_t>(static_cast<const uint8_t *>(codes), cs, levels, level_bytes, ls, lb_ub, bs,
                                            codebook, R, M, d, K, n_bit, out, st);
    return gq::dispatch_levels<int32_t>(static_cast<const int32_t *>(codes), cs, levels, level_bytes, ls, lb_ub, bs,
                                        codebook, R, M, d, K, n_bit, out, st);
}

GQ_API int gq_hsq_decode_sum(const void *codes, int code_bytes, const void *levels, int level_bytes,
                             const float *lb_ub, const float *codebook, int R, int64_t M, int d, int K, int n_bit,
                             float *out, void *stream) {
    const int lsz = level_bytes == 0 ? 4 : level_bytes;
    return gq_hsq_decode_sum_strided(codes, code_bytes, M * (int64_t)code_bytes, levels, level_bytes, M * (int64_t)lsz,
                                     lb_ub, 8, codebook, R, M, d, K, n_bit, out, stream);
}
