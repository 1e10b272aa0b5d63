// ProbabilisticVectorCompressor encode (unbiased vector quantisation) for gfx950.
//
// Replaces probabilistic_vector_compressor.py:42-63 of the reference with its INTENDED
// semantics (the reference's own implementation cannot run: it opens a codebook directory that
// does not exist and calls argmin on a bool tensor, which torch >= 1.x rejects -- SURVEY.md 8c;
// parity is therefore pinned by self-consistency tests, not by golden vectors):
//     p     = C_dagger . v                    C_dagger = pinv(codewords^T),  [K, d]     (:47)
//     l1    = sum_k |p_k|                                                            (:48)
//     code  = first k with  cumsum_k(|p|) / l1  >=  r - 1e-5 ,   r ~ U[0,1) per subvector (:52-58)
//     u     = sign(p_code) * l1                                                     (:60-61)
// so that E[ codewords[code] * u ] = sum_k p_k c_k = v  (unbiased for a full-rank codebook).
// One subvector per lane; C_dagger is broadcast from LDS (or read through L1 when it does not
// fit); two sweeps over the K codewords (l1, then the inverse-CDF walk) with the same fmaf chain
// as the NearestNeighbor encode.  Not on the headline path: VALU-bound, ~2*K*d FMAs per subvector.
#include "gq_common.hpp"

namespace gq {

constexpr int PV_THREADS = 256;

template <typename CodeT, int D, bool LDS_CB>
__global__ __launch_bounds__(PV_THREADS) void pvq_encode_kernel(const float *__restrict__ grad,
                                                               const float *__restrict__ cdag, int64_t M, int K,
                                                               int random_mode, const float *__restrict__ r,
                                                               uint64_t seed, CodeT *__restrict__ codes,
                                                               float *__restrict__ u,
                                                               float *__restrict__ partials) {
    extern __shared__ float s_cd[];
    if (LDS_CB) {
        for (int i = threadIdx.x; i < K * D; i += PV_THREADS) s_cd[i] = cdag[i];
        __syncthreads();
    }
    const float *cd = LDS_CB ? s_cd : cdag;
    float lmin = INFINITY, lmax = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * PV_THREADS;
    for (int64_t m = (int64_t)blockIdx.x * PV_THREADS + threadIdx.x; m < M; m += stride) {
        float v[D];
#pragma unroll
        for (int j = 0; j < D; ++j) v[j] = grad[m * D + j];
        float l1 = 0.0f;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) acc = __fmaf_rn(cd[k * D + j], v[j], acc);
            l1 = l1 + fabsf(acc);
        }
        const float rr = (random_mode == GQ_RANDOM_GIVEN) ? r[m] : uniform01(seed, (uint64_t)m);
        const float thr = rr - 1e-5f;
        float cum = 0.0f, sel = 0.0f;
        int code = K - 1;
        bool found = false;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) acc = __fmaf_rn(cd[k * D + j], v[j], acc);
            cum = cum + fabsf(acc) / l1;   // cumsum(|p| / l1), as the reference divides first (:49,:57)
            const bool hit = !found && (cum >= thr);
            if (hit || (!found && k == K - 1)) {
                code = k;
                sel = acc;
            }
            found = found || hit;
        }
        const float sg = (sel > 0.0f) ? 1.0f : ((sel < 0.0f) ? -1.0f : 0.0f);
        const float val = sg * l1;
        codes[m] = (CodeT)code;
        u[m] = val;
        lmin = fminf(lmin, val);
        lmax = fmaxf(lmax, val);
    }
    // per-workgroup (min,max) in the gq_hsq_levels workspace format
    __shared__ float s_min[PV_THREADS / 64], s_max[PV_THREADS / 64];
    lmin = wave_min(lmin);
    lmax = wave_max(lmax);
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = lmin;
        s_max[threadIdx.x >> 6] = lmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = s_min[0], b = s_max[0];
        for (int w = 1; w < PV_THREADS / 64; ++w) {
            a = fminf(a, s_min[w]);
            b = fmaxf(b, s_max[w]);
        }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
        if (blockIdx.x == 0) reinterpret_cast<int *>(partials + 2 * GQ_MAX_PARTIALS)[2] = 0;  // pairs are partials
    }
    if (blockIdx.x == 0)
        for (int i = gridDim.x + threadIdx.x; i < GQ_MAX_PARTIALS; i += PV_THREADS) {
            partials[2 * i] = INFINITY;
            partials[2 * i + 1] = -INFINITY;
        }
}

template <typename CodeT, int D>
static int launch_pvq(const float *grad, const float *cdag, int64_t M, int K, int random_mode, const float *r,
                      uint64_t seed, CodeT *codes, float *u, float *ws, hipStream_t st) {
    int64_t blocks = (M + PV_THREADS - 1) / PV_THREADS;
    int64_t cap = (int64_t)cu_count() * 4;
    if (cap > GQ_MAX_PARTIALS - GQ_FIXUP_PARTIALS) cap = GQ_MAX_PARTIALS - GQ_FIXUP_PARTIALS;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)K * D * sizeof(float);
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(pvq_encode_kernel<CodeT, D, true>), dim3((unsigned)blocks), dim3(PV_THREADS),
                           lds, st, grad, cdag, M, K, random_mode, r, seed, codes, u, ws);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(pvq_encode_kernel<CodeT, D, false>), dim3((unsigned)blocks),
                           dim3(PV_THREADS), 0, st, grad, cdag, M, K, random_mode, r, seed, codes, u, ws);
    GQ_CHECK_LAUNCH("gq_pvq_encode");
    return GQ_OK;
}

template <typename CodeT>
static int dispatch_pvq(const float *grad, const float *cdag, int64_t M, int d, int K, int random_mode,
                        const float *r, uint64_t seed, CodeT *codes, float *u, float *ws, hipStream_t st) {
    switch (d) {
#define GQ_PV_CASE(DD) \
    case DD:           \
        return launch_pvq<CodeT, DD>(grad, cdag, M, K, random_mode, r, seed, codes, u, ws, st);
        GQ_PV_CASE(4)
        GQ_PV_CASE(8)
        GQ_PV_CASE(12)
        GQ_PV_CASE(16)
        GQ_PV_CASE(24)
        GQ_PV_CASE(32)
        GQ_PV_CASE(64)
#undef GQ_PV_CASE
        default:
            return fail(GQ_ERR_UNSUPPORTED, "gq_pvq_encode: built for d in {4,8,12,16,24,32,64}, got %d", d);
    }
}

}  // namespace gq

GQ_API int gq_pvq_encode(const float *grad, const float *c_dagger, int64_t M, int d, int K, int random_mode,
                         const float *r, uint64_t seed, void *codes, int code_bytes, float *u, float *workspace,
                         void *stream) {
    if (M < 1 || d < 1 || K < 1) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: bad sizes");
    if (!grad || !c_dagger || !codes || !u || !workspace) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: null pointer");
    if (random_mode != GQ_RANDOM_GIVEN && random_mode != GQ_RANDOM_DEVICE)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: random_mode must be GIVEN or DEVICE (the sampler needs draws)");
    if (random_mode == GQ_RANDOM_GIVEN && !r) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: r is null");
    hipStream_t st = gq::as_stream(stream);
    if (code_bytes == 1) {
        if (K > 256) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: uint8 codes need K <= 256");
        return gq::dispatch_pvq<uint8_t>(grad, c_dagger, M, d, K, random_mode, r, seed, static_cast<uint8_t *>(codes), u,
                                         workspace, st);
    }
    if (code_bytes == 4)
        return gq::dispatch_pvq<int32_t>(grad, c_dagger, M, d, K, random_mode, r, seed, static_cast<int32_t *>(codes), u,
                                         workspace, st);
    return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: code_bytes must be 1 or 4");
}
