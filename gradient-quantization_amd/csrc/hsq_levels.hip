// Scalar (norm) level quantiser for gfx950.
//
// Replaces probabilistic_scalar_compressor.py:12-27 of the reference.  The global
// min/max (a whole-tensor dependency) is finished here from the per-workgroup
// partials the encode kernel left behind, so compress is two launches and `u` makes
// one round trip through HBM/L2 (0.25 B per gradient element each way).
// Built with -ffp-contract=off: sub, IEEE divide, exact *2^n_bit, truncation -- the
// same roundings as the reference's separate elementwise ops.
#include "hsq_encode_common.hpp"
#include "hsq_levels_common.hpp"
#include <type_traits>

namespace gq {

constexpr int LV_THREADS = 256;

template <typename LevelT>
__global__ __launch_bounds__(LV_THREADS) void hsq_levels_kernel(const float *__restrict__ u, int64_t M, int n_bit,
                                                               int random_mode, const float *__restrict__ r,
                                                               uint64_t seed, const float *__restrict__ partials,
                                                               float *__restrict__ lb_ub,
                                                               LevelT *__restrict__ levels) {
    // Everything this thread needs from memory is requested up front, in ONE round trip: its first two groups of four
    // projections, the `final` flag and its share of the (min,max) partials.  (Flag -> partials -> projections as three
    // dependent round trips made this kernel 5.6 us for 7.8 MB; the partials are read whether or not the flag says they
    // are needed: 8 KiB per workgroup from L2.)
    constexpr bool PK6 = std::is_same<LevelT, Packed6>::value;
    const bool vec4 = (reinterpret_cast<uintptr_t>(u) & 15) == 0 &&
                      (PK6 || (reinterpret_cast<uintptr_t>(levels) & (4 * (PK6 ? 1 : sizeof(LevelT)) - 1)) == 0);
    const int64_t M4 = vec4 ? (M >> 2) : 0;   // whole groups of four, read as one dwordx4
    const int64_t stride = (int64_t)gridDim.x * LV_THREADS;
    const int64_t i0 = (int64_t)blockIdx.x * LV_THREADS + threadIdx.x;
    f32x4 pre[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (i0 + k * stride < M4) pre[k] = reinterpret_cast<const f32x4 *>(u)[i0 + k * stride];
    const int final_flag = ws_counter(partials)[2];
    float2 part[GQ_MAX_PARTIALS / LV_THREADS];
#pragma unroll
    for (int k = 0; k < GQ_MAX_PARTIALS / LV_THREADS; ++k)
        part[k] = reinterpret_cast<const float2 *>(partials)[threadIdx.x + k * LV_THREADS];
    // ---- lb / ub: already final at pair 0 (prefilter encode: its last workgroup), or finished here by
    // every block from the GQ_MAX_PARTIALS pairs ----
    __shared__ float s_min[LV_THREADS / 64], s_max[LV_THREADS / 64];
    float lo, hi;
    if (final_flag != 0) {   // the caller's final pair (gq_hsq.h): rare, read again by everybody
        lo = partials[0];
        hi = partials[1];
    } else {
        lo = INFINITY;
        hi = -INFINITY;
        bool nan = false;   // a (NaN, NaN) pair: some projection was NaN -> lb = ub = NaN, as torch.min / torch.max give
#pragma unroll
        for (int k = 0; k < GQ_MAX_PARTIALS / LV_THREADS; ++k) {
            nan = nan || (part[k].x != part[k].x);
            lo = fminf(lo, part[k].x);
            hi = fmaxf(hi, part[k].y);
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (__ballot(nan) != 0) lo = hi = __uint_as_float(0x7FC00000u);
        if ((threadIdx.x & 63) == 0) {
            s_min[threadIdx.x >> 6] = lo;
            s_max[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        lo = s_min[0];
        hi = s_max[0];
        nan = lo != lo;
#pragma unroll
        for (int w = 1; w < LV_THREADS / 64; ++w) {
            nan = nan || (s_min[w] != s_min[w]);
            lo = fminf(lo, s_min[w]);
            hi = fmaxf(hi, s_max[w]);
        }
        if (nan) lo = hi = __uint_as_float(0x7FC00000u);
    }
    const float lb = lo, ub = hi;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        lb_ub[0] = lb;
        lb_ub[1] = ub;
    }

    const LevelQuant lq(lb, ub, n_bit, random_mode, r, seed);
    auto level_of = [&](float uu, int64_t i) -> int { return lq.level(uu, i); };
    if constexpr (std::is_same<LevelT, Packed6>::value) {
        // four levels per thread and iteration -> one 24-bit group (a NaN quotient's INT_MIN is stored as 0, like the byte form)
        uint8_t *const sec = reinterpret_cast<uint8_t *>(levels);
        const int64_t groups = (M + 3) >> 2;
        int trip = 0;
        for (int64_t i = i0; i < groups; i += stride, ++trip) {
            int l[4] = {0, 0, 0, 0};
            if (i < M4) {
                const f32x4 uu = trip < 2 ? pre[trip < 1 ? 0 : 1] : reinterpret_cast<const f32x4 *>(u)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = level_of(uu[e], 4 * i + e);
            } else {
                for (int e = 0; e < 4 && 4 * i + e < M; ++e) l[e] = level_of(u[4 * i + e], 4 * i + e);
            }
            store_packed6(sec + 3 * i, l[0], l[1], l[2], l[3]);
        }
        return;
    } else {
    // four levels per thread and iteration: one dwordx4 load (the first two: requested at the top), one packed store
    int trip = 0;
    for (int64_t i = i0; i < M4; i += stride, ++trip) {
        const f32x4 uu = trip < 2 ? pre[trip < 1 ? 0 : 1] : reinterpret_cast<const f32x4 *>(u)[i];
        LevelT out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (LevelT)level_of(uu[e], 4 * i + e);
        struct alignas(4 * sizeof(LevelT)) Pack {
            LevelT v[4];
        } pk = {{out[0], out[1], out[2], out[3]}};
        reinterpret_cast<Pack *>(levels)[i] = pk;
    }
    for (int64_t i = 4 * M4 + (int64_t)blockIdx.x * LV_THREADS + threadIdx.x; i < M; i += stride)
        levels[i] = (LevelT)level_of(u[i], i);
    }
}

}  // namespace gq

GQ_API int gq_hsq_levels(const float *u, int64_t M, int n_bit, int random_mode, const float *r, uint64_t seed,
                         const float *minmax_partials, float *lb_ub, void *levels, int level_bytes, void *stream) {
    if (M < 1 || n_bit < 1 || n_bit > 30)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels: bad sizes M=%lld n_bit=%d", (long long)M, n_bit);
    if (!u || !minmax_partials || !lb_ub || !levels) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels: null pointer");
    if (random_mode < GQ_RANDOM_OFF || random_mode > GQ_RANDOM_DEVICE)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels: random_mode %d", random_mode);
    if (random_mode == GQ_RANDOM_GIVEN && !r) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels: r is null");
    const int64_t top = ((int64_t)1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0);
    if ((level_bytes == 1 && top > 255) || (level_bytes == 2 && top > 65535) || (level_bytes == GQ_LEVELS_PACKED6 && top > 63))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels: level_bytes=%d cannot hold level %lld", level_bytes,
                        (long long)top);
    int64_t blocks = (M + gq::LV_THREADS * 8 - 1) / (gq::LV_THREADS * 8);
    const int64_t cap = (int64_t)gq::cu_count() * 4;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipStream_t st = gq::as_stream(stream);
#define GQ_LAUNCH_LEVELS(T)                                                                                      \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gq::hsq_levels_kernel<T>), dim3((unsigned)blocks), dim3(gq::LV_THREADS), 0, \
                       st, u, M, n_bit, random_mode, r, seed, minmax_partials, lb_ub, static_cast<T *>(levels))
    if (level_bytes == 1)
        GQ_LAUNCH_LEVELS(uint8_t);
    else if (level_bytes == 2)
        GQ_LAUNCH_LEVELS(uint16_t);
    else if (level_bytes == 4)
        GQ_LAUNCH_LEVELS(int32_t);
    else if (level_bytes == GQ_LEVELS_PACKED6)
        GQ_LAUNCH_LEVELS(gq::Packed6);
    else
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels: level_bytes must be 1, 2, 4 or GQ_LEVELS_PACKED6");
#undef GQ_LAUNCH_LEVELS
    GQ_CHECK_LAUNCH("gq_hsq_levels");
    return GQ_OK;
}

// ---- standalone min/max partials (ProbabilisticScalarCompressor used on its own) -------
namespace gq {
__global__ __launch_bounds__(LV_THREADS) void minmax_partials_kernel(const float *__restrict__ v, int64_t n,
                                                                    float *__restrict__ partials) {
    __shared__ float s_min[LV_THREADS / 64], s_max[LV_THREADS / 64];
    float lo = INFINITY, hi = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * LV_THREADS;
    bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * LV_THREADS + threadIdx.x; i < n; i += stride) {
        const float x = v[i];
        nan = nan || (x != x);
        lo = fminf(lo, x);
        hi = fmaxf(hi, x);
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (__ballot(nan) != 0) lo = hi = __uint_as_float(0x7FC00000u);   // torch.min / torch.max propagate NaN
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = lo;
        s_max[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool anynan = lo != lo;
#pragma unroll
        for (int w = 1; w < LV_THREADS / 64; ++w) {
            anynan = anynan || (s_min[w] != s_min[w]);
            lo = fminf(lo, s_min[w]);
            hi = fmaxf(hi, s_max[w]);
        }
        if (anynan) lo = hi = __uint_as_float(0x7FC00000u);
        partials[2 * blockIdx.x] = lo;
        partials[2 * blockIdx.x + 1] = hi;
        if (blockIdx.x == 0) ws_counter(partials)[2] = 0;
    }
    if (blockIdx.x == 0)
        for (int i = gridDim.x + threadIdx.x; i < GQ_MAX_PARTIALS; i += LV_THREADS) {
            partials[2 * i] = INFINITY;
            partials[2 * i + 1] = -INFINITY;
        }
}
}  // namespace gq

GQ_API int gq_minmax_partials(const float *v, int64_t n, float *minmax_partials, void *stream) {
    if (n < 1 || !v || !minmax_partials) return gq::fail(GQ_ERR_INVALID_ARG, "gq_minmax_partials: bad arguments");
    int64_t blocks = (n + gq::LV_THREADS * 8 - 1) / (gq::LV_THREADS * 8);
    if (blocks > GQ_MAX_PARTIALS) blocks = GQ_MAX_PARTIALS;
    hipLaunchKernelGGL(gq::minmax_partials_kernel, dim3((unsigned)blocks), dim3(gq::LV_THREADS), 0,
                       gq::as_stream(stream), v, n, minmax_partials);
    GQ_CHECK_LAUNCH("gq_minmax_partials");
    return GQ_OK;
}
