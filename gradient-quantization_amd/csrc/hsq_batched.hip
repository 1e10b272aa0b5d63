// Multi-tensor (segment table) level quantiser and decode-mean: one launch for all the
// tensors of a model that share the (d = 16, K = 256) codebook -- ResNet-50 has 76 of them
// between 1,024 and 2.4 M elements, and per-tensor launches would be launch-bound
// (SURVEY.md 7.3-5).  Same arithmetic as hsq_levels.hip / hsq_decode.hip (which replace
// probabilistic_scalar_compressor.py:12-33, nearest_neighbor_compressor.py:80-90 and
// ps_quantizer.py:48); lb / ub stay PER TENSOR exactly as in the reference.
//
// Index space: every tensor is padded to whole 64-subvector tiles; tile_seg[tile] names its
// tensor, seg_table[seg] = { grad ptr, M, first tile, codes off, levels off, lb/ub off,
// out off (floats), - } (include/gq_hsq.h).  The matching encode is gq_hsq_encode_batched
// (hsq_encode_pf.hip).
#include "hsq_encode_common.hpp"
#include "hsq_levels_common.hpp"
#include <type_traits>

namespace gq {

constexpr int BT_THREADS = 256;

__device__ __forceinline__ float order_unmap_f(unsigned m) {
    return __uint_as_float(m ^ ((m >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// 4 consecutive padded subvectors (one tile quarter-row) per thread and iteration.  LevelT: uint8_t (top level
// <= 255), int16_t or int32_t (gq_hsq_levels_batched_any: e.g. n_bit = 8 with stochastic rounding reaches 256).
template <typename LevelT>
__global__ __launch_bounds__(BT_THREADS) void hsq_levels_batched_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const float *__restrict__ u_flat, const unsigned *__restrict__ seg_minmax, int n_bit, int random_mode,
    uint64_t seed, const float *__restrict__ r_flat, uint8_t *__restrict__ wire, const int64_t *__restrict__ dense_table, int ndense) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    const int64_t total4 = ntiles * 16;
    const int64_t stride = (int64_t)gridDim.x * BT_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * BT_THREADS + threadIdx.x; i < total4; i += stride) {
        const int64_t tile = i >> 4;
        const int seg = tile_seg[tile];
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int64_t m = rec[1];
        const int64_t local = (tile - rec[2]) * 64 + 4 * (i & 15);
        if (local >= m) continue;
        const float lb = order_unmap_f(seg_minmax[2 * seg]), ub = order_unmap_f(seg_minmax[2 * seg + 1]);
        if (local == 0 && !std::is_same<LevelT, float>::value) {
            float *lbub = reinterpret_cast<float *>(wire + rec[5]);
            lbub[0] = lb;
            lbub[1] = ub;
        }
        const float range = ub - lb;
        const bool flat = (lb - ub) == 0.0f;
        const uint64_t sd = random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, lb, ub) : seed;
        const f32x4 uu = *reinterpret_cast<const f32x4 *>(u_flat + 4 * i);
        typename std::conditional<std::is_same<LevelT, Packed6>::value, int, LevelT>::type out[4];   // Packed6: past-the-end slots stay 0
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (std::is_same<LevelT, float>::value) {   // n_bit == 32: the projections travel as they are
                out[e] = uu[e];
                continue;
            }
            int l = 0;
            if (!flat) {
                const float q = (uu[e] - lb) / range;
                const float x = fabsf(q) * s;
                const float c = fminf(fmaxf(x, 0.0f), smax);
                l = (x != x) ? INT32_MIN : (int)c;   // clamp(NaN) stays NaN and NaN -> int32 is INT_MIN in the reference (x86)
                if (random_mode != GQ_RANDOM_OFF) {   // GIVEN: the reference's draws, laid out like u_flat
                    const float prob = x - (float)l;
                    const float rr = random_mode == GQ_RANDOM_GIVEN ? r_flat[4 * i + e] : uniform01(sd, (uint64_t)(4 * i + e));
                    l += (prob > rr) ? 1 : 0;
                }
            }
            if constexpr (std::is_same<LevelT, Packed6>::value)
                out[e] = local + e < m ? l : 0;
            else
                out[e] = (LevelT)l;
        }
        if constexpr (std::is_same<LevelT, Packed6>::value) {
            store_packed6(wire + rec[4] + 3 * (local >> 2), out[0], out[1], out[2], out[3]);   // `local` is a multiple of 4
            continue;
        } else {
        LevelT *dst = reinterpret_cast<LevelT *>(wire + rec[4]) + local;   // sections start 16-byte aligned
        if (local + 3 < m) {
            typedef LevelT L4 __attribute__((ext_vector_type(4)));
            const L4 o4 = {out[0], out[1], out[2], out[3]};
            *reinterpret_cast<L4 *>(dst) = o4;
        } else {
            for (int e = 0; e < 4 && local + e < m; ++e) dst[e] = out[e];
        }
        }
    }
}

// Error-feedback form of the level kernel (ps_quantizer.py:36-39 for every tensor in one launch):
// besides the levels it decodes the tensor it has just quantised and writes
//     error = grad - codebook[code] * (l*(ub-lb)/2^n_bit + lb)
// into the error buffer named by seg_table[seg][7] (skipped when 0).  `grad` is what the EF encode
// left there (grad + scale*error_old).  One thread per (padded subvector, quarter); the four
// threads of a subvector compute the same level, quarter 0 stores it.
template <bool PACKED6>
__global__ __launch_bounds__(BT_THREADS) void hsq_levels_ef_batched_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const float *__restrict__ u_flat, const unsigned *__restrict__ seg_minmax, int n_bit, int random_mode,
    uint64_t seed, const float *__restrict__ r_flat, const float *__restrict__ cb, int K, uint8_t *__restrict__ wire, const int64_t *__restrict__ dense_table, int ndense) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    // rows 20 floats apart: an odd number of 16-byte units spreads the random-row gathers over the banks
    __shared__ __attribute__((aligned(16))) float s_cb[256 * 20];
    for (int i = threadIdx.x; i < K * 16 / 4; i += BT_THREADS)   // (K <= 256 rows; codes stay below K)
        *reinterpret_cast<f32x4 *>(s_cb + (i >> 2) * 20 + 4 * (i & 3)) = reinterpret_cast<const f32x4 *>(cb)[i];
    __syncthreads();
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    const int64_t total = ntiles * 64 * 4;
    const int64_t stride = (int64_t)gridDim.x * BT_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * BT_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t g = i >> 2;
        const int q = (int)(i & 3);
        const int64_t tile = g >> 6;
        const int seg = tile_seg[tile];
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int64_t local = (tile - rec[2]) * 64 + (g & 63);
        if (local >= rec[1]) continue;
        const float lb = order_unmap_f(seg_minmax[2 * seg]), ub = order_unmap_f(seg_minmax[2 * seg + 1]);
        const float range = ub - lb;
        int l = 0;
        if ((lb - ub) != 0.0f) {
            const float x = fabsf((u_flat[g] - lb) / range) * s;
            const float c = fminf(fmaxf(x, 0.0f), smax);
            l = (x != x) ? INT32_MIN : (int)c;   // clamp(NaN) stays NaN and NaN -> int32 is INT_MIN in the reference (x86)
            if (random_mode != GQ_RANDOM_OFF) {   // GIVEN: the reference's draws, laid out like u_flat
                const float prob = x - (float)l;
                const float rr = random_mode == GQ_RANDOM_GIVEN ? r_flat[g] : uniform01(random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, lb, ub) : seed, (uint64_t)g);
                l += (prob > rr) ? 1 : 0;
            }
        }
        if (q == 0) {
            if constexpr (!PACKED6) wire[rec[4] + local] = (uint8_t)l;
            if (local == 0) {
                float *lbub = reinterpret_cast<float *>(wire + rec[5]);
                lbub[0] = lb;
                lbub[1] = ub;
            }
        }
        if constexpr (PACKED6) {
            // the four subvectors of a group sit in 16 consecutive lanes (4 quarters each; groups never straddle a
            // tile); the group's first lane collects their levels.  Lanes of subvectors past the tensor's end have left
            // the loop (`continue` above), so their levels are read as 0 through the ballot.
            const int lane = threadIdx.x & 63, base = lane & ~15;
            const int l1 = __shfl(l, base + 4, 64), l2 = __shfl(l, base + 8, 64), l3 = __shfl(l, base + 12, 64);
            const int64_t m = rec[1];
            if ((lane & 15) == 0)
                store_packed6(wire + rec[4] + 3 * (local >> 2), l, local + 1 < m ? l1 : 0, local + 2 < m ? l2 : 0,
                              local + 3 < m ? l3 : 0);
        }
        float *err = reinterpret_cast<float *>(rec[7]);
        if (!err) continue;
        float n = (float)l * range;   // prob_scalar:31-32, unfused
        n = n / s;
        n = n + lb;
        const int code = wire[rec[3] + local];
        const f32x4 c = *reinterpret_cast<const f32x4 *>(s_cb + code * 20 + 4 * q);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(rec[0]) + local * 16 + 4 * q);
        f32x4 e;
        e[0] = v[0] - c[0] * n;
        e[1] = v[1] - c[1] * n;
        e[2] = v[2] - c[2] * n;
        e[3] = v[3] - c[3] * n;
        *reinterpret_cast<f32x4 *>(err + local * 16 + 4 * q) = e;
    }
}

// One thread per 4 output floats of the padded space; payload r starts at gathered + r*user_stride.
__global__ __launch_bounds__(BT_THREADS) void hsq_decode_sum_batched_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const uint8_t *__restrict__ gathered, int64_t user_stride, int R, const float *__restrict__ cb, int K, int n_bit,
    float *__restrict__ out, int plain) {
    // rows 20 floats apart: an odd number of 16-byte units spreads the random-row gathers over the banks
    __shared__ __attribute__((aligned(16))) float s_cb[256 * 20];
    for (int i = threadIdx.x; i < K * 16 / 4; i += BT_THREADS)   // (K <= 256 rows; codes stay below K)
        *reinterpret_cast<f32x4 *>(s_cb + (i >> 2) * 20 + 4 * (i & 3)) = reinterpret_cast<const f32x4 *>(cb)[i];
    __syncthreads();
    const float s = (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);   // the aggregate of R users (ps_quantizer.py:48) unless the caller asked for the plain decode
    const int64_t total = ntiles * 64 * 4;
    const int64_t stride = (int64_t)gridDim.x * BT_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * BT_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t g = i >> 2;
        const int q = (int)(i & 3);
        const int64_t tile = g >> 6;
        const int seg = tile_seg[tile];
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int64_t local = (tile - rec[2]) * 64 + (g & 63);
        if (local >= rec[1]) continue;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int r = 0; r < R; ++r) {
            const uint8_t *p = gathered + (int64_t)r * user_stride;
            const int code = p[rec[3] + local];
            const float *lbub = reinterpret_cast<const float *>(p + rec[5]);
            const float lb = lbub[0], range = lbub[1] - lb;
            float n = (float)p[rec[4] + local] * range;   // prob_scalar:31-32, unfused
            n = n / s;
            n = n + lb;
            const f32x4 c = *reinterpret_cast<const f32x4 *>(s_cb + code * 20 + 4 * q);
            f32x4 dec;
            dec[0] = c[0] * n;
            dec[1] = c[1] * n;
            dec[2] = c[2] * n;
            dec[3] = c[3] * n;
            if (r == 0) {
                acc = dec;
            } else {
                acc[0] = acc[0] + dec[0];
                acc[1] = acc[1] + dec[1];
                acc[2] = acc[2] + dec[2];
                acc[3] = acc[3] + dec[3];
            }
        }
        if (md.apply) {
            acc[0] = mean_div(acc[0], md);
            acc[1] = mean_div(acc[1], md);
            acc[2] = mean_div(acc[2], md);
            acc[3] = mean_div(acc[3], md);
        }
        *reinterpret_cast<f32x4 *>(out + rec[6] + local * 16 + 4 * q) = acc;
    }
}

// The d = 16 / K = 256 multi-tensor decode-mean is built like the per-tensor one (hsq_decode.hip,
// hsq_decode_sum_d16u8_r_kernel): a thread produces one quarter of FOUR consecutive padded subvectors (they share a
// tile, hence a tensor); the codebook is staged four times (row r, copy c at byte r*256 + c*64) and the four 4-lane
// teams of every ds_read_b128 lane group read copies 0..3.  Helpers first, the kernels below.

// LDS byte address of a codebook row for this lane: [0, 0, code_k, lane_const] by one v_perm_b32
template <int K4>
__device__ __forceinline__ unsigned bt4_row_addr(unsigned c4, unsigned lane_const) {
    return __builtin_amdgcn_perm(c4, lane_const, 0x0c0c0000u | ((4u + K4) << 8));
}

typedef const f32x4 __attribute__((address_space(3))) bt4_lds_f32x4;
template <bool FIRST, bool PACKED6, bool ABS0 = false, bool FMA = false>   // ABS0: the image starts at LDS address 0; FMA: opt-in fused accumulation (see dec16_payload)
__device__ __forceinline__ void bt4_payload(f32x4 (&acc)[4], unsigned c4, unsigned l4, float lb, float ub, float inv_s, int q,
                                            const char *cb_bytes, unsigned lane_const) {
    const float range = ub - lb;
    // lane q of a team works out the norm of subvector q; the team shares them by quad-permute DPP moves
    const unsigned lq = PACKED6 ? ((l4 >> (6 * q)) & 63u) : ((l4 >> (8 * q)) & 255u);
    float n_own = (float)lq * range;   // prob_scalar:31-32, unfused
    n_own = n_own * inv_s;                                   // == / 2^n_bit exactly
    n_own = n_own + lb;
    const int n_bits = __builtin_bit_cast(int, n_own);
    const float n_team[4] = {
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0x00, 0xF, 0xF, true)),
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0x55, 0xF, 0xF, true)),
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0xAA, 0xF, 0xF, true)),
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0xFF, 0xF, 0xF, true))};
    const unsigned a[4] = {bt4_row_addr<0>(c4, lane_const), bt4_row_addr<1>(c4, lane_const), bt4_row_addr<2>(c4, lane_const),
                           bt4_row_addr<3>(c4, lane_const)};
    if constexpr (FMA && !FIRST) {      // one v_fmac_f32_dpp per element: the team's norms are read across the quad by the multiply-add itself
        const float n_rdy = quad_norm_ready(n_own);
        const f32x4 c0 = ABS0 ? *reinterpret_cast<bt4_lds_f32x4 *>((uintptr_t)a[0]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[0]);
        const f32x4 c1 = ABS0 ? *reinterpret_cast<bt4_lds_f32x4 *>((uintptr_t)a[1]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[1]);
        const f32x4 c2 = ABS0 ? *reinterpret_cast<bt4_lds_f32x4 *>((uintptr_t)a[2]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[2]);
        const f32x4 c3 = ABS0 ? *reinterpret_cast<bt4_lds_f32x4 *>((uintptr_t)a[3]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[3]);
        acc[0] = fmac_quad4<0>(acc[0], n_rdy, c0);
        acc[1] = fmac_quad4<1>(acc[1], n_rdy, c1);
        acc[2] = fmac_quad4<2>(acc[2], n_rdy, c2);
        acc[3] = fmac_quad4<3>(acc[3], n_rdy, c3);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float n = n_team[k];
        const f32x4 c = ABS0 ? *reinterpret_cast<bt4_lds_f32x4 *>((uintptr_t)a[k])
                             : *reinterpret_cast<const f32x4 *>(cb_bytes + a[k]);
        const f32x4 n4 = {n, n, n, n};
        const f32x4 dec = c * n4;
        if constexpr (FIRST) {
            acc[k] = dec;
        } else {
            acc[k] = acc[k] + dec;
        }
    }
}

// The same for a compile-time payload count R <= BT4_RMAX, software-pipelined across a wave's tiles like
// hsq_decode_sum_d16u8_r_kernel (hsq_decode.hip): a wave's 64 items are exactly one tile, so the segment record of a tile
// is wave-uniform -- it is fetched by scalar loads two tiles ahead, the (code, level) words and (lb, ub) of the NEXT tile
// are re-requested into the registers of the payload just consumed, and the stores of a tile drain under the next
// tile's arithmetic.  In the kernel above every item walks tile_seg -> segment record -> payload words -> stores as one
// dependent chain per lane, behind the previous item's stores.
constexpr int BT4_RMAX = 16;

struct Bt4Tile {          // wave-uniform: where a tile's bytes are
    uint64_t code_off;    // byte offset of the tile's first code inside a payload
    uint64_t level_off;   // ... of its first level (byte or packed group)
    uint64_t lbub_off;    // ... of the segment's (lb, ub)
    int64_t out_off;      // float offset of the tile's first output
    int left;             // subvectors of the segment from the tile's start on (>= 1), capped at 64
};

template <bool PACKED6>
__device__ __forceinline__ Bt4Tile bt4_tile(const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t tile) {
    const int seg = tile_seg[tile];
    const int64_t *rec = seg_table + 8 * (int64_t)seg;
    const int64_t local0 = (tile - rec[2]) * 64;
    const int64_t left = rec[1] - local0;
    Bt4Tile t;
    t.code_off = (uint64_t)(rec[3] + local0);
    t.level_off = (uint64_t)(rec[4] + (PACKED6 ? 3 * (local0 >> 2) : local0));
    t.lbub_off = (uint64_t)rec[5];
    t.out_off = rec[6] + local0 * 16;
    t.left = left < 64 ? (int)left : 64;
    return t;
}

// R >= BT4_R_NARROW: 2R word registers + 16 sums + the rows in flight do not fit the 64 registers of 8 waves per SIMD
// (the compiler spilled 12-13 dwords per lane and trip); those run 768-thread workgroups, 6 waves per SIMD, 80 registers.
#ifndef GQ_BT4_R_NARROW
#define GQ_BT4_R_NARROW 6
#endif
// R > 8: (lb, ub) of 9..16 payloads in flight for two tiles push the kernel past 80 registers: one 1024-thread workgroup
// per CU, 4 waves per SIMD (the arithmetic of that many payloads outweighs the stores anyway).
constexpr int bt4r_threads(int R) { return R > 8 ? 1024 : (R >= GQ_BT4_R_NARROW ? 768 : 1024); }
constexpr int bt4r_waves(int R) { return R > 8 ? 4 : (R >= GQ_BT4_R_NARROW ? 6 : 8); }

template <int R, bool PACKED6, bool FMA = false>
__global__ __launch_bounds__(bt4r_threads(R)) __attribute__((amdgpu_waves_per_eu(bt4r_waves(R), bt4r_waves(R))))
void hsq_decode_sum_batched4_r_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const uint8_t *__restrict__ gathered, int64_t user_stride, const float *__restrict__ cb, int K, int n_bit,
    float *__restrict__ out, int plain, const StepTail tail) {
    extern __shared__ __attribute__((aligned(16))) float s_cb4[];   // [256][4 copies][16] at LDS address 0 (bt4_payload<.., ABS0>)
    step_tail_run(tail);
    constexpr int THREADS = bt4r_threads(R);
    const float inv_s = 1.0f / (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);
    const int q = threadIdx.x & 3;
    const unsigned lane_const = (unsigned)(((threadIdx.x >> 3) & 3) * 64 + 16 * q);
    const char *const cb_bytes = reinterpret_cast<const char *>(s_cb4);
    const unsigned g = (unsigned)(threadIdx.x & 63) & ~3u;   // this lane's first subvector inside a tile
    const unsigned out_lane = (g * 16 + 4 * q) * 4;          // bytes
    // one tile per wave and trip; the wave's tile number is uniform by construction, told to the compiler
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)));
    const int64_t wstride = (int64_t)gridDim.x * (THREADS / 64);
    typedef const uint8_t __attribute__((address_space(1))) gbyte;
    typedef const unsigned __attribute__((address_space(1))) gword;
    typedef const unsigned __attribute__((address_space(1), aligned(1))) gword_any;   // packed levels start at any byte
    const uint64_t wire0 = reinterpret_cast<uint64_t>(gathered);
    // Each lane of a team of four fetches the word pair of ONE payload in four (lane q: payloads q, 4 + q, ...) and a
    // payload's pair reaches the team by a quad-permute DPP move when its turn comes (hsq_decode_sum_d16u8_r_kernel): a
    // quarter of the wave-wide loads and of the word registers.  The lane's payload inside a group of four goes into the 32-bit
    // lane offset (launcher: 3 payload strides fit); lanes past R in the last group repeat payload R - 1.
    constexpr int NW = (R + 3) / 4;
    constexpr int LASTN = R - 4 * (NW - 1);
    unsigned cw[NW], lw[NW];
    float lb[R], ub[R];
    const unsigned qs = (unsigned)q * (unsigned)user_stride;
    const unsigned qs_last = (unsigned)(q < LASTN ? q : LASTN - 1) * (unsigned)user_stride;
    // requests of tile t for the payloads of group j: (scalar base) + (32-bit lane offset) loads; lanes past the segment's
    // end re-read the tile's first group of subvectors
    auto request_group = [&](const Bt4Tile &t, unsigned gl, int j) {
        const uint64_t base = wire0 + (uint64_t)(4 * j) * (uint64_t)user_stride;
        const unsigned qo = (LASTN != 4 && j == NW - 1) ? qs_last : qs;
        cw[j] = *reinterpret_cast<gword *>(reinterpret_cast<gbyte *>(base + t.code_off) + (gl + qo));
        lw[j] = PACKED6 ? (unsigned)*reinterpret_cast<gword_any *>(reinterpret_cast<gbyte *>(base + t.level_off) + (3u * (gl >> 2) + qo))
                        : *reinterpret_cast<gword *>(reinterpret_cast<gbyte *>(base + t.level_off) + (gl + qo));
    };
    auto request_lbub = [&](const Bt4Tile &t, int r) {
        const float *lbub = reinterpret_cast<const float *>(gathered + (int64_t)r * user_stride + (int64_t)t.lbub_off);
        lb[r] = lbub[0];
        ub[r] = lbub[1];
    };
    // the first tile's record and payload words are requested BEFORE the codebook image is staged: their round trips run
    // under the staging.  (A wave without a tile still takes part in the barrier.)
    const bool active = wave0 < ntiles;
    int64_t tile = active ? wave0 : ntiles - 1;
    Bt4Tile cur = bt4_tile<PACKED6>(seg_table, tile_seg, tile);
    int64_t tn = tile + wstride < ntiles ? tile + wstride : tile;
    Bt4Tile nxt = bt4_tile<PACKED6>(seg_table, tile_seg, tn);
    {
        const unsigned gl = (int)g < cur.left ? g : 0u;
#pragma unroll
        for (int j = 0; j < NW; ++j) request_group(cur, gl, j);
#pragma unroll
        for (int r = 0; r < R; ++r) request_lbub(cur, r);
    }
    {
        constexpr int STAGE = (256 * 16 + THREADS - 1) / THREADS;
        f32x4 stage[STAGE];
#pragma unroll
        for (int n = 0; n < STAGE; ++n) {
            const int e = threadIdx.x + n * THREADS;   // (row, copy, quarter)
            if (e < K * 16) stage[n] = *reinterpret_cast<const f32x4 *>(cb + (e >> 4) * 16 + 4 * (e & 3));
        }
#pragma unroll
        for (int n = 0; n < STAGE; ++n) {
            const int e = threadIdx.x + n * THREADS;
            const int row = e >> 4, c = (e >> 2) & 3, qq = e & 3;
            if (e < K * 16) *reinterpret_cast<f32x4 *>(s_cb4 + row * 64 + c * 16 + 4 * qq) = stage[n];
        }
    }
    __syncthreads();
    if (!active) return;
    while (true) {
        const int64_t t2 = tn + wstride < ntiles ? tn + wstride : tn;
        const Bt4Tile aft = bt4_tile<PACKED6>(seg_table, tile_seg, t2);   // scalar loads, two tiles ahead
        unsigned gl = (int)g < nxt.left ? g : 0u;
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned c4 = team_word(cw[r >> 2], r & 3), l4 = team_word(lw[r >> 2], r & 3);
            if (r == 0)
                bt4_payload<true, PACKED6, true>(acc, c4, l4, lb[r], ub[r], inv_s, q, cb_bytes, lane_const);
            else
                bt4_payload<false, PACKED6, true, FMA>(acc, c4, l4, lb[r], ub[r], inv_s, q, cb_bytes, lane_const);
            // keep each re-request behind the payload it replaces (hoisted to the top of the trip, the new words were
            // spilled until their registers came free): the lane offset is made to "depend" on the payload's last sum.
            // No instruction; a sched_barrier or a volatile asm counts as a store and turns the scalar (lb, ub) loads into vector loads
            asm("" : "+v"(gl) : "v"(acc[3][3]));
            request_lbub(nxt, r);
            if ((r & 3) == 3 || r == R - 1) request_group(nxt, gl, r >> 2);   // the group's payloads are summed: its registers take the next tile's
        }
        if (md.apply) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[k][0] = mean_div(acc[k][0], md);
                acc[k][1] = mean_div(acc[k][1], md);
                acc[k][2] = mean_div(acc[k][2], md);
                acc[k][3] = mean_div(acc[k][3], md);
            }
        }
        f32x4 *o = reinterpret_cast<f32x4 *>(out + cur.out_off + (out_lane >> 2));   // through `out` itself: a pointer rebuilt from an
                                                                                   // integer may alias the wire as far as the compiler knows,
                                                                                   // and the uniform (lb, ub) loads stop being scalar loads
        if (cur.left == 64) {   // wave-uniform: the whole tile belongs to the segment
#pragma unroll
            for (int k = 0; k < 4; ++k) o[4 * k] = acc[k];
        } else {
            const int nv = cur.left - (int)g;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) o[4 * k] = acc[k];
        }
        if (tile + wstride >= ntiles) break;
        tile += wstride;
        cur = nxt;
        nxt = aft;
        tn = t2;
    }
}

template <int R, bool P6, bool FMA = false>
static void launch_bt4_r(const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles, const uint8_t *gathered,
                         int64_t user_stride, const float *cb, int K, int n_bit, float *out, int plain, hipStream_t st, const StepTail &tail) {
    static const int bpc = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_decode_sum_batched4_r_kernel<R, P6, FMA>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipGetLastError();
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hsq_decode_sum_batched4_r_kernel<R, P6, FMA>, bt4r_threads(R),
                                                         (size_t)64 * 1024) != hipSuccess || n < 1)
            n = 1;
        return n;
    }();
    int64_t blocks = (ntiles * 64 + bt4r_threads(R) - 1) / bt4r_threads(R);
    if (blocks > (int64_t)cu_count() * bpc) blocks = (int64_t)cu_count() * bpc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_batched4_r_kernel<R, P6, FMA>), dim3((unsigned)blocks), dim3(bt4r_threads(R)),
                       (size_t)64 * 1024, st, seg_table, tile_seg, ntiles, gathered, user_stride, cb, K, n_bit, out, plain, tail);
}

// Any R above BT4_RMAX: the same pipeline in chunks of BT4_RMAX payloads (see hsq_decode_sum_d16u8_rc_kernel): the words of
// (tile, chunk) sit in registers and each pair is re-requested for the next chunk -- of the same tile, or the first chunk
// of the wave's next tile -- as soon as it has been consumed; every trip issues its loads unconditionally (a short last
// chunk re-reads the last payload), which keeps the compiler's vmcnt bookkeeping exact.  (lb, ub) per payload by scalar
// loads at consumption (they were vector loads per lane before).  Sums start from +0 (R > 1: the mean adds +0 anyway).
constexpr int BT4_RC_THREADS = 1024;
template <bool PACKED6>
__global__ __launch_bounds__(BT4_RC_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))   // 16 word registers + 16 sums + rows + two tile records: ~90 VGPRs
void hsq_decode_sum_batched4_rc_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const uint8_t *__restrict__ gathered, int64_t user_stride, int R, const float *__restrict__ cb, int K, int n_bit,
    float *__restrict__ out, int plain, const StepTail tail) {
    extern __shared__ __attribute__((aligned(16))) float s_cb4[];   // [256][4 copies][16] at LDS address 0
    step_tail_run(tail);
    constexpr int THREADS = BT4_RC_THREADS, C = 8;
    const float inv_s = 1.0f / (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);
    const int q = threadIdx.x & 3;
    const unsigned lane_const = (unsigned)(((threadIdx.x >> 3) & 3) * 64 + 16 * q);
    const char *const cb_bytes = reinterpret_cast<const char *>(s_cb4);
    const unsigned g = (unsigned)(threadIdx.x & 63) & ~3u;
    const unsigned out_lane = (g * 16 + 4 * q) * 4;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)));
    const int64_t wstride = (int64_t)gridDim.x * (THREADS / 64);
    const int nchunks = (R + C - 1) / C;
    typedef const uint8_t __attribute__((address_space(1))) gbyte;
    typedef const unsigned __attribute__((address_space(1))) gword;
    typedef const unsigned __attribute__((address_space(1), aligned(1))) gword_any;
    const uint64_t wire0 = reinterpret_cast<uint64_t>(gathered);
    unsigned c4[C], l4[C];
    auto request = [&](const Bt4Tile &t, unsigned gl, int chunk, int jj) {
        int r = chunk * C + jj;
        r = r < R ? r : R - 1;
        const uint64_t base = wire0 + (uint64_t)r * (uint64_t)user_stride;
        c4[jj] = *reinterpret_cast<gword *>(reinterpret_cast<gbyte *>(base + t.code_off) + gl);
        l4[jj] = PACKED6 ? (unsigned)*reinterpret_cast<gword_any *>(reinterpret_cast<gbyte *>(base + t.level_off) + 3u * (gl >> 2))
                         : *reinterpret_cast<gword *>(reinterpret_cast<gbyte *>(base + t.level_off) + gl);
    };
    const bool active = wave0 < ntiles;
    int64_t tile = active ? wave0 : ntiles - 1;
    Bt4Tile cur = bt4_tile<PACKED6>(seg_table, tile_seg, tile);
    int64_t tn = tile + wstride < ntiles ? tile + wstride : tile;
    Bt4Tile nxt = bt4_tile<PACKED6>(seg_table, tile_seg, tn);
    {
        const unsigned gl = (int)g < cur.left ? g : 0u;
#pragma unroll
        for (int jj = 0; jj < C; ++jj) request(cur, gl, 0, jj);
    }
    {
        constexpr int STAGE = (256 * 16 + THREADS - 1) / THREADS;
        f32x4 stage[STAGE];
#pragma unroll
        for (int n = 0; n < STAGE; ++n) {
            const int e = threadIdx.x + n * THREADS;
            if (e < K * 16) stage[n] = *reinterpret_cast<const f32x4 *>(cb + (e >> 4) * 16 + 4 * (e & 3));
        }
#pragma unroll
        for (int n = 0; n < STAGE; ++n) {
            const int e = threadIdx.x + n * THREADS;
            const int row = e >> 4, c = (e >> 2) & 3, qq = e & 3;
            if (e < K * 16) *reinterpret_cast<f32x4 *>(s_cb4 + row * 64 + c * 16 + 4 * qq) = stage[n];
        }
    }
    __syncthreads();
    if (!active) return;
    while (true) {
        const int64_t t2 = tn + wstride < ntiles ? tn + wstride : tn;
        const Bt4Tile aft = bt4_tile<PACKED6>(seg_table, tile_seg, t2);
        const unsigned gl_cur = (int)g < cur.left ? g : 0u, gl_nxt = (int)g < nxt.left ? g : 0u;
        f32x4 acc[4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool last = chunk + 1 == nchunks;
            const int nc = last ? 0 : chunk + 1;
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                const int r = chunk * C + jj;
                if (r < R) {
                    const float *lbub = reinterpret_cast<const float *>(gathered + (int64_t)r * user_stride + (int64_t)cur.lbub_off);
                    bt4_payload<false, PACKED6, true>(acc, c4[jj], l4[jj], lbub[0], lbub[1], inv_s, q, cb_bytes, lane_const);
                }
                if (last)
                    request(nxt, gl_nxt, 0, jj);
                else
                    request(cur, gl_cur, nc, jj);
            }
        }
        if (md.apply) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[k][0] = mean_div(acc[k][0], md);
                acc[k][1] = mean_div(acc[k][1], md);
                acc[k][2] = mean_div(acc[k][2], md);
                acc[k][3] = mean_div(acc[k][3], md);
            }
        }
        f32x4 *o = reinterpret_cast<f32x4 *>(out + cur.out_off + (out_lane >> 2));
        if (cur.left == 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[4 * k] = acc[k];
        } else {
            const int nv = cur.left - (int)g;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) o[4 * k] = acc[k];
        }
        if (tile + wstride >= ntiles) break;
        tile += wstride;
        cur = nxt;
        nxt = aft;
        tn = t2;
    }
}

template <bool P6>
static void launch_bt4_rc(int R, const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles, const uint8_t *gathered,
                          int64_t user_stride, const float *cb, int K, int n_bit, float *out, int plain, hipStream_t st, const StepTail &tail) {
    static const int bpc = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_decode_sum_batched4_rc_kernel<P6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipGetLastError();
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hsq_decode_sum_batched4_rc_kernel<P6>, BT4_RC_THREADS, (size_t)64 * 1024) != hipSuccess ||
            n < 1)
            n = 1;
        return n;
    }();
    int64_t blocks = (ntiles * 64 + BT4_RC_THREADS - 1) / BT4_RC_THREADS;
    if (blocks > (int64_t)cu_count() * bpc) blocks = (int64_t)cu_count() * bpc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_batched4_rc_kernel<P6>), dim3((unsigned)blocks), dim3(BT4_RC_THREADS), (size_t)64 * 1024, st,
                       seg_table, tile_seg, ntiles, gathered, user_stride, R, cb, K, n_bit, out, plain, tail);
}

template <bool P6>
static void launch_bt4_fixed_r(int R, const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles, const uint8_t *gathered,
                               int64_t user_stride, const float *cb, int K, int n_bit, float *out, int plain, hipStream_t st, bool fma,
                               const StepTail &tail) {
    // (the lane's payload inside a group of four travels in the 32-bit offset of its loads: 3 strides + a payload must fit;
    // wires of a gigabyte and more per user take the chunked kernel, whose bases are 64-bit)
    const bool fits32 = user_stride >= 0 && 4 * user_stride < ((int64_t)1 << 32);
    if (fma && fits32 && !plain) {   // GQ_AGGREGATE_FMA: the power-of-two payload counts; every other R keeps the exact kernels
        switch (R) {
#define GQ_BT4_FMA(N) case N: launch_bt4_r<N, P6, true>(seg_table, tile_seg, ntiles, gathered, user_stride, cb, K, n_bit, out, plain, st, tail); return;
            GQ_BT4_FMA(2) GQ_BT4_FMA(4) GQ_BT4_FMA(8) GQ_BT4_FMA(16)
#undef GQ_BT4_FMA
            default: break;
        }
    }
    switch (fits32 ? R : 0) {
#define GQ_BT4_CASE(N) case N: launch_bt4_r<N, P6>(seg_table, tile_seg, ntiles, gathered, user_stride, cb, K, n_bit, out, plain, st, tail); return;
        GQ_BT4_CASE(1) GQ_BT4_CASE(2) GQ_BT4_CASE(3) GQ_BT4_CASE(4)
        GQ_BT4_CASE(5) GQ_BT4_CASE(6) GQ_BT4_CASE(7) GQ_BT4_CASE(8)
        GQ_BT4_CASE(9) GQ_BT4_CASE(10) GQ_BT4_CASE(11) GQ_BT4_CASE(12)
        GQ_BT4_CASE(13) GQ_BT4_CASE(14) GQ_BT4_CASE(15) GQ_BT4_CASE(16)
#undef GQ_BT4_CASE
        default:
            launch_bt4_rc<P6>(R, seg_table, tile_seg, ntiles, gathered, user_stride, cb, K, n_bit, out, plain, st, tail);
            return;
    }
}

// Error-feedback level kernel for the other prefilter sub-dimensions (D = 8, 32): as
// hsq_levels_ef_batched_kernel, one thread per (padded subvector, 4-float unit).
template <int D>
__global__ __launch_bounds__(BT_THREADS) void hsq_levels_ef_batched_d_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const float *__restrict__ u_flat, const unsigned *__restrict__ seg_minmax, int n_bit, int random_mode,
    uint64_t seed, const float *__restrict__ r_flat, const float *__restrict__ cb, int K, uint8_t *__restrict__ wire, const int64_t *__restrict__ dense_table, int ndense) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    constexpr int UPS = D / 4;
    constexpr int RS = ((D / 4) & 1) ? D : D + 4;
    __shared__ __attribute__((aligned(16))) float s_cb[256 * RS];
    for (int i = threadIdx.x; i < K * UPS; i += BT_THREADS)   // (K <= 256 rows; codes stay below K)
        *reinterpret_cast<f32x4 *>(s_cb + (i / UPS) * RS + 4 * (i % UPS)) = reinterpret_cast<const f32x4 *>(cb)[i];
    __syncthreads();
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    const int64_t total = ntiles * 64 * UPS;
    const int64_t stride = (int64_t)gridDim.x * BT_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * BT_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t g = i / UPS;
        const int q = (int)(i % UPS);
        const int64_t tile = g >> 6;
        const int seg = tile_seg[tile];
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int64_t local = (tile - rec[2]) * 64 + (g & 63);
        if (local >= rec[1]) continue;
        const float lb = order_unmap_f(seg_minmax[2 * seg]), ub = order_unmap_f(seg_minmax[2 * seg + 1]);
        const float range = ub - lb;
        int l = 0;
        if ((lb - ub) != 0.0f) {
            const float x = fabsf((u_flat[g] - lb) / range) * s;
            const float c = fminf(fmaxf(x, 0.0f), smax);
            l = (x != x) ? INT32_MIN : (int)c;   // clamp(NaN) stays NaN and NaN -> int32 is INT_MIN in the reference (x86)
            if (random_mode != GQ_RANDOM_OFF) {   // GIVEN: the reference's draws, laid out like u_flat
                const float prob = x - (float)l;
                const float rr = random_mode == GQ_RANDOM_GIVEN ? r_flat[g] : uniform01(random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, lb, ub) : seed, (uint64_t)g);
                l += (prob > rr) ? 1 : 0;
            }
        }
        if (q == 0) {
            wire[rec[4] + local] = (uint8_t)l;
            if (local == 0) {
                float *lbub = reinterpret_cast<float *>(wire + rec[5]);
                lbub[0] = lb;
                lbub[1] = ub;
            }
        }
        float *err = reinterpret_cast<float *>(rec[7]);
        if (!err) continue;
        float n = (float)l * range;   // prob_scalar:31-32, unfused
        n = n / s;
        n = n + lb;
        const int code = wire[rec[3] + local];
        const f32x4 c = *reinterpret_cast<const f32x4 *>(s_cb + code * RS + 4 * q);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(rec[0]) + local * D + 4 * q);
        f32x4 e;
        e[0] = v[0] - c[0] * n;
        e[1] = v[1] - c[1] * n;
        e[2] = v[2] - c[2] * n;
        e[3] = v[3] - c[3] * n;
        *reinterpret_cast<f32x4 *>(err + local * D + 4 * q) = e;
    }
}

// decode-mean over a segment table for K = 256 and D = 8 / 16 / 32 with byte or 16-bit levels (main.py:90-92's own defaults
// -- c_dim 32, n_bit 8, stochastic rounding -- reach level 256: the levels travel as int16): a WAVE per tile of 64 padded
// subvectors.  Round 4 ran one thread per (subvector, 16-byte unit) -- every thread fetched its tensor's record, the
// payload's (lb, ub), the code and the level for 16 bytes of output: 38.8 us for the ResNet-50 list at D = 32, int16 levels,
// one payload (2.4 TB/s written).  Here lane l is subvector l when the payloads are read -- ONE code and ONE level load per
// subvector and payload, the norm computed once (level * range, * 2^-n, + lb: unfused, probabilistic_scalar_compressor.py:31-32)
// and parked with the code in the wave's LDS slots -- and unit l of each of the D / 4 passes when the tile is written:
// unit u = 64 p + l belongs to subvector u / (D / 4), its lanes read that subvector's (code, norm) pair (a broadcast read),
// the codebook row's four floats, multiply, add in PAYLOAD ORDER (chunks of DT_RCH payloads, ascending), and every store
// instruction of the wave writes one contiguous kilobyte.  The tile's record and the payloads' (lb, ub) are wave-uniform.
constexpr int DT_THREADS = 512;
constexpr int DT_WAVES = DT_THREADS / 64;
constexpr int DT_RCH = 8;   // payloads staged per chunk
constexpr int DT_RB = 4;    // ... of which this many are read back from LDS together in a pass
// gq_step_tail.ticket of gq_hsq_levels_decode_batched: (1 + GQ_TICKET_SHARDS) counters GQ_TICKET_STRIDE words apart (include/gq_hsq.h: GQ_TICKET_WORDS)
#define GQ_TICKET_SHARDS 16
#define GQ_TICKET_STRIDE 32
#ifndef GQ_EF_LEVELS_TILE
#define GQ_EF_LEVELS_TILE 1   // 0: round 4's one-thread-per-unit error-feedback level kernels (A/B builds)
#endif
// RB: payloads whose loads go out together (4 for R >= 4, 2 for R = 2 / 3); 0: the one-by-one loops alone (fewer registers:
// R = 1 ran 3-8 % slower with the batch code merely present).
template <int D, typename LevelT, int RB>
__global__ __launch_bounds__(DT_THREADS) void hsq_decode_sum_batched_tile_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const uint8_t *__restrict__ gathered, int64_t user_stride, int R, const float *__restrict__ cb, int K, int n_bit,
    float *__restrict__ out, int plain, const StepTail tail) {
    step_tail_run(tail);
    constexpr int DT_RBK = RB > 0 ? RB : 1;
    constexpr int UPS = D / 4;                               // 16-byte units per subvector = passes per tile
    constexpr int RS = ((D / 4) & 1) ? D : D + 4;            // LDS row stride in floats: an odd number of 16-byte units
    __shared__ __attribute__((aligned(16))) float s_cb[256 * RS];
    __shared__ __attribute__((aligned(8))) unsigned s_pair[DT_WAVES * DT_RCH * 64 * 2];   // { code, bits of the norm } per (wave, payload of the chunk, subvector)
    for (int i = threadIdx.x; i < K * UPS; i += DT_THREADS)   // (K <= 256 rows; codes stay below K)
        *reinterpret_cast<f32x4 *>(s_cb + (i / UPS) * RS + 4 * (i % UPS)) = reinterpret_cast<const f32x4 *>(cb)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float inv_s = 1.0f / (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);
    auto uniform64 = [](int64_t v) {   // a wave-uniform value read through a vector path -> SGPRs
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    unsigned *const pairs = s_pair + wave * (DT_RCH * 64 * 2);
    const int q = lane % UPS, ls = lane / UPS;               // this lane's unit inside a subvector / subvector inside a pass
    // (the tensor's record stays in scalar registers while a wave's tiles stay with one tensor; the next tile's tensor word is
    // requested a tile ahead: as in hsq_levels_ef_tile_kernel)
    const int64_t tstride = (int64_t)gridDim.x * DT_WAVES;
    int64_t tile = (int64_t)blockIdx.x * DT_WAVES + wave;
    int seg = -1;
    int64_t r_m = 0, r_tile0 = 0, r_codes = 0, r_levels = 0, r_lbub = 0, r_out = 0;
    int seg_nv = tile < ntiles ? tile_seg[tile] : 0;
    for (; tile < ntiles; tile += tstride) {
        const int seg_t = __builtin_amdgcn_readfirstlane(seg_nv);
        if (tile + tstride < ntiles) seg_nv = tile_seg[tile + tstride];
        if (seg_t != seg) {   // (wave-uniform)
            seg = seg_t;
            const int64_t *rec = seg_table + 8 * (int64_t)seg;
            r_m = uniform64(rec[1]);
            r_tile0 = uniform64(rec[2]);
            r_codes = uniform64(rec[3]);
            r_levels = uniform64(rec[4]);
            r_lbub = uniform64(rec[5]);
            r_out = uniform64(rec[6]);
        }
        const int64_t m = r_m, sv0 = (tile - r_tile0) * 64;
        const int64_t code_off = r_codes + sv0, level_off = r_levels, lbub_off = r_lbub;
        float *const dst = out + r_out + sv0 * D;
        const int valid = (int)(m - sv0 < 64 ? m - sv0 : 64);      // subvectors of this tile that exist (>= 1)
        const int lsv = lane < valid ? lane : valid - 1;           // (a padding lane re-reads the last one; its units are not stored)
        f32x4 acc[UPS];
        for (int r0 = 0; r0 < R; r0 += DT_RCH) {
            const int nr = R - r0 < DT_RCH ? R - r0 : DT_RCH;
            // lane = subvector: this chunk's (code, norm) pairs into the wave's slots, DT_RB payloads' loads requested together
            // (payload by payload, a tile paid R round trips to memory in a row: 43.7 us at R = 8 where one payload takes 20.9)
            int rb = 0;
            for (; RB > 0 && rb + DT_RBK <= nr; rb += DT_RBK) {
                unsigned code_[DT_RBK];
                float lvl_[DT_RBK], lb_[DT_RBK], ub_[DT_RBK];
#pragma unroll
                for (int k = 0; k < DT_RBK; ++k) {
                    const uint8_t *p = gathered + (int64_t)(r0 + rb + k) * user_stride;
                    const float *lbub = reinterpret_cast<const float *>(p + lbub_off);
                    lb_[k] = lbub[0];
                    ub_[k] = lbub[1];
                    lvl_[k] = (float)reinterpret_cast<const LevelT *>(p + level_off)[sv0 + lsv];
                    code_[k] = p[code_off + lsv];
                }
#pragma unroll
                for (int k = 0; k < DT_RBK; ++k) {
                    const float lb = lb_[k], range = ub_[k] - lb;
                    float n = lvl_[k] * range;   // prob_scalar:31-32, unfused
                    n = n * inv_s;               // == / 2^n_bit exactly
                    n = n + lb;
                    *reinterpret_cast<uint2 *>(pairs + ((rb + k) * 64 + lane) * 2) = make_uint2(code_[k], __float_as_uint(n));
                }
            }
            for (; rb < nr; ++rb) {   // the payloads that do not fill a batch (all of them for R < DT_RB), one by one
                const uint8_t *p = gathered + (int64_t)(r0 + rb) * user_stride;
                const float *lbub = reinterpret_cast<const float *>(p + lbub_off);
                const float lb = lbub[0], range = lbub[1] - lb;
                float n = (float)reinterpret_cast<const LevelT *>(p + level_off)[sv0 + lsv] * range;
                n = n * inv_s;
                n = n + lb;
                const unsigned code = p[code_off + lsv];
                *reinterpret_cast<uint2 *>(pairs + (rb * 64 + lane) * 2) = make_uint2(code, __float_as_uint(n));
            }
            __builtin_amdgcn_wave_barrier();   // (written and read by this wave only: LDS operations of a wave stay in order)
            // lane = unit of a pass: products in payload order; DT_RB pairs, then their rows, are read together (one after the
            // other, every payload waited for two dependent LDS round trips)
#pragma unroll
            for (int pss = 0; pss < UPS; ++pss) {
                const int s = pss * (64 / UPS) + ls;
                auto one = [&](int rr, bool first) {
                    const uint2 cn = *reinterpret_cast<const uint2 *>(pairs + (rr * 64 + s) * 2);
                    const f32x4 c = *reinterpret_cast<const f32x4 *>(s_cb + cn.x * RS + 4 * q);
                    const float n = __uint_as_float(cn.y);
                    const f32x4 n4 = {n, n, n, n};
                    const f32x4 dec = c * n4;
                    if (first) {   // (wave-uniform: a scalar branch, not four selects per payload)
                        acc[pss] = dec;
                    } else {
                        acc[pss] = acc[pss] + dec;
                    }
                };
                int rr = 0;
                for (; RB > 0 && rr + DT_RBK <= nr; rr += DT_RBK) {
                    uint2 cn[DT_RBK];
                    f32x4 c[DT_RBK];
#pragma unroll
                    for (int k = 0; k < DT_RBK; ++k) cn[k] = *reinterpret_cast<const uint2 *>(pairs + ((rr + k) * 64 + s) * 2);
#pragma unroll
                    for (int k = 0; k < DT_RBK; ++k) c[k] = *reinterpret_cast<const f32x4 *>(s_cb + cn[k].x * RS + 4 * q);
                    const bool first = r0 == 0 && rr == 0;   // (wave-uniform)
#pragma unroll
                    for (int k = 0; k < DT_RBK; ++k) {
                        const float n = __uint_as_float(cn[k].y);
                        const f32x4 n4 = {n, n, n, n};
                        const f32x4 dec = c[k] * n4;
                        if (k == 0 && first) {
                            acc[pss] = dec;
                        } else {
                            acc[pss] = acc[pss] + dec;
                        }
                    }
                }
                for (; rr < nr; ++rr) one(rr, r0 == 0 && rr == 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int pss = 0; pss < UPS; ++pss) {
            const int s = pss * (64 / UPS) + ls;
            f32x4 v = acc[pss];
            if (md.apply) {
                v[0] = mean_div(v[0], md);
                v[1] = mean_div(v[1], md);
                v[2] = mean_div(v[2], md);
                v[3] = mean_div(v[3], md);
            }
            if (s < valid) *reinterpret_cast<f32x4 *>(dst + (pss * 64 + lane) * 4) = v;
        }
    }
}

// Error-feedback level launch, a WAVE per tile (K = 256, D = 8 / 16 / 32, byte or 16-bit levels): the level quantiser
// (probabilistic_scalar_compressor.py:12-27) and  error = v - codebook[code] * (l (ub - lb) / 2^n + lb)  (ps_quantizer.py:37-39) for
// every tensor in one pass.  Round 4's kernels ran one thread per (subvector, 16-byte unit): every thread looked up its
// tensor's record and (lb, ub) and worked the level out again (an IEEE division and, with stochastic rounding, a hash per
// thread: four times per subvector for D = 16) -- 42.5 us for the ResNet-50 list (195 MB: 4.6 TB/s).  Here lane l is
// subvector l for the level -- once per subvector, stored as one coalesced 64 / 128 bytes per wave, (code, norm) parked in the
// wave's LDS slots -- and unit l of each of the D / 4 passes for the residual: v and error move as contiguous kilobytes.
//
// OUT (round 5: the one-rank step's twin of gq_hsq_levels_decode): the same pass also DECODES the payload it has just finished
// -- out = (+0 + codebook[code] * norm) / 1, the aggregate of one user (plain: the decompress as it is) -- so that a step of one
// rank and one user is two kernels, encode and this one.  What the aggregate's launch would have taken along rides here too:
// the uncompressed tensors' mean (of one row: straight from the gradients, next to their copy into the wire), and -- by the
// LAST workgroup to finish, told by a ticket counter, because every workgroup reads the draws' words and the accumulators
// first -- the step of the { seed, step } words and the accumulators' reset.
template <int D, typename LevelT, bool OUT>
__global__ __launch_bounds__(DT_THREADS) void hsq_levels_ef_tile_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const float *__restrict__ u_flat, const unsigned *seg_minmax, int n_bit, int random_mode,      // (seg_minmax: NOT restrict -- OUT's last workgroup rewrites these words through ft.reset_dst)
    uint64_t seed, const float *__restrict__ r_flat, const float *__restrict__ cb, int K, uint8_t *__restrict__ wire,
    const int64_t *__restrict__ dense_table, int ndense, int write_error, float *__restrict__ out, int plain, const FusedTail ft) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    if (OUT && ft.dense_mean) {   // the mean of ONE row: (+0 + x) / 1 (plain: x), from the gradients themselves
        typedef const float __attribute__((address_space(1))) *gcf_ptr;
        for (int t = blockIdx.x; t < ndense; t += gridDim.x) {
            const gcf_ptr src = (gcf_ptr)(uintptr_t)dense_table[3 * t];
            float *dst = ft.dense_mean + (dense_table[3 * t + 1] - ft.dense_off) / 4;
            const int64_t n = dense_table[3 * t + 2];
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = plain ? src[i] : src[i] + 0.0f;
        }
    }
    const MeanDiv md = mean_div_of(1, !plain);
    constexpr int UPS = D / 4;
    constexpr int RS = ((D / 4) & 1) ? D : D + 4;
    __shared__ __attribute__((aligned(16))) float s_cb[256 * RS];
    __shared__ __attribute__((aligned(8))) unsigned s_pair[DT_WAVES * 64 * 2];   // { code, bits of the norm } per (wave, subvector)
    for (int i = threadIdx.x; i < K * UPS; i += DT_THREADS)   // (K <= 256 rows; codes stay below K)
        *reinterpret_cast<f32x4 *>(s_cb + (i / UPS) * RS + 4 * (i % UPS)) = reinterpret_cast<const f32x4 *>(cb)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float s = (float)(1 << n_bit);
    auto uniform64 = [](int64_t v) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    unsigned *const pairs = s_pair + wave * (64 * 2);
    const int q = lane % UPS, ls = lane / UPS;
    // A tile's chain of dependent round trips was  tile -> tensor  ->  the tensor's record  ->  (lb, ub), u, code: three or four
    // latencies per tile and wave.  The record stays in scalar registers while a wave's tiles stay with one tensor (they do,
    // thousands of tiles long), and the next tile's tensor word and u are requested while the current tile is worked on.
    const int64_t tstride = (int64_t)gridDim.x * DT_WAVES;
    int64_t tile = (int64_t)blockIdx.x * DT_WAVES + wave;
    int seg = -1;
    int64_t r_grad = 0, r_m = 0, r_tile0 = 0, r_codes = 0, r_levels = 0, r_lbub = 0, r_out = 0, r_err = 0;
    float lb = 0.0f, ub = 0.0f;
    int seg_nv = tile < ntiles ? tile_seg[tile] : 0;                    // (vector registers until they are needed)
    float u_nv = tile < ntiles ? u_flat[tile * 64 + lane] : 0.0f;       // u_flat is padded to whole tiles
    for (; tile < ntiles; tile += tstride) {
        const int seg_t = __builtin_amdgcn_readfirstlane(seg_nv);
        const float u_t = u_nv;
        if (seg_t != seg) {   // (wave-uniform) a new tensor: its record and its (lb, ub)
            seg = seg_t;
            const int64_t *rec = seg_table + 8 * (int64_t)seg;
            r_grad = uniform64(rec[0]);
            r_m = uniform64(rec[1]);
            r_tile0 = uniform64(rec[2]);
            r_codes = uniform64(rec[3]);
            r_levels = uniform64(rec[4]);
            r_lbub = uniform64(rec[5]);
            r_out = uniform64(rec[6]);
            r_err = uniform64(rec[7]);
            lb = order_unmap_f(seg_minmax[2 * seg]);
            ub = order_unmap_f(seg_minmax[2 * seg + 1]);
        }
        const int64_t m = r_m, sv0 = (tile - r_tile0) * 64;
        const int valid = (int)(m - sv0 < 64 ? m - sv0 : 64);
        float *const err = write_error ? reinterpret_cast<float *>(r_err) : nullptr;
        // this tile's code (needed after the level) and the next tile's words, requested before the level arithmetic
        const unsigned code = (OUT || err) ? wire[r_codes + sv0 + (lane < valid ? lane : valid - 1)] : 0u;
        if (tile + tstride < ntiles) {
            seg_nv = tile_seg[tile + tstride];
            u_nv = u_flat[(tile + tstride) * 64 + lane];
        }
        const LevelQuant lq(lb, ub, n_bit, random_mode == GQ_RANDOM_DEVICE_KEYED ? GQ_RANDOM_DEVICE : random_mode, r_flat,
                            random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, lb, ub) : seed);
        const int64_t flat = tile * 64 + lane;            // index into u_flat / r_flat and of this subvector's draw
        int l = 0;
        if (lane < valid) {
            l = lq.level(u_t, flat);
            reinterpret_cast<LevelT *>(wire + r_levels)[sv0 + lane] = (LevelT)l;
            if (sv0 == 0 && lane == 0) {
                float *lbub = reinterpret_cast<float *>(wire + r_lbub);
                lbub[0] = lb;
                lbub[1] = ub;
            }
        }
        if (!OUT && !err) continue;                        // (wave-uniform: a tensor without an error buffer gets its levels only)
        const float range = ub - lb;
        float n = (float)l * range;   // prob_scalar:31-32, unfused
        n = n / s;
        n = n + lb;
        *reinterpret_cast<uint2 *>(pairs + lane * 2) = make_uint2(code, __float_as_uint(n));
        __builtin_amdgcn_wave_barrier();   // (written and read by this wave only: LDS operations of a wave stay in order)
        const float *const grad = reinterpret_cast<const float *>(r_grad) + sv0 * D;
        float *const edst = err + sv0 * D;
        float *const odst = OUT ? out + r_out + sv0 * D : nullptr;
#pragma unroll
        for (int pss = 0; pss < UPS; ++pss) {
            const int sv = pss * (64 / UPS) + ls;
            if (sv < valid) {
                const uint2 cn = *reinterpret_cast<const uint2 *>(pairs + sv * 2);
                const f32x4 c = *reinterpret_cast<const f32x4 *>(s_cb + cn.x * RS + 4 * q);
                const float nn = __uint_as_float(cn.y);
                f32x4 dec;
                dec[0] = c[0] * nn;
                dec[1] = c[1] * nn;
                dec[2] = c[2] * nn;
                dec[3] = c[3] * nn;
                if (err) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(grad + (pss * 64 + lane) * 4);
                    f32x4 e;
                    e[0] = v[0] - dec[0];
                    e[1] = v[1] - dec[1];
                    e[2] = v[2] - dec[2];
                    e[3] = v[3] - dec[3];
                    *reinterpret_cast<f32x4 *>(edst + (pss * 64 + lane) * 4) = e;
                }
                if (OUT) {
                    if (md.apply) {
                        dec[0] = mean_div(dec[0], md);
                        dec[1] = mean_div(dec[1], md);
                        dec[2] = mean_div(dec[2], md);
                        dec[3] = mean_div(dec[3], md);
                    }
                    *reinterpret_cast<f32x4 *>(odst + (pss * 64 + lane) * 4) = dec;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (OUT && ft.ticket && (ft.rng_state || ft.reset_words)) {
        // every workgroup has read the draws' words (resolve_seed) and the accumulators (lb, ub) by now: the last one to get
        // here steps the words and puts the accumulators back to their empty state for the next step's encode
        // (a returning atomic on ONE word serialises at ~90 per microsecond -- MI355X_MICROARCH.md, 'dequeue' -- and a persistent
        // grid ends all at once: 1,000+ workgroups on one counter were 33 us of a 56 us launch.  Sixteen shard counters, each
        // on a line of its own, and a top counter for the shards' last arrivers.)
        __shared__ int s_last;
        __syncthreads();
        // No fence: what must come first are this workgroup's READS of the words (their data has been used by now -- the
        // barrier above is behind every wave's last tile), what comes after the last add are the last workgroup's WRITES.
        // (A __threadfence() here -- buffer_wbl2: the XCD L2's dirty lines, i.e. the launch's own output, written back by
        // every workgroup -- made this a 57 us launch instead of 19.)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);      // compiler-only: no load of the words may be sunk or re-issued behind the adds below
        if (threadIdx.x == 0) {
            const unsigned shard = blockIdx.x & (GQ_TICKET_SHARDS - 1);
            const unsigned members = (gridDim.x - shard + GQ_TICKET_SHARDS - 1) / GQ_TICKET_SHARDS;      // workgroups b with b % 16 == shard
            int last = 0;
            if (atomicAdd(ft.ticket + GQ_TICKET_STRIDE * (1 + shard), 1u) == members - 1) {
                ft.ticket[GQ_TICKET_STRIDE * (1 + shard)] = 0u;
                const unsigned shards = gridDim.x < GQ_TICKET_SHARDS ? gridDim.x : GQ_TICKET_SHARDS;
                last = atomicAdd(ft.ticket, 1u) == shards - 1 ? 1 : 0;
            }
            s_last = last;
        }
        __syncthreads();
        if (s_last) {
            if (ft.rng_state && (int)threadIdx.x < ft.rng_pairs) ft.rng_state[2 * threadIdx.x + 1] += 1;
            for (int i = threadIdx.x; i < ft.reset_words; i += blockDim.x) ft.reset_dst[i] = ft.reset_src[i];
            if (threadIdx.x == 0) *ft.ticket = 0u;
        }
    }
}

// Any (d, K), any code / level width: decode-mean over a segment table, or -- ERR -- the error-feedback
// pass  error = grad - decoded  of the single wire just written (ps_quantizer.py:39; the tensors and their
// error buffers are columns 0 and 7 of the table, rows without an error buffer are skipped).
// A workgroup walks tiles (64 padded subvectors of one tensor); a thread owns VEC consecutive floats of a
// subvector (VEC = 4 when d % 4 == 0, else 1).  The codebook is staged in LDS when it fits 64 KiB and
// comes from L1/L2 otherwise.
template <typename CodeT, typename LevelT, int VEC, bool ERR>
__global__ __launch_bounds__(BT_THREADS) void hsq_decode_sum_batched_any_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ tile_seg, int64_t ntiles,
    const uint8_t *__restrict__ gathered, int64_t user_stride, int R, const float *__restrict__ cb, int d, int K,
    int n_bit, float *__restrict__ out, int cb_in_lds, int plain) {
    extern __shared__ __attribute__((aligned(16))) float s_cb_any[];
    const float *cbp = cb;
    if (cb_in_lds) {
        for (int i = threadIdx.x; i < K * d; i += BT_THREADS) s_cb_any[i] = cb[i];
        __syncthreads();
        cbp = s_cb_any;
    }
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const float inv_s = 1.0f / (float)(1u << n_bit);   // exact; n * inv_s == n / 2^n_bit
    const MeanDiv md = mean_div_of(R, !ERR && !plain);   // ERR: R = 1, the plain decode whose difference to grad is the residual
    const int ups = d / VEC, units = 64 * ups;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int seg = tile_seg[tile];
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int64_t sv0 = (tile - rec[2]) * 64;
        const int64_t m = rec[1];
        float *dst = ERR ? reinterpret_cast<float *>(rec[7]) : out + rec[6];
        if (ERR && !dst) continue;
        const float *grad = reinterpret_cast<const float *>(rec[0]);
        for (int unit = threadIdx.x; unit < units; unit += BT_THREADS) {
            const int sv = unit / ups, q = unit - sv * ups;
            const int64_t local = sv0 + sv;
            if (local >= m) continue;
            vec_t acc;
            for (int r = 0; r < R; ++r) {
                const uint8_t *p = gathered + (int64_t)r * user_stride;
                const float *lbub = reinterpret_cast<const float *>(p + rec[5]);
                const float lb = lbub[0], range = lbub[1] - lb;
                float n = (float)reinterpret_cast<const LevelT *>(p + rec[4])[local];
                if constexpr (!std::is_same<LevelT, float>::value) {
                    n = n * range;   // prob_scalar:31-32, unfused
                    n = n * inv_s;
                    n = n + lb;
                }
                const int64_t code = (int64_t)reinterpret_cast<const CodeT *>(p + rec[3])[local];
                const vec_t c = *reinterpret_cast<const vec_t *>(cbp + code * d + VEC * q);
                const vec_t dec = c * n;
                if (r == 0) {
                    acc = dec;
                } else {
                    acc = acc + dec;
                }
            }
            if (md.apply) acc = mean_div(acc, md);
            const int64_t at = local * d + VEC * q;
            if (ERR) acc = *reinterpret_cast<const vec_t *>(grad + at) - acc;
            *reinterpret_cast<vec_t *>(dst + at) = acc;
        }
    }
}

template <typename CodeT, typename LevelT, bool ERR>
static int launch_decode_any(const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles, const uint8_t *gathered,
                             int64_t user_stride, int R, const float *cb, int d, int K, int n_bit, float *out,
                             hipStream_t st, const char *what, int plain_arg) {
    const int plain = (!ERR && (plain_arg & 1)) ? 1 : 0;   // the ring's hop / a round trip, not the aggregate (bit 1: GQ_AGGREGATE_FMA, not served here)
    const size_t cb_bytes = (size_t)K * d * sizeof(float);
    const int in_lds = cb_bytes <= 64 * 1024;
    const size_t lds = in_lds ? cb_bytes : 0;
    int64_t blocks = ntiles;
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    if ((d & 3) == 0) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_batched_any_kernel<CodeT, LevelT, 4, ERR>),
                           dim3((unsigned)blocks), dim3(BT_THREADS), lds, st, seg_table, tile_seg, ntiles, gathered,
                           user_stride, R, cb, d, K, n_bit, out, in_lds, plain);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_batched_any_kernel<CodeT, LevelT, 1, ERR>),
                           dim3((unsigned)blocks), dim3(BT_THREADS), lds, st, seg_table, tile_seg, ntiles, gathered,
                           user_stride, R, cb, d, K, n_bit, out, in_lds, plain);
    }
    GQ_CHECK_LAUNCH(what);
    return GQ_OK;
}

template <bool ERR>
static int decode_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                      const uint8_t *gathered, int64_t user_stride, int R, const float *cb, int d, int K,
                      int code_bytes, int level_bytes, int n_bit, float *out, void *stream, const char *what, int plain) {
    if (nseg < 1 || ntiles < 1 || R < 1 || d < 1 || K < 1 || n_bit < 1 || (n_bit > 30 && level_bytes != 0))
        return fail(GQ_ERR_INVALID_ARG, "%s: bad sizes", what);
    if (level_bytes == 0) n_bit = 1;   // f32 norms: no level scaling
    if (!seg_table || !tile_seg || !gathered || !cb || (!ERR && !out))
        return fail(GQ_ERR_INVALID_ARG, "%s: null pointer", what);
    hipStream_t st = as_stream(stream);
#define GQ_ANY_CASE(CB, LB, CT, LT)                                                                                  \
    if (code_bytes == CB && level_bytes == LB)                                                                      \
        return launch_decode_any<CT, LT, ERR>(seg_table, tile_seg, ntiles, gathered, user_stride, R, cb, d, K, n_bit, \
                                              out, st, what, plain);
    GQ_ANY_CASE(1, 0, uint8_t, float)
    GQ_ANY_CASE(4, 0, int32_t, float)
    GQ_ANY_CASE(1, 1, uint8_t, uint8_t)
    GQ_ANY_CASE(1, 2, uint8_t, int16_t)
    GQ_ANY_CASE(1, 4, uint8_t, int32_t)
    GQ_ANY_CASE(4, 1, int32_t, uint8_t)
    GQ_ANY_CASE(4, 2, int32_t, int16_t)
    GQ_ANY_CASE(4, 4, int32_t, int32_t)
#undef GQ_ANY_CASE
    return fail(GQ_ERR_INVALID_ARG, "%s: code_bytes must be 1 or 4, level_bytes 0 (f32 norms), 1, 2 or 4", what);
}

static inline int64_t bt_grid(int64_t items) {
    int64_t blocks = (items + BT_THREADS - 1) / BT_THREADS;
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

}  // namespace gq

namespace gq {
template <int D, typename LevelT, bool OUT = false>
static void launch_levels_ef_tile(const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles, const float *u_flat,
                                  const uint32_t *seg_minmax, int n_bit, int random_mode, uint64_t seed, const float *r_flat,
                                  const float *cb, int K, uint8_t *wire, const int64_t *dense_table, int ndense, hipStream_t st,
                                  int write_error = 1, float *out = nullptr, int plain = 0, const FusedTail &ft = FusedTail{}) {
    static const int bpc = resident_blocks_per_cu(hsq_levels_ef_tile_kernel<D, LevelT, OUT>, DT_THREADS, 0);
    int64_t blocks = (ntiles + DT_WAVES - 1) / DT_WAVES;
    const int64_t cap = (int64_t)cu_count() * bpc;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_levels_ef_tile_kernel<D, LevelT, OUT>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(DT_THREADS), 0,
                       st, seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, cb, K, wire, dense_table, ndense,
                       write_error, out, plain, ft);
}
}  // namespace gq

GQ_INTERNAL int gqi_hsq_levels_batched_d16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                           const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                           uint64_t seed, const float *r_flat, const float *ef_codebook, int K, int packed6,
                                           uint8_t *wire, const int64_t *dense_table, int ndense, void *stream) {
    // ef_codebook != NULL: additionally error = v - decode(wire) for the rows that have an error buffer (d = 16)
    if (nseg < 1 || ntiles < 1 || n_bit < 1 || n_bit > 8)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: bad sizes");
    if (!seg_table || !tile_seg || !u_flat || !seg_minmax || !wire)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: null pointer");
    if (random_mode == GQ_RANDOM_GIVEN && !r_flat)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: GQ_RANDOM_GIVEN needs r_flat");
    if (((int64_t)1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0) > (packed6 ? 63 : 255))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: levels do not fit %s", packed6 ? "6 bits" : "uint8");
    const dim3 block(gq::BT_THREADS);
    hipStream_t st = gq::as_stream(stream);
    if (ef_codebook && packed6)
        hipLaunchKernelGGL(gq::hsq_levels_ef_batched_kernel<true>, dim3((unsigned)gq::bt_grid(ntiles * 256)), block, 0, st,
                           seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, ef_codebook, K, wire, dense_table, ndense);
    else if (ef_codebook && GQ_EF_LEVELS_TILE)
        gq::launch_levels_ef_tile<16, uint8_t>(seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, ef_codebook, K, wire,
                                               dense_table, ndense, st);
    else if (ef_codebook)
        hipLaunchKernelGGL(gq::hsq_levels_ef_batched_kernel<false>, dim3((unsigned)gq::bt_grid(ntiles * 256)), block, 0, st,
                           seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, ef_codebook, K, wire, dense_table, ndense);
    else if (packed6)
        hipLaunchKernelGGL(gq::hsq_levels_batched_kernel<gq::Packed6>, dim3((unsigned)gq::bt_grid(ntiles * 16)), block, 0, st,
                           seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, wire, dense_table, ndense);
    else
        hipLaunchKernelGGL(gq::hsq_levels_batched_kernel<uint8_t>, dim3((unsigned)gq::bt_grid(ntiles * 16)), block, 0, st,
                           seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, wire, dense_table, ndense);
    GQ_CHECK_LAUNCH("gq_hsq_levels_batched");
    return GQ_OK;
}

GQ_INTERNAL int gqi_hsq_decode_sum_batched_d16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                               const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                               const float *codebook, int K, int n_bit, int packed6, float *out, int plain,
                                               const gq::StepTail *tail_or_null, int *tail_taken, void *stream) {
    if (tail_taken) *tail_taken = 0;
    if (nseg < 1 || ntiles < 1 || R < 1 || n_bit < 1 || n_bit > 8)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum_batched: bad sizes");
    if (!seg_table || !tile_seg || !gathered || !codebook || !out)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum_batched: null pointer");
    const bool fma = (plain & 2) != 0 && R >= 2;   // GQ_AGGREGATE_FMA >> 7 in the flags word
    plain = (plain & 1) ? 1 : 0;
    if ((user_stride_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(gathered) & 3) == 0) {
        // compile-time-R kernels up to BT4_RMAX payloads, the chunked one above: every R is served
        const gq::StepTail tail = tail_or_null ? *tail_or_null : gq::StepTail{};
        if (packed6)
            gq::launch_bt4_fixed_r<true>(R, seg_table, tile_seg, ntiles, gathered, user_stride_bytes, codebook, K, n_bit, out, plain,
                                         gq::as_stream(stream), fma, tail);
        else
            gq::launch_bt4_fixed_r<false>(R, seg_table, tile_seg, ntiles, gathered, user_stride_bytes, codebook, K, n_bit, out, plain,
                                          gq::as_stream(stream), fma, tail);
        if (tail_taken) *tail_taken = 1;
    } else if (packed6) {
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_decode_sum_batched: packed levels need 4-byte aligned wires");
    } else {
        hipLaunchKernelGGL(gq::hsq_decode_sum_batched_kernel, dim3((unsigned)gq::bt_grid(ntiles * 256)),
                           dim3(gq::BT_THREADS), 0, gq::as_stream(stream), seg_table, tile_seg, ntiles, gathered,
                           user_stride_bytes, R, codebook, K, n_bit, out, plain);
    }
    GQ_CHECK_LAUNCH("gq_hsq_decode_sum_batched");
    return GQ_OK;
}

// K = 256, d = 8 / 16 / 32, byte codes, byte or 16-bit levels: a wave per tile (hsq_decode_sum_batched_tile_kernel)
namespace gq {
template <int D, typename LevelT>
static void launch_decode_tile(const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles, const uint8_t *gathered,
                               int64_t user_stride_bytes, int R, const float *codebook, int K, int n_bit, float *out, int plain,
                               hipStream_t st, const StepTail &tail) {
    int64_t blocks = (ntiles + DT_WAVES - 1) / DT_WAVES;
#define GQ_DT_LAUNCH(RBV)                                                                                                        \
    do {                                                                                                                         \
        static const int bpc = resident_blocks_per_cu(hsq_decode_sum_batched_tile_kernel<D, LevelT, RBV>, DT_THREADS, 0);        \
        const int64_t cap = (int64_t)cu_count() * bpc;                                                                           \
        if (blocks > cap) blocks = cap;                                                                                          \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_batched_tile_kernel<D, LevelT, RBV>),                                  \
                           dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(DT_THREADS), 0, st, seg_table, tile_seg, ntiles,      \
                           gathered, user_stride_bytes, R, codebook, K, n_bit, out, plain, tail);                                   \
    } while (0)
    if (R >= DT_RB) GQ_DT_LAUNCH(DT_RB);
    else if (R >= 2) GQ_DT_LAUNCH(2);
    else GQ_DT_LAUNCH(0);
#undef GQ_DT_LAUNCH
}
}  // namespace gq

GQ_INTERNAL int gqi_hsq_decode_sum_batched_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                             const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                             const float *codebook, int d, int K, int level_bytes, int n_bit, float *out, int plain,
                                             const gq::StepTail *tail_or_null, int *tail_taken, void *stream) {
    const gq::StepTail tail = tail_or_null ? *tail_or_null : gq::StepTail{};
    if (tail_taken) *tail_taken = 0;
    if (nseg < 1 || ntiles < 1 || R < 1 || n_bit < 1 || n_bit > (level_bytes == 1 ? 8 : 15))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum_batched: bad sizes");
    if (!seg_table || !tile_seg || !gathered || !codebook || !out)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum_batched: null pointer");
    plain = (plain & 1) ? 1 : 0;
    hipStream_t st = gq::as_stream(stream);
#define GQ_DT_CASE(DD, LB, LT)                                                                                          \
    if (d == DD && level_bytes == LB) {                                                                                 \
        gq::launch_decode_tile<DD, LT>(seg_table, tile_seg, ntiles, gathered, user_stride_bytes, R, codebook, K, n_bit, out, \
                                       plain, st, tail);                                                                \
        GQ_CHECK_LAUNCH("gq_hsq_decode_sum_batched");                                                                   \
        if (tail_taken) *tail_taken = 1;                                                                                \
        return GQ_OK;                                                                                                   \
    }
    GQ_DT_CASE(32, 1, uint8_t)
    GQ_DT_CASE(32, 2, uint16_t)
    GQ_DT_CASE(8, 1, uint8_t)
    GQ_DT_CASE(8, 2, uint16_t)
    GQ_DT_CASE(16, 2, uint16_t)
#undef GQ_DT_CASE
    return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_decode_sum_batched: this path serves d = 8 / 32 (byte or 16-bit levels) and d = 16 (16-bit levels), K = 256");
}

// error-feedback level kernel of d = 8 / 32 (K = 256): levels + error = v - decode(wire) in one launch
GQ_INTERNAL int gqi_hsq_levels_batched_ef_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                            const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                            uint64_t seed, const float *r_flat, const float *codebook, int d, int K, uint8_t *wire, const int64_t *dense_table, int ndense,
                                            void *stream) {
    if (nseg < 1 || ntiles < 1 || n_bit < 1 || n_bit > 8)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: bad sizes");
    if (!seg_table || !tile_seg || !u_flat || !seg_minmax || !codebook || !wire)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: null pointer");
    if (random_mode == GQ_RANDOM_GIVEN && !r_flat)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: GQ_RANDOM_GIVEN needs r_flat");
    if (((int64_t)1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0) > 255)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: levels do not fit uint8");
    if (GQ_EF_LEVELS_TILE && (d == 8 || d == 32)) {
        if (d == 8)
            gq::launch_levels_ef_tile<8, uint8_t>(seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, codebook, K, wire,
                                                  dense_table, ndense, gq::as_stream(stream));
        else
            gq::launch_levels_ef_tile<32, uint8_t>(seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, codebook, K, wire,
                                                   dense_table, ndense, gq::as_stream(stream));
    } else if (d == 8) {
        hipLaunchKernelGGL(gq::hsq_levels_ef_batched_d_kernel<8>, dim3((unsigned)gq::bt_grid(ntiles * 64 * 2)),
                           dim3(gq::BT_THREADS), 0, gq::as_stream(stream), seg_table, tile_seg, ntiles, u_flat, seg_minmax,
                           n_bit, random_mode, seed, r_flat, codebook, K, wire, dense_table, ndense);
    } else if (d == 32) {
        hipLaunchKernelGGL(gq::hsq_levels_ef_batched_d_kernel<32>, dim3((unsigned)gq::bt_grid(ntiles * 64 * 8)),
                           dim3(gq::BT_THREADS), 0, gq::as_stream(stream), seg_table, tile_seg, ntiles, u_flat, seg_minmax,
                           n_bit, random_mode, seed, r_flat, codebook, K, wire, dense_table, ndense);
    } else {
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_levels_batched: the fused error-feedback form serves d = 8, 16 or 32 (K = 256)");
    }
    GQ_CHECK_LAUNCH("gq_hsq_levels_batched");
    return GQ_OK;
}

GQ_INTERNAL int gqi_hsq_levels_batched_ef16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                            const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                            uint64_t seed, const float *r_flat, const float *codebook, int d, int K, uint8_t *wire,
                                            const int64_t *dense_table, int ndense, void *stream) {
    if (nseg < 1 || ntiles < 1 || n_bit < 1 || n_bit > 15)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: bad sizes");
    if (!seg_table || !tile_seg || !u_flat || !seg_minmax || !codebook || !wire)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: null pointer");
    if (random_mode == GQ_RANDOM_GIVEN && !r_flat)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: GQ_RANDOM_GIVEN needs r_flat");
    hipStream_t st = gq::as_stream(stream);
#define GQ_EF16_CASE(DD)                                                                                                       \
    if (d == DD) {                                                                                                             \
        gq::launch_levels_ef_tile<DD, uint16_t>(seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat, \
                                                codebook, K, wire, dense_table, ndense, st);                                      \
        GQ_CHECK_LAUNCH("gq_hsq_levels_batched");                                                                              \
        return GQ_OK;                                                                                                          \
    }
    GQ_EF16_CASE(32)
    GQ_EF16_CASE(16)
    GQ_EF16_CASE(8)
#undef GQ_EF16_CASE
    return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_levels_batched: d must be 8, 16 or 32 (K = 256)");
}

// levels (+ residual) + decode of the finished payload + the step's tail in ONE launch (K = 256, d = 8 / 16 / 32, byte or 16-bit levels)
GQ_INTERNAL int gqi_hsq_levels_decode_batched(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                              const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                              uint64_t seed, const float *r_flat, const float *codebook, int d, int K, int level_bytes,
                                              uint8_t *wire, const int64_t *dense_table, int ndense, int write_error, float *out,
                                              int plain, const gq::FusedTail *ft, void *stream) {
    if (nseg < 1 || ntiles < 1 || n_bit < 1 || n_bit > (level_bytes == 1 ? 8 : 15))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: bad sizes");
    if (!seg_table || !tile_seg || !u_flat || !seg_minmax || !codebook || !wire || !out)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: null pointer");
    if (random_mode == GQ_RANDOM_GIVEN && !r_flat)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: GQ_RANDOM_GIVEN needs r_flat");
    if (level_bytes == 1 && ((int64_t)1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0) > 255)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: levels do not fit uint8");
    hipStream_t st = gq::as_stream(stream);
    const gq::FusedTail tail = ft ? *ft : gq::FusedTail{};
#define GQ_LD_CASE(DD, LB, LT)                                                                                                       \
    if (d == DD && level_bytes == LB) {                                                                                              \
        gq::launch_levels_ef_tile<DD, LT, true>(seg_table, tile_seg, ntiles, u_flat, seg_minmax, n_bit, random_mode, seed, r_flat,    \
                                                codebook, K, wire, dense_table, ndense, st, write_error, out, plain & 1, tail);         \
        GQ_CHECK_LAUNCH("gq_hsq_levels_decode_batched");                                                                             \
        return GQ_OK;                                                                                                                \
    }
    GQ_LD_CASE(16, 1, uint8_t)
    GQ_LD_CASE(32, 1, uint8_t)
    GQ_LD_CASE(32, 2, uint16_t)
    GQ_LD_CASE(8, 1, uint8_t)
    GQ_LD_CASE(8, 2, uint16_t)
    GQ_LD_CASE(16, 2, uint16_t)
#undef GQ_LD_CASE
    return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_levels_decode_batched: d must be 8, 16 or 32 (K = 256), byte or 16-bit levels");
}

// any level width (1 / 2 / 4 bytes, or 0: the f32 projections travel); independent of (d, K)
GQ_INTERNAL int gqi_hsq_levels_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                           const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                           uint64_t seed, const float *r_flat, int level_bytes, uint8_t *wire, const int64_t *dense_table, int ndense, void *stream) {
    if (nseg < 1 || ntiles < 1 || n_bit < 1 || (n_bit > 30 && level_bytes != 0))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: bad sizes");
    if (!seg_table || !tile_seg || !u_flat || !seg_minmax || !wire)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: null pointer");
    if (random_mode == GQ_RANDOM_GIVEN && !r_flat && level_bytes != 0)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: GQ_RANDOM_GIVEN needs r_flat");
    if (level_bytes == 0) {   // n_bit == 32 (nearest_neighbor_compressor.py:14,75-76): u itself is the payload
        hipLaunchKernelGGL(gq::hsq_levels_batched_kernel<float>, dim3((unsigned)gq::bt_grid(ntiles * 16)),
                           dim3(gq::BT_THREADS), 0, gq::as_stream(stream), seg_table, tile_seg, ntiles, u_flat, seg_minmax,
                           1, GQ_RANDOM_OFF, (uint64_t)0, (const float *)nullptr, wire, dense_table, ndense);
        GQ_CHECK_LAUNCH("gq_hsq_levels_batched");
        return GQ_OK;
    }
    const int64_t top = ((int64_t)1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0);
    if ((level_bytes == 1 && top > 255) || (level_bytes == 2 && top > 32767))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: levels up to %lld do not fit %d byte(s)",
                        (long long)top, level_bytes);
    const dim3 grid((unsigned)gq::bt_grid(ntiles * 16)), block(gq::BT_THREADS);
    hipStream_t st = gq::as_stream(stream);
    if (level_bytes == 1)
        hipLaunchKernelGGL(gq::hsq_levels_batched_kernel<uint8_t>, grid, block, 0, st, seg_table, tile_seg, ntiles, u_flat,
                           seg_minmax, n_bit, random_mode, seed, r_flat, wire, dense_table, ndense);
    else if (level_bytes == 2)
        hipLaunchKernelGGL(gq::hsq_levels_batched_kernel<int16_t>, grid, block, 0, st, seg_table, tile_seg, ntiles, u_flat,
                           seg_minmax, n_bit, random_mode, seed, r_flat, wire, dense_table, ndense);
    else if (level_bytes == 4)
        hipLaunchKernelGGL(gq::hsq_levels_batched_kernel<int32_t>, grid, block, 0, st, seg_table, tile_seg, ntiles, u_flat,
                           seg_minmax, n_bit, random_mode, seed, r_flat, wire, dense_table, ndense);
    else
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: level_bytes must be 0, 1, 2 or 4");
    GQ_CHECK_LAUNCH("gq_hsq_levels_batched");
    return GQ_OK;
}

GQ_INTERNAL int gqi_hsq_decode_sum_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                               const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                               const float *codebook, int d, int K, int code_bytes, int level_bytes,
                                               int n_bit, float *out, int plain, void *stream) {
    return gq::decode_any<false>(seg_table, tile_seg, nseg, ntiles, gathered, user_stride_bytes, R, codebook, d, K,
                                 code_bytes, level_bytes, n_bit, out, stream, "gq_hsq_decode_sum_batched", plain);
}

// error = grad - decode(wire) for the rows that have an error buffer (ps_quantizer.py:39), any widths
GQ_INTERNAL int gqi_hsq_error_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                          const uint8_t *wire, const float *codebook, int d, int K, int code_bytes,
                                          int level_bytes, int n_bit, void *stream) {
    return gq::decode_any<true>(seg_table, tile_seg, nseg, ntiles, wire, 0, 1, codebook, d, K, code_bytes, level_bytes,
                                n_bit, nullptr, stream, "gq_hsq_levels_batched (error feedback)", 0);
}
