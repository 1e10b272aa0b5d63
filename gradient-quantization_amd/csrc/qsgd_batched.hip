// QSGD on a packed wire, multi-tensor (segment table) form -- BASELINE config 5.
//
// Same arithmetic as qsgd.hip (which mirrors the reference's signature: f32 norm per bucket,
// bool signs, int32 levels -- qsgd_compressor.py:42-71), but what is WRITTEN is a real wire format:
//     norm f32[buckets] | one code per element = sign<<(bits-1) | level ,  bits = 4 (n_bit <= 2), 8 (n_bit <= 6) or 16
// 4-bit codes are packed two per byte (element 2i in the low nibble).  ResNet-50 with c_dim=128,
// n_bit=2: 0.53 B per gradient element instead of the 2 B of separate sign / level arrays.
// One launch serves every tensor of a model: bucket_seg[bucket] names its tensor and
// seg_table[seg] = { grad ptr, d, first bucket, norm off, codes off (bytes, inside ONE user's wire),
// out off (floats), buckets, - }.  A zero bucket (0/0 = NaN level in the reference, decodes to 0)
// is written as level 0.  HBM-bound: 4 B read + 0.5..1 B written per element; one wave per bucket.
#include "gq_common.hpp"
#include <type_traits>

namespace gq {

constexpr int QB_THREADS = 256;
constexpr int QB_LDS_SEGS = 256;   // segment records kept in LDS by the 4-bit compress kernel (16 KiB)

// The draws of this file's kernels (GQ_RANDOM_DEVICE*: the library's own numbers, only their distribution is specified):
// element e of bucket b draws  u = top 24 bits of mix(key(seed, b) + e * phi) * 2^-24  with key = the library's three-round
// hash of (seed, b), taken ONCE per bucket and lane, and mix = one multiply-xorshift round.  The elements of a bucket walk a
// Weyl sequence through a bijective mixer; buckets and steps are separated by the full hash.  Round 5 ran the three-round hash
// (and a 64-bit index) per ELEMENT: ~20 of the ~46 vector instructions an element cost.
__device__ __forceinline__ uint32_t bucket_draw_key(uint64_t seed, int64_t b) { return uniform_bits(seed, (uint64_t)b); }
__device__ __forceinline__ float bucket_draw(uint32_t key, uint32_t e) {
    uint32_t h = key + e * 0x9E3779B1u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;                                   // (the top 24 bits of the product are its best-mixed ones)
    return (float)(h >> 8) * 5.9604644775390625e-08f;   // k * 2^-24, the grid torch.rand uses for float32
}

// qsgd_compressor.py:50-61 for one element: |v / norm| * s, clamp, truncate, stochastic round up; the sign above the level bits.
// FAST: x = RN(|v| / norm) * s from the bucket's ONE reciprocal by Markstein's correction (shared_quotient: the correctly rounded
// quotient, bit for bit what v_div_* gives, in three operations instead of ~11) -- taken of |v| and norm / s with the reciprocal
// y * s: s is a power of two, so RN(|v| / (norm / s)) IS RN(|v| / norm) * s and the multiplication by s goes too.  The caller has
// checked the operand window (quotient_window) for every element of the lane; then no NaN can occur and the quotient is >= 0:
// the NaN test and the lower clamp go as well.  The sign bit is clamp(bits(v), 0, 1) (v is finite there: > 0 iff its bits, as a
// signed integer, are).  FAST: norm_s = norm / s, y_s = RN(1 / norm) * s; otherwise norm_s = norm and y_s is not read.
// RND: 1 / 0 = the caller has tested random_mode once for all of a lane's elements, -1 = tested here.
template <bool FAST, int RND = -1>
__device__ __forceinline__ unsigned qsgd_code(float v, float norm_s, float y_s, float s, float smax, int random_mode, uint32_t key, uint32_t e,
                                              int bits) {
    const float x = FAST ? shared_quotient(fabsf(v), norm_s, y_s) : fabsf(v / norm_s) * s;
    unsigned l = 0, sgn;
    if constexpr (FAST) {
        // (as inline asm: the compiler turns min(max(bits, 0), 1) back into v_cmp + v_cndmask + v_or through VCC, with the
        // wait states gfx950 wants between a VALU write of VCC and its VALU read)
        asm("v_med3_i32 %0, %1, 0, 1" : "=v"(sgn) : "v"(__float_as_uint(v)));
    } else {
        sgn = v > 0.0f ? 1u : 0u;
    }
    if (!FAST && x != x) {
        // NaN (a zero bucket's 0 / 0, a NaN norm): the reference's cast makes it INT_MIN, a NEGATIVE level, and decodes
        // (-2^31) (2 sign - 1) norm / s (qsgd_compressor.py:53,69-70) -- for a zero bucket (-2^31)(-1)(0) = +0.  The wire's level
        // is 0 and the level's sign goes into the sign bit: a zero bucket decodes to +0 too (round 5 wrote sign 0: -0, which
        // only a bit-for-bit comparison of a PLAIN decode sees -- the aggregate starts from +0).
        sgn ^= 1u;
    } else {
        const float c = FAST ? fminf(x, smax) : fminf(fmaxf(x, 0.0f), smax);
        l = (unsigned)(int)c;
        if (RND == 1 || (RND == -1 && random_mode >= GQ_RANDOM_DEVICE)) {   // DEVICE, or DEVICE_KEYED with the bucket's keyed seed handed in
            const float prob = x - (float)l;
            l += (prob > bucket_draw(key, e)) ? 1u : 0u;
        }
    }
    return l | (sgn << (bits - 1));
}
// the operand window of the FAST form for a bucket norm and the smallest |v| of the lane's elements: shared_quotient needs
// 2^-80 <= norm / s <= 2^20 (s <= 2^16) and every |v| >= 2^-102 (gq_common.hpp).  A lane that holds an exact zero next to non-zero
// elements takes the division: rare outside all-zero buckets, whose norm is outside the window anyway.
__device__ __forceinline__ bool quotient_window(float norm, float min_abs) {
    return norm >= 0x1p-64f && norm <= 0x1p20f && min_abs >= 0x1p-102f;
}

// one wave per bucket; lane handles element pairs (2*lane, 2*lane+1), strided by 128.
// EF: error feedback fused around the codec (ps_quantizer.py:35-39): the bucket is read as
// v = grad + ef_scale*error (product rounded, then the add), v is written back over grad, and
// error = v - decode(code) replaces the old error -- all in this one pass.  seg_table[seg][7] is the
// tensor's error buffer (0 = none).
template <bool EF>
__global__ __launch_bounds__(QB_THREADS) void qsgd_compress_batched_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ bucket_seg, int64_t nbuckets, int n_bit,
    int bits, int random_mode, uint64_t seed, float ef_scale, uint8_t *__restrict__ wire, const int64_t *__restrict__ dense_table, int ndense) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (QB_THREADS / 64);
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    const unsigned lmask = (1u << (bits - 1)) - 1u;
    for (int64_t b = (int64_t)blockIdx.x * (QB_THREADS / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); b < nbuckets; b += nw) {
        const int seg = __builtin_amdgcn_readfirstlane(bucket_seg[b]);
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int d = (int)rec[1];
        const int64_t lb = b - rec[2];
        float *v = reinterpret_cast<float *>(rec[0]) + lb * d;
        float *err = (EF && rec[7]) ? reinterpret_cast<float *>(rec[7]) + lb * d : nullptr;
        auto load = [&](int e) {
            float2 p = *reinterpret_cast<const float2 *>(v + e);
            if (EF && err) {
                const float2 q = *reinterpret_cast<const float2 *>(err + e);
                const float p0 = ef_scale * q.x, p1 = ef_scale * q.y;
                p.x = p.x + p0;
                p.y = p.y + p1;
            }
            return p;
        };
        float mx = 0.0f;
        for (int e = 2 * lane; e < d; e += 128) {
            const float2 p = load(e);
            mx = absmax3_nan(mx, p.x, p.y);   // NaN-propagating, like torch.max (qsgd_compressor.py:49)
        }
        mx = wave_max_nan(mx);
        if (lane == 0) reinterpret_cast<float *>(wire + rec[3])[lb] = mx;
        const uint64_t sd = random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, mx, mx) : seed;   // keyed by the bucket's norm
        const uint32_t key = bucket_draw_key(sd, b);   // the draws' stream of this bucket (element index inside the bucket)
        uint8_t *dst = wire + rec[4] + ((lb * d * bits) >> 3);
        for (int e = 2 * lane; e < d; e += 128) {
            const float2 p = load(e);
            const unsigned c0 = qsgd_code<false>(p.x, mx, 0.0f, s, smax, random_mode, key, (uint32_t)e, bits);
            const unsigned c1 = qsgd_code<false>(p.y, mx, 0.0f, s, smax, random_mode, key, (uint32_t)e + 1u, bits);
            if (bits == 4) {
                dst[e >> 1] = (uint8_t)(c0 | (c1 << 4));
            } else if (bits == 8) {
                *reinterpret_cast<uchar2 *>(dst + e) = make_uchar2((uint8_t)c0, (uint8_t)c1);
            } else {
                *reinterpret_cast<unsigned *>(dst + 2 * e) = c0 | (c1 << 16);
            }
            if (EF && err) {
                // qsgd_compressor.py:69-70 on this element's own code, then ps_quantizer.py:39
                float t0 = (float)(c0 & lmask) * (2.0f * (float)(c0 >> (bits - 1)) - 1.0f);
                float t1 = (float)(c1 & lmask) * (2.0f * (float)(c1 >> (bits - 1)) - 1.0f);
                t0 = t0 * mx;
                t1 = t1 * mx;
                t0 = t0 / s;
                t1 = t1 / s;
                *reinterpret_cast<float2 *>(v + e) = p;
                *reinterpret_cast<float2 *>(err + e) = make_float2(p.x - t0, p.y - t1);
            }
        }
    }
}

// 4-bit wire, buckets of up to 256 elements (d % 8 == 0): 16 lanes per bucket (a wave takes four
// buckets), a lane holds 8 (16 for d > 128) consecutive elements in registers -- the bucket is read from
// HBM ONCE: max-abs over the lane's values, a 16-lane butterfly, then the codes straight from the
// registers as one dword per 8 elements.  EF as in the kernel above.  (The wave-per-bucket form read
// every bucket twice with 8-byte loads and spent most of its time in the 64-bit RNG: 71 us for the
// 23.5 M-element ResNet-50 list.)
// SEGLDS: the segment records come from an LDS copy (nseg <= QB_LDS_SEGS) and the bucket -> tensor word of the next
// item is fetched an item ahead: looked up in global memory, bucket -> tensor -> record -> data is three dependent
// round trips per item and the kernel was bound by that latency (41 us for the ResNet-50 list, the same with the
// division taken out).
// BITS: 4, 8 or 16 per code (round 5: the 8- and 16-bit wires, n_bit 3 ... 8, ran on the wave-per-bucket kernel above --
// 0.114 against 0.064 ms per ResNet-50 step).  A lane's 8 codes are one dword, two or four.
// LPB: lanes per bucket (16, 8, 4 or 2: a wave takes 4 ... 32 buckets).  A bucket of d elements keeps d / 8 lanes busy: at the
// reference's default --c-dim 32, 16 lanes per bucket left three quarters of every wave idle (0.158 ms per ResNet-50 step
// against 0.064 at c_dim 128).  The launcher picks LPB from the descriptor's bucket-width hint; any LPB is correct for any d.
template <bool EF, bool SEGLDS, int BITS = 4, int LPB = 16>
__global__ __launch_bounds__(QB_THREADS) void qsgd_compress_batched4_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ bucket_seg, int nseg, int64_t nbuckets, int n_bit,
    int random_mode, uint64_t seed, float ef_scale, uint8_t *__restrict__ wire, const int64_t *__restrict__ dense_table, int ndense) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    __shared__ int64_t s_seg[SEGLDS ? QB_LDS_SEGS * 8 : 1];
    if (SEGLDS) {
        for (int i = threadIdx.x; i < nseg * 8; i += QB_THREADS) s_seg[i] = seg_table[i];
        __syncthreads();
    }
    constexpr int BPW = 64 / LPB;   // buckets per wave
    const int lane = threadIdx.x & 63, sub = lane / LPB, c0 = lane & (LPB - 1);
    constexpr unsigned LMASK = (1u << (BITS - 1)) - 1u;   // level bits of a code; the sign sits above them
    constexpr int SB = BITS - 1;
    const int64_t nw = (int64_t)gridDim.x * (QB_THREADS / 64);
    const float s = (float)(1 << n_bit), smax = s - 1.0f, inv_s = 1.0f / s;
    const int64_t nquads = (nbuckets + BPW - 1) / BPW;
    int64_t qd = (int64_t)blockIdx.x * (QB_THREADS / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int seg_next = (qd < nquads && BPW * qd + sub < nbuckets) ? bucket_seg[BPW * qd + sub] : 0;
    for (; qd < nquads; qd += nw) {
        const int64_t b = BPW * qd + sub;
        const bool live = b < nbuckets;
        const int seg = seg_next;
        {
            const int64_t bn = BPW * (qd + nw) + sub;
            seg_next = (qd + nw < nquads && bn < nbuckets) ? bucket_seg[bn] : 0;   // the next item's tensor
        }
        int64_t recv[8];
        if constexpr (SEGLDS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) recv[i] = s_seg[8 * seg + i];
        } else {
            typedef const int64_t __attribute__((address_space(1))) *grec_ptr;
            const grec_ptr gr = (grec_ptr)(seg_table + 8 * (int64_t)seg);
#pragma unroll
            for (int i = 0; i < 8; ++i) recv[i] = gr[i];
        }
        const int64_t *rec = recv;
        const int d = live ? (int)rec[1] : 0;
        const int64_t lb = b - rec[2];
        // global address-space pointers: as plain pointers these would be flat (see hsq_encode_pf.hip)
        typedef float __attribute__((address_space(1))) *gf_ptr;
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef v2f __attribute__((address_space(1))) *gf2_ptr;
        typedef f32x4 __attribute__((address_space(1))) *gv_ptr;
        const gf_ptr v = (gf_ptr)(uintptr_t)rec[0] + lb * d;
        const gf_ptr err = (EF && rec[7]) ? (gf_ptr)(uintptr_t)rec[7] + lb * d : (gf_ptr)0;
        if (live && d > 16 * LPB && (d & 7) == 0) {
            // wider buckets of whole 8-element units: the bucket's lanes walk it twice, a unit (32 bytes in, one packed unit of
            // codes out) per lane and trip -- the arithmetic of the register path below, element for element.  (Before: the
            // element-pair walk further down, 8 bytes per lane and trip: c_dim 512 ran 0.089 ms per ResNet-50 step.)
            auto load8 = [&](int e, f32x4 &a, f32x4 &b2) {
                a = *(gv_ptr)(v + e);
                b2 = *(gv_ptr)(v + e + 4);
                if (EF && err) {
                    const f32x4 q0 = *(gv_ptr)(err + e);
                    const f32x4 q1 = *(gv_ptr)(err + e + 4);
                    a = a + q0 * ef_scale;   // product rounded, then the add (-ffp-contract=off)
                    b2 = b2 + q1 * ef_scale;
                }
            };
            float m2 = 0.0f, n2 = __builtin_inff();
            for (int c = c0; 8 * c < d; c += LPB) {
                f32x4 a, b2;
                load8(8 * c, a, b2);
                m2 = absmax3_nan(m2, a[0], a[1]);   // NaN-propagating, like torch.max
                m2 = absmax3_nan(m2, a[2], a[3]);
                m2 = absmax3_nan(m2, b2[0], b2[1]);
                m2 = absmax3_nan(m2, b2[2], b2[3]);
                n2 = fminf(fminf(n2, fabsf(a[0])), fabsf(a[1]));
                n2 = fminf(fminf(n2, fabsf(a[2])), fabsf(a[3]));
                n2 = fminf(fminf(n2, fabsf(b2[0])), fabsf(b2[1]));
                n2 = fminf(fminf(n2, fabsf(b2[2])), fabsf(b2[3]));
            }
#pragma unroll
            for (int o = LPB / 2; o > 0; o >>= 1) m2 = max_nan(m2, __shfl_xor(m2, o, 64));
            if (c0 == 0) reinterpret_cast<float *>(wire + rec[3])[lb] = m2;
            uint8_t *dst2 = wire + rec[4] + ((lb * d * BITS) >> 3);
            const uint64_t sd2 = random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, m2, m2) : seed;
            const uint32_t key2 = bucket_draw_key(sd2, b);
            const float y2 = 1.0f / m2;
            const bool fast2 = !EF && quotient_window(m2, n2);   // (EF: see `fast` below)
            for (int c = c0; 8 * c < d; c += LPB) {
                f32x4 xx[2];
                load8(8 * c, xx[0], xx[1]);
                unsigned code[8];
                f32x4 dec[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned cc = fast2 ? qsgd_code<true>(xx[k >> 2][k & 3], m2 * inv_s, y2 * s, s, smax, random_mode, key2, (uint32_t)(8 * c + k), BITS)
                                              : qsgd_code<false>(xx[k >> 2][k & 3], m2, 0.0f, s, smax, random_mode, key2, (uint32_t)(8 * c + k), BITS);
                    code[k] = cc;
                    if (EF) {
                        float t = __uint_as_float(__float_as_uint((float)(cc & LMASK)) | (((cc >> SB) ^ 1u) << 31));
                        t = t * m2;
                        dec[k >> 2][k & 3] = t * inv_s;
                    }
                }
                if (BITS == 4) {
                    unsigned word = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) word |= code[k] << (4 * k);
                    *reinterpret_cast<unsigned *>(dst2 + 4 * c) = word;
                } else if (BITS == 8) {
                    *reinterpret_cast<uint2 *>(dst2 + 8 * c) = make_uint2(code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24),
                                                                          code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24));
                } else {
                    *reinterpret_cast<uint4 *>(dst2 + 16 * c) = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16),
                                                                           code[4] | (code[5] << 16), code[6] | (code[7] << 16));
                }
                if (EF && err) {
                    *(gv_ptr)(v + 8 * c) = xx[0];
                    *(gv_ptr)(v + 8 * c + 4) = xx[1];
                    *(gv_ptr)(err + 8 * c) = xx[0] - dec[0];      // ps_quantizer.py:39
                    *(gv_ptr)(err + 8 * c + 4) = xx[1] - dec[1];
                }
            }
            continue;   // (the other buckets of this wave take the register path below on their own lanes)
        }
        if (live && (d > 16 * LPB || (d & 7) != 0)) {
            // other bucket widths: the 16 lanes walk the bucket twice, an element pair at a time
            auto load = [&](int e) {
                const v2f pv = *(gf2_ptr)(v + e);
                float2 p = make_float2(pv[0], pv[1]);
                if (EF && err) {
                    const v2f qv = *(gf2_ptr)(err + e);
                    const float2 q = make_float2(qv[0], qv[1]);
                    const float p0 = ef_scale * q.x, p1 = ef_scale * q.y;
                    p.x = p.x + p0;
                    p.y = p.y + p1;
                }
                return p;
            };
            float m2 = 0.0f;
            for (int e = 2 * c0; e < d; e += 2 * LPB) {
                const float2 p = load(e);
                m2 = absmax3_nan(m2, p.x, p.y);   // NaN-propagating, like torch.max
            }
#pragma unroll
            for (int o = LPB / 2; o > 0; o >>= 1) m2 = max_nan(m2, __shfl_xor(m2, o, 64));
            if (c0 == 0) reinterpret_cast<float *>(wire + rec[3])[lb] = m2;
            uint8_t *dst2 = wire + rec[4] + ((lb * d * BITS) >> 3);
            const uint64_t sd2 = random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, m2, m2) : seed;
            const uint32_t key2 = bucket_draw_key(sd2, b);
            for (int e = 2 * c0; e < d; e += 2 * LPB) {
                const float2 p = load(e);
                const unsigned k0 = qsgd_code<false>(p.x, m2, 0.0f, s, smax, random_mode, key2, (uint32_t)e, BITS);
                const unsigned k1 = qsgd_code<false>(p.y, m2, 0.0f, s, smax, random_mode, key2, (uint32_t)e + 1u, BITS);
                if (BITS == 4) {
                    dst2[e >> 1] = (uint8_t)(k0 | (k1 << 4));
                } else if (BITS == 8) {
                    dst2[e] = (uint8_t)k0;
                    dst2[e + 1] = (uint8_t)k1;
                } else {
                    reinterpret_cast<uint16_t *>(dst2)[e] = (uint16_t)k0;
                    reinterpret_cast<uint16_t *>(dst2)[e + 1] = (uint16_t)k1;
                }
                if (EF && err) {
                    float t0 = (float)(k0 & LMASK) * (2.0f * (float)(k0 >> SB) - 1.0f);
                    float t1 = (float)(k1 & LMASK) * (2.0f * (float)(k1 >> SB) - 1.0f);
                    t0 = t0 * m2;
                    t1 = t1 * m2;
                    t0 = t0 * inv_s;
                    t1 = t1 * inv_s;
                    *(gf2_ptr)(v + e) = v2f{p.x, p.y};
                    *(gf2_ptr)(err + e) = v2f{p.x - t0, p.y - t1};
                }
            }
            continue;   // (the other buckets of this wave take the register path below on their own lanes)
        }
        // chunk j of this lane covers elements [8 (c0 + LPB j), + 8)
        f32x4 x[2][2];
        float mx = 0.0f, mn = __builtin_inff();
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
            const int e = 8 * (c0 + LPB * jc);
            if (e < d) {
                x[jc][0] = *(gv_ptr)(v + e);
                x[jc][1] = *(gv_ptr)(v + e + 4);
                if (EF && err) {
                    const f32x4 q0 = *(gv_ptr)(err + e);
                    const f32x4 q1 = *(gv_ptr)(err + e + 4);
                    x[jc][0] = x[jc][0] + q0 * ef_scale;   // product rounded, then the add (-ffp-contract=off)
                    x[jc][1] = x[jc][1] + q1 * ef_scale;
                }
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    mx = absmax3_nan(mx, x[jc][k >> 2][k & 3], x[jc][(k + 1) >> 2][(k + 1) & 3]);   // NaN-propagating
                    mn = fminf(fminf(mn, fabsf(x[jc][k >> 2][k & 3])), fabsf(x[jc][(k + 1) >> 2][(k + 1) & 3]));   // (v_min3_f32: the lane's smallest |v|)
                }
            }
        }
#pragma unroll
        for (int o = LPB / 2; o > 0; o >>= 1) mx = max_nan(mx, __shfl_xor(mx, o, 64));   // the bucket's LPB lanes
        if (live && c0 == 0) reinterpret_cast<float *>(wire + rec[3])[lb] = mx;
        const uint64_t sd = random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, mx, mx) : seed;   // keyed by the bucket's norm
        const uint32_t key = bucket_draw_key(sd, b);   // the draws' stream of this bucket
        const float y = 1.0f / mx;                     // the bucket's ONE division (FAST: see qsgd_code)
        // (the error-feedback form moves 16.5 B per element and is bound by HBM: the quick quotient buys it nothing and its second
        // code path cost 14 registers -- 96 -> 110, four waves per SIMD instead of five, 0.111 -> 0.119 ms per ResNet-50 step)
        const bool fast = !EF && quotient_window(mx, mn);
        uint8_t *dst = wire + rec[4] + ((lb * d * BITS) >> 3);
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
            const int e = 8 * (c0 + LPB * jc);
            if (e < d) {
                unsigned code[8];
                f32x4 dec[2];
                if (fast && random_mode >= GQ_RANDOM_DEVICE) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) code[k] = qsgd_code<true, 1>(x[jc][k >> 2][k & 3], mx * inv_s, y * s, s, smax, random_mode, key, (uint32_t)(e + k), BITS);
                } else if (fast) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) code[k] = qsgd_code<true, 0>(x[jc][k >> 2][k & 3], mx * inv_s, y * s, s, smax, random_mode, key, (uint32_t)(e + k), BITS);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) code[k] = qsgd_code<false>(x[jc][k >> 2][k & 3], mx, 0.0f, s, smax, random_mode, key, (uint32_t)(e + k), BITS);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned c = code[k];
                    if (EF) {
                        // qsgd_compressor.py:69-70 on this element's own code (sign on the float's sign bit)
                        float t = __uint_as_float(__float_as_uint((float)(c & LMASK)) | (((c >> SB) ^ 1u) << 31));
                        t = t * mx;
                        dec[k >> 2][k & 3] = t * inv_s;
                    }
                }
                if (BITS == 4) {
                    unsigned word = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) word |= code[k] << (4 * k);
                    *reinterpret_cast<unsigned *>(dst + 4 * (c0 + LPB * jc)) = word;
                } else if (BITS == 8) {
                    *reinterpret_cast<uint2 *>(dst + e) = make_uint2(code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24),
                                                                     code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24));
                } else {
                    *reinterpret_cast<uint4 *>(dst + 2 * e) = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16),
                                                                         code[4] | (code[5] << 16), code[6] | (code[7] << 16));
                }
                if (EF && err) {
                    *(gv_ptr)(v + e) = x[jc][0];
                    *(gv_ptr)(v + e + 4) = x[jc][1];
                    *(gv_ptr)(err + e) = x[jc][0] - dec[0];      // ps_quantizer.py:39
                    *(gv_ptr)(err + e + 4) = x[jc][1] - dec[1];
                }
            }
        }
    }
}

// decode + mean over R users: one wave per bucket, out = ( sum_r (l * (2*sign-1)) * norm / 2^n_bit ) / R
__global__ __launch_bounds__(QB_THREADS) void qsgd_decode_sum_batched_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ bucket_seg, int64_t nbuckets, int n_bit,
    int bits, const uint8_t *__restrict__ gathered, int64_t user_stride, int R, float *__restrict__ out, int plain) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (QB_THREADS / 64);
    const float s = (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);   // the aggregate of R users (ps_quantizer.py:48)
    const unsigned lmask = (1u << (bits - 1)) - 1u;
    for (int64_t b = (int64_t)blockIdx.x * (QB_THREADS / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); b < nbuckets; b += nw) {
        const int seg = __builtin_amdgcn_readfirstlane(bucket_seg[b]);
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int d = (int)rec[1];
        const int64_t lb = b - rec[2];
        float *o = out + rec[5] + lb * d;
        for (int e = 2 * lane; e < d; e += 128) {
            float a0 = 0.0f, a1 = 0.0f;
            for (int r = 0; r < R; ++r) {
                const uint8_t *p = gathered + (int64_t)r * user_stride;
                const float norm = reinterpret_cast<const float *>(p + rec[3])[lb];
                unsigned c0, c1;
                if (bits == 4) {
                    const unsigned byte = p[rec[4] + ((lb * d + e) >> 1)];
                    c0 = byte & 15u;
                    c1 = byte >> 4;
                } else if (bits == 8) {
                    const uchar2 cc = *reinterpret_cast<const uchar2 *>(p + rec[4] + lb * d + e);
                    c0 = cc.x;
                    c1 = cc.y;
                } else {
                    const unsigned cc = *reinterpret_cast<const unsigned *>(p + rec[4] + 2 * (lb * d + e));
                    c0 = cc & 0xFFFFu;
                    c1 = cc >> 16;
                }
                // qsgd_compressor.py:69-70: (l * (2*signs - 1)) * norm / s
                float t0 = (float)(c0 & lmask) * (2.0f * (float)(c0 >> (bits - 1)) - 1.0f);
                float t1 = (float)(c1 & lmask) * (2.0f * (float)(c1 >> (bits - 1)) - 1.0f);
                t0 = t0 * norm;
                t1 = t1 * norm;
                t0 = t0 / s;
                t1 = t1 / s;
                a0 = (r == 0) ? t0 : a0 + t0;
                a1 = (r == 0) ? t1 : a1 + t1;
            }
            if (md.apply) {
                a0 = mean_div(a0, md);
                a1 = mean_div(a1, md);
            }
            *reinterpret_cast<float2 *>(o + e) = make_float2(a0, a1);
        }
    }
}

// decode + mean for the 4-bit packed wire: 16 lanes per bucket (a wave takes four buckets), a lane
// decodes 8 consecutive elements from ONE dword per payload and stores 32 contiguous bytes.  The first
// form (one wave per bucket, an element pair per lane) spent its time on byte loads, float2 stores and two
// IEEE divisions per element: 41 us (R = 1) ... 146 us (R = 8) for the 23.5 M-element ResNet-50 list.
// Same arithmetic: ((+-l) * norm) / 2^n_bit with the sign applied to the float's sign bit (l = 0 with a
// cleared sign bit decodes to -0 like the reference's 0 * -1), the division as an exact scaling.
__global__ __launch_bounds__(QB_THREADS) void qsgd_decode_sum_batched4_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ bucket_seg, int64_t nbuckets, int n_bit,
    const uint8_t *__restrict__ gathered, int64_t user_stride, int R, float *__restrict__ out, int plain) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, c0 = lane & 15;
    const int64_t nw = (int64_t)gridDim.x * (QB_THREADS / 64);
    const float inv_s = 1.0f / (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);   // the aggregate of R users (ps_quantizer.py:48)
    const int64_t nquads = (nbuckets + 3) >> 2;
    for (int64_t qd = (int64_t)blockIdx.x * (QB_THREADS / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); qd < nquads; qd += nw) {
        const int64_t b = 4 * qd + sub;
        if (b >= nbuckets) continue;
        const int seg = bucket_seg[b];
        const int64_t *rec = seg_table + 8 * (int64_t)seg;
        const int d = (int)rec[1];
        const int64_t lb = b - rec[2];
        const int64_t norm_off = rec[3], code_off = rec[4] + ((lb * d) >> 1);
        float *o = out + rec[5] + lb * d;
        if ((d & 7) == 0) {
            for (int c = c0; 8 * c < d; c += 16) {
                f32x4 acc[2];
                auto payload = [&](int r, auto first) {
                    const uint8_t *p = gathered + (int64_t)r * user_stride;
                    const float norm = reinterpret_cast<const float *>(p + norm_off)[lb];
                    const unsigned w = *reinterpret_cast<const unsigned *>(p + code_off + 4 * c);
                    const unsigned nw_ = ~w;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float lf = (float)((w >> (4 * k)) & 7u);
                        const unsigned sgn = (nw_ >> (4 * k + 3)) & 1u;          // 1: negative
                        float t = __uint_as_float(__float_as_uint(lf) | (sgn << 31));
                        t = t * norm;
                        t = t * inv_s;
                        if constexpr (decltype(first)::value) {
                            acc[k >> 2][k & 3] = t;
                        } else {
                            acc[k >> 2][k & 3] = acc[k >> 2][k & 3] + t;
                        }
                    }
                };
                payload(0, std::true_type{});
                for (int r = 1; r < R; ++r) payload(r, std::false_type{});
                if (md.apply) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k >> 2][k & 3] = mean_div(acc[k >> 2][k & 3], md);
                }
                *reinterpret_cast<f32x4 *>(o + 8 * c) = acc[0];
                *reinterpret_cast<f32x4 *>(o + 8 * c + 4) = acc[1];
            }
        } else {   // odd bucket widths: an element pair (one byte) at a time
            for (int e = 2 * c0; e < d; e += 32) {
                float a0 = 0.0f, a1 = 0.0f;
                for (int r = 0; r < R; ++r) {
                    const uint8_t *p = gathered + (int64_t)r * user_stride;
                    const float norm = reinterpret_cast<const float *>(p + norm_off)[lb];
                    const unsigned byte = p[code_off + (e >> 1)];
                    const unsigned c0_ = byte & 15u, c1_ = byte >> 4;
                    float t0 = (float)(c0_ & 7u) * (2.0f * (float)(c0_ >> 3) - 1.0f);
                    float t1 = (float)(c1_ & 7u) * (2.0f * (float)(c1_ >> 3) - 1.0f);
                    t0 = t0 * norm;
                    t1 = t1 * norm;
                    t0 = t0 * inv_s;
                    t1 = t1 * inv_s;
                    a0 = (r == 0) ? t0 : a0 + t0;
                    a1 = (r == 0) ? t1 : a1 + t1;
                }
                if (md.apply) {
                    a0 = mean_div(a0, md);
                    a1 = mean_div(a1, md);
                }
                *reinterpret_cast<float2 *>(o + e) = make_float2(a0, a1);
            }
        }
    }
}

// The same for a compile-time payload count R <= QB4_RMAX and a segment table that fits in LDS, software-pipelined across
// a wave's items like hsq_decode_sum_d16u8_r_kernel (hsq_decode.hip): the kernel above walks bucket -> segment record ->
// (norm, code word) of payload 0 .. R-1 as one chain of dependent round trips per item, each behind the previous item's
// stores.  Here an item's R (code word, norm) pairs live in registers and each pair is re-requested for the wave's NEXT
// item right after it has been consumed, the bucket -> segment word is fetched two items ahead and the record comes from
// the LDS copy of the table.  Buckets wider than 128 elements (more than one 8-element unit per lane) and widths that are
// not a multiple of 8 finish through the plain code inside the item.
constexpr int QB4_RMAX = 8;

// (+-level) as a float: the sign bit of the code is 1 for positive values, applied to the float's sign bit (a zero level
// with a cleared sign bit decodes to -0 like the reference's 0 * -1); a lane's unit of 8 codes is BITS / 4 dwords, codes in
// ascending element order from the low end
// ... of code k of a unit, straight from the unit's dwords `w` and their complements `nw` (two bit-field extracts per element)
template <int BITS>
__device__ __forceinline__ float unit_signed_level(const unsigned (&w)[BITS / 4], const unsigned (&nw)[BITS / 4], int k) {
    constexpr int PER = 32 / BITS;   // codes per dword
    const int word = k / PER, sh = BITS * (k % PER);
    const float lf = (float)((w[word] >> sh) & ((1u << (BITS - 1)) - 1u));
    const unsigned sgn = (nw[word] >> (sh + BITS - 1)) & 1u;          // 1: negative
    return __uint_as_float(__float_as_uint(lf) | (sgn << 31));
}
template <int BITS>
__device__ __forceinline__ void load_unit(const uint8_t *p, unsigned (&w)[BITS / 4]) {
    if (BITS == 4) {
        w[0] = *reinterpret_cast<const unsigned *>(p);
    } else if (BITS == 8) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        w[0] = v.x;
        w[BITS == 8 ? 1 : 0] = v.y;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        w[0] = v.x;
        w[BITS == 16 ? 1 : 0] = v.y;
        w[BITS == 16 ? 2 : 0] = v.z;
        w[BITS == 16 ? 3 : 0] = v.w;
    }
}

template <int R, int BITS = 4, int LPBL = -1>   // LPBL: log2 of the lanes per bucket when it is known at compile time (16: the 4-bit wire's common case)
__global__ __launch_bounds__(QB_THREADS) void qsgd_decode_sum_batched4_r_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ bucket_seg, int64_t nbuckets, int nseg, int n_bit,
    const uint8_t *__restrict__ gathered, int64_t user_stride, float *__restrict__ out, int plain, const StepTail tail, int lpb_log2) {
    __shared__ int64_t s_seg[QB_LDS_SEGS * 8];
    step_tail_run(tail);      // the aggregate's small per-step work (gq_qsgd_decode_sum_batched_tail)
    for (int i = threadIdx.x; i < nseg * 8; i += QB_THREADS) s_seg[i] = seg_table[i];
    __syncthreads();
    // lanes per bucket (16, 8 or 4; qsgd_compress_batched4_kernel's LPB): here it only moves indices, a kernel argument
    const int lpbl = LPBL >= 0 ? LPBL : lpb_log2;
    const int lpb = 1 << lpbl, bpw = 64 >> lpbl;
    const int lane = threadIdx.x & 63, sub = lane >> lpbl, c0 = lane & (lpb - 1);
    const int64_t nw = (int64_t)gridDim.x * (QB_THREADS / 64);
    const float inv_s = 1.0f / (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);
    const int64_t nquads = (nbuckets + bpw - 1) / bpw;
    typedef const uint8_t __attribute__((address_space(1))) gbyte;
    typedef const unsigned __attribute__((address_space(1))) gword;
    typedef unsigned uv2 __attribute__((ext_vector_type(2)));
    typedef unsigned uv4 __attribute__((ext_vector_type(4)));
    typedef const uv2 __attribute__((address_space(1))) gword2;
    typedef const uv4 __attribute__((address_space(1))) gword4;
    constexpr int W = BITS / 4;   // dwords of a lane's unit of 8 codes
    typedef const float __attribute__((address_space(1))) gfloat;
    const uint64_t wire0 = reinterpret_cast<uint64_t>(gathered);
    struct Item {
        unsigned norm_off, code_off;   // bytes inside one payload: this bucket's norm, this lane's code word
        int64_t out_off;               // floats: this lane's first output
        int64_t lb;                    // bucket index inside its tensor
        int d;                         // bucket width
        int seg;
    };
    auto bucket_of = [&](int64_t qd) {   // lanes past the last bucket redo it (nothing is stored for them)
        const int64_t b = (int64_t)bpw * qd + sub;
        return b < nbuckets ? b : nbuckets - 1;
    };
    auto item_of = [&](int64_t b, int seg) {
        const int64_t *rec = s_seg + 8 * seg;
        Item it;
        it.seg = seg;
        it.d = (int)rec[1];
        it.lb = b - rec[2];
        it.norm_off = (unsigned)(rec[3] + 4 * it.lb);
        it.code_off = (unsigned)(rec[4] + ((it.lb * it.d * BITS) >> 3) + BITS * c0);
        it.out_off = rec[5] + it.lb * it.d + 8 * c0;
        return it;
    };
    unsigned w[R][W];
    float nm[R];
    auto request = [&](const Item &it, unsigned guard, int r) {
        const uint64_t base = wire0 + (uint64_t)r * (uint64_t)user_stride;
        // a lane without a unit in this bucket (8 * c0 >= d) reads the bucket's first word instead of one past its codes
        const unsigned co = (8 * c0 < it.d ? it.code_off : it.code_off - (unsigned)BITS * (unsigned)c0) + guard;
        if constexpr (BITS == 4) {
            w[r][0] = *reinterpret_cast<gword *>(reinterpret_cast<gbyte *>(base) + co);
        } else if constexpr (BITS == 8) {
            const uv2 v = *reinterpret_cast<gword2 *>(reinterpret_cast<gbyte *>(base) + co);
            w[r][0] = v[0];
            w[r][1] = v[1];
        } else {
            const uv4 v = *reinterpret_cast<gword4 *>(reinterpret_cast<gbyte *>(base) + co);
            w[r][0] = v[0];
            w[r][1] = v[1];
            w[r][2] = v[2];
            w[r][3] = v[3];
        }
        nm[r] = *reinterpret_cast<gfloat *>(reinterpret_cast<gbyte *>(base) + (it.norm_off + guard));
    };
    int64_t qd = (int64_t)blockIdx.x * (QB_THREADS / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (qd >= nquads) return;
    int64_t qn = qd + nw < nquads ? qd + nw : qd;
    Item cur = item_of(bucket_of(qd), bucket_seg[bucket_of(qd)]);
    int seg_n = bucket_seg[bucket_of(qn)];
#pragma unroll
    for (int r = 0; r < R; ++r) request(cur, 0u, r);
    while (true) {
        const int64_t q2 = qn + nw < nquads ? qn + nw : qn;
        const int seg_2 = bucket_seg[bucket_of(q2)];            // consumed a whole item later
        const Item nxt = item_of(bucket_of(qn), seg_n);
        const bool mine = (int64_t)bpw * qd + sub < nbuckets;
        f32x4 acc[2];
        unsigned guard = 0;
        auto payload = [&](const unsigned (&ww)[W], float norm, auto first) {
            unsigned nww[W];
#pragma unroll
            for (int i = 0; i < W; ++i) nww[i] = ~ww[i];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = unit_signed_level<BITS>(ww, nww, k);
                t = t * norm;
                t = t * inv_s;
                if constexpr (decltype(first)::value) {
                    acc[k >> 2][k & 3] = t;
                } else {
                    acc[k >> 2][k & 3] = acc[k >> 2][k & 3] + t;
                }
            }
        };
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r == 0)
                payload(w[r], nm[r], std::true_type{});
            else
                payload(w[r], nm[r], std::false_type{});
            // each re-request stays behind the payload it replaces: its offset "depends" on the payload's last sum
            // (an empty non-volatile asm: no instruction, and not a store as far as the compiler's alias analysis goes)
            asm("" : "+v"(guard) : "v"(acc[1][3]));
            request(nxt, guard, r);
        }
        if (md.apply) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k >> 2][k & 3] = mean_div(acc[k >> 2][k & 3], md);
        }
        float *o = out + cur.out_off;
        if ((cur.d & 7) == 0) {
            if (mine && 8 * c0 < cur.d) {
                *reinterpret_cast<f32x4 *>(o) = acc[0];
                *reinterpret_cast<f32x4 *>(o + 4) = acc[1];
            }
            if (cur.d > 8 * lpb && mine) {   // further units of a wide bucket: the plain form
                for (int c = c0 + lpb; 8 * c < cur.d; c += lpb) {
                    f32x4 a2[2];
                    for (int r = 0; r < R; ++r) {
                        const uint8_t *p = gathered + (int64_t)r * user_stride;
                        const float norm = *reinterpret_cast<const float *>(p + cur.norm_off);
                        unsigned ww[W], nww[W];
                        load_unit<BITS>(p + cur.code_off + BITS * (c - c0), ww);
#pragma unroll
                        for (int i = 0; i < W; ++i) nww[i] = ~ww[i];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float t = unit_signed_level<BITS>(ww, nww, k);
                            t = t * norm;
                            t = t * inv_s;
                            a2[k >> 2][k & 3] = r == 0 ? t : a2[k >> 2][k & 3] + t;
                        }
                    }
                    if (md.apply) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) a2[k >> 2][k & 3] = mean_div(a2[k >> 2][k & 3], md);
                    }
                    *reinterpret_cast<f32x4 *>(o + 8 * (c - c0)) = a2[0];
                    *reinterpret_cast<f32x4 *>(o + 8 * (c - c0) + 4) = a2[1];
                }
            }
        } else if (mine) {   // odd bucket widths: an element pair (one byte) at a time, as in the kernel above
            const int64_t code0 = (int64_t)cur.code_off - BITS * c0;
            float *ob = o - 8 * c0;
            for (int e = 2 * c0; e < cur.d; e += 2 * lpb) {
                float a0 = 0.0f, a1 = 0.0f;
                for (int r = 0; r < R; ++r) {
                    const uint8_t *p = gathered + (int64_t)r * user_stride;
                    const float norm = *reinterpret_cast<const float *>(p + cur.norm_off);
                    unsigned c0_, c1_;
                    if (BITS == 4) {
                        const unsigned byte = p[code0 + (e >> 1)];
                        c0_ = byte & 15u;
                        c1_ = byte >> 4;
                    } else if (BITS == 8) {
                        c0_ = p[code0 + e];
                        c1_ = p[code0 + e + 1];
                    } else {
                        c0_ = reinterpret_cast<const uint16_t *>(p + code0)[e];
                        c1_ = reinterpret_cast<const uint16_t *>(p + code0)[e + 1];
                    }
                    constexpr unsigned LM = (1u << (BITS - 1)) - 1u;
                    float t0 = (float)(c0_ & LM) * (2.0f * (float)(c0_ >> (BITS - 1)) - 1.0f);
                    float t1 = (float)(c1_ & LM) * (2.0f * (float)(c1_ >> (BITS - 1)) - 1.0f);
                    t0 = t0 * norm;
                    t1 = t1 * norm;
                    t0 = t0 * inv_s;
                    t1 = t1 * inv_s;
                    a0 = (r == 0) ? t0 : a0 + t0;
                    a1 = (r == 0) ? t1 : a1 + t1;
                }
                if (md.apply) {
                    a0 = mean_div(a0, md);
                    a1 = mean_div(a1, md);
                }
                *reinterpret_cast<float2 *>(ob + e) = make_float2(a0, a1);
            }
        }
        if (qd + nw >= nquads) break;
        qd += nw;
        cur = nxt;
        qn = q2;
        seg_n = seg_2;
    }
}

template <int R, int BITS>
static void launch_qb4_r(const int64_t *seg_table, const int32_t *bucket_seg, int64_t nbuckets, int nseg, int n_bit,
                         const uint8_t *gathered, int64_t user_stride, float *out, int plain, hipStream_t st, const StepTail &tail, int lpb_log2);

// One resident wave of workgroups for `kernel` (the occupancy API) instead of a fixed 8 per CU: the 4-bit compress kernel
// holds 5-6 waves per SIMD (74-84 registers), so a quarter of an 8-per-CU grid queued behind the resident workgroups.
// (Measured: no difference for the ResNet-50 list, 36.7 us either way -- the kernel is bound by VALU issue, the quarter-rate
// v_mul_lo_u32 of the draws and v_rcp_f32 / v_div_* of the IEEE division; a software-pipelined loop, the next item's
// elements requested before the current item's arithmetic, cost 104 registers / 4 waves per SIMD and ran 41.5 us:
// profiles/r04_qsgd_pipeline.txt.)
template <typename KernelT>
static int64_t qb_grid_resident(KernelT kernel, int64_t nitems) {
    static const int bpc = [&] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, QB_THREADS, 0) != hipSuccess || n < 1) n = 8;
        return n;
    }();
    int64_t blocks = (nitems + (QB_THREADS / 64) - 1) / (QB_THREADS / 64);
    const int64_t cap = (int64_t)cu_count() * bpc;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

static inline int64_t qb_grid(int64_t nbuckets) {
    int64_t blocks = (nbuckets + (QB_THREADS / 64) - 1) / (QB_THREADS / 64);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

template <int R, int BITS>
static void launch_qb4_r(const int64_t *seg_table, const int32_t *bucket_seg, int64_t nbuckets, int nseg, int n_bit,
                         const uint8_t *gathered, int64_t user_stride, float *out, int plain, hipStream_t st, const StepTail &tail, int lpb_log2) {
    const int bpw = 64 >> lpb_log2;
    if (BITS == 4 && lpb_log2 == 4) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_decode_sum_batched4_r_kernel<R, BITS, 4>),
                           dim3((unsigned)qb_grid_resident(qsgd_decode_sum_batched4_r_kernel<R, BITS, 4>, (nbuckets + bpw - 1) / bpw)),
                           dim3(QB_THREADS), 0, st, seg_table, bucket_seg, nbuckets, nseg, n_bit, gathered, user_stride, out, plain, tail, lpb_log2);
        return;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_decode_sum_batched4_r_kernel<R, BITS>),
                       dim3((unsigned)qb_grid_resident(qsgd_decode_sum_batched4_r_kernel<R, BITS>, (nbuckets + bpw - 1) / bpw)),
                       dim3(QB_THREADS), 0, st, seg_table, bucket_seg, nbuckets, nseg, n_bit, gathered, user_stride, out, plain, tail, lpb_log2);
}

// lanes per bucket for a bucket-width hint: d / 8 lanes, a power of two between 2 and 16 (log2)
static inline int lpb_log2_of(int bucket_hint) {
    if (bucket_hint > 0 && bucket_hint <= 16) return 1;
    if (bucket_hint > 0 && bucket_hint <= 32) return 2;
    if (bucket_hint > 0 && bucket_hint <= 64) return 3;
    return 4;
}

template <int BITS>
static bool launch_qb4_fixed_r(int R, const int64_t *seg_table, const int32_t *bucket_seg, int64_t nbuckets, int nseg, int n_bit,
                               const uint8_t *gathered, int64_t user_stride, float *out, int plain, hipStream_t st, const StepTail &tail,
                               int lpb_log2) {
    // byte offsets inside a payload are 32-bit in this kernel, the table sits in LDS
    if (nseg > QB_LDS_SEGS || user_stride >= ((int64_t)1 << 31)) return false;
    switch (R) {
#define GQ_QB4_CASE(N) case N: launch_qb4_r<N, BITS>(seg_table, bucket_seg, nbuckets, nseg, n_bit, gathered, user_stride, out, plain, st, tail, lpb_log2); return true;
        GQ_QB4_CASE(1) GQ_QB4_CASE(2) GQ_QB4_CASE(3) GQ_QB4_CASE(4)
        GQ_QB4_CASE(5) GQ_QB4_CASE(6) GQ_QB4_CASE(7) GQ_QB4_CASE(8)
#undef GQ_QB4_CASE
        default: return false;
    }
}

}  // namespace gq

GQ_API int gq_qsgd_code_bits(int n_bit, int random_mode) {
    // levels reach 2^n_bit with stochastic rounding, 2^n_bit - 1 without; one more bit for the sign
    const int top = (1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0);
    if (top <= 7) return 4;
    if (top <= 127) return 8;
    if (top <= 32767) return 16;   // e.g. 8-bit QSGD with stochastic rounding (level 256)
    return 0;  // no packed format: use gq_qsgd_compress / gq_qsgd_decode_sum
}

namespace gq {
template <bool EF>
static int qsgd_compress_batched(const char *what, const int64_t *seg_table, const int32_t *bucket_seg, int nseg,
                                 int64_t nbuckets, int n_bit, int random_mode, uint64_t seed, float ef_scale,
                                 uint8_t *wire, const int64_t *dense_table, int ndense, int bucket_hint, void *stream) {
    if (nseg < 1 || nbuckets < 1 || n_bit < 1) return fail(GQ_ERR_INVALID_ARG, "%s: bad sizes", what);
    if (!seg_table || !bucket_seg || !wire) return fail(GQ_ERR_INVALID_ARG, "%s: null pointer", what);
    if (random_mode != GQ_RANDOM_OFF && random_mode != GQ_RANDOM_DEVICE && random_mode != GQ_RANDOM_DEVICE_KEYED &&
        random_mode != GQ_RANDOM_DEVICE_COUNTER)
        return fail(GQ_ERR_UNSUPPORTED, "%s: random_mode must be OFF, DEVICE, DEVICE_KEYED or DEVICE_COUNTER", what);
    const int bits = gq_qsgd_code_bits(n_bit, random_mode);
    if (!bits) return fail(GQ_ERR_UNSUPPORTED, "%s: n_bit %d has no packed format", what, n_bit);
    const int lpb_log2 = lpb_log2_of(bucket_hint);
#define GQ_QC_LAUNCH(BITSV, LPBV)                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_compress_batched4_kernel<EF, true, BITSV, LPBV>),                                         \
                       dim3((unsigned)qb_grid_resident(qsgd_compress_batched4_kernel<EF, true, BITSV, LPBV>,                          \
                                                       (nbuckets + 64 / LPBV - 1) / (64 / LPBV))),                                   \
                       dim3(QB_THREADS), 0, as_stream(stream), seg_table, bucket_seg, nseg, nbuckets, n_bit, random_mode, seed,      \
                       ef_scale, wire, dense_table, ndense)
#define GQ_QC_BITS(BITSV)                                  \
    do {                                                   \
        if (lpb_log2 == 1) GQ_QC_LAUNCH(BITSV, 2);         \
        else if (lpb_log2 == 2) GQ_QC_LAUNCH(BITSV, 4);    \
        else if (lpb_log2 == 3) GQ_QC_LAUNCH(BITSV, 8);    \
        else GQ_QC_LAUNCH(BITSV, 16);                      \
    } while (0)
    if (bits == 4 && nseg <= QB_LDS_SEGS) {
        GQ_QC_BITS(4);
    } else if (bits == 8 && nseg <= QB_LDS_SEGS) {
        GQ_QC_BITS(8);
    } else if (bits == 16 && nseg <= QB_LDS_SEGS) {
        GQ_QC_BITS(16);
#undef GQ_QC_BITS
#undef GQ_QC_LAUNCH
    } else if (bits == 4) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_compress_batched4_kernel<EF, false>),
                           dim3((unsigned)qb_grid_resident(qsgd_compress_batched4_kernel<EF, false>, (nbuckets + 3) / 4)),
                           dim3(QB_THREADS), 0, as_stream(stream), seg_table, bucket_seg, nseg, nbuckets, n_bit,
                           random_mode, seed, ef_scale, wire, dense_table, ndense);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(qsgd_compress_batched_kernel<EF>), dim3((unsigned)qb_grid(nbuckets)),
                           dim3(QB_THREADS), 0, as_stream(stream), seg_table, bucket_seg, nbuckets, n_bit, bits,
                           random_mode, seed, ef_scale, wire, dense_table, ndense);
    }
    GQ_CHECK_LAUNCH(what);
    return GQ_OK;
}
}  // namespace gq

GQ_INTERNAL int gqi_qsgd_compress_batched(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                                          int n_bit, int random_mode, uint64_t seed, int ef, float ef_scale, uint8_t *wire, const int64_t *dense_table, int ndense,
                                          int bucket_hint, void *stream) {
    if (ef)
        return gq::qsgd_compress_batched<true>("gq_qsgd_compress_batched", seg_table, bucket_seg, nseg, nbuckets, n_bit,
                                               random_mode, seed, ef_scale, wire, dense_table, ndense, bucket_hint, stream);
    return gq::qsgd_compress_batched<false>("gq_qsgd_compress_batched", seg_table, bucket_seg, nseg, nbuckets, n_bit,
                                            random_mode, seed, 0.0f, wire, dense_table, ndense, bucket_hint, stream);
}

GQ_INTERNAL int gqi_qsgd_decode_sum_batched(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                                            int n_bit, int bits, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                            float *out, int plain, const gq::StepTail *tail_or_null, int *tail_taken, int bucket_hint,
                                            void *stream) {
    plain = plain ? 1 : 0;
    const int lpb_log2 = gq::lpb_log2_of(bucket_hint);
    if (tail_taken) *tail_taken = 0;
    if (nseg < 1 || nbuckets < 1 || n_bit < 1 || R < 1 || (bits != 4 && bits != 8 && bits != 16))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum_batched: bad sizes");
    if (!seg_table || !bucket_seg || !gathered || !out)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum_batched: null pointer");
    if (bits == 4 && (user_stride_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(gathered) & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        if (gq::launch_qb4_fixed_r<4>(R, seg_table, bucket_seg, nbuckets, nseg, n_bit, gathered, user_stride_bytes, out, plain,
                                      gq::as_stream(stream), tail_or_null ? *tail_or_null : gq::StepTail{}, lpb_log2)) {
            GQ_CHECK_LAUNCH("gq_qsgd_decode_sum_batched");
            if (tail_taken) *tail_taken = 1;
            return GQ_OK;
        }
        hipLaunchKernelGGL(gq::qsgd_decode_sum_batched4_kernel, dim3((unsigned)gq::qb_grid((nbuckets + 3) / 4)),
                           dim3(gq::QB_THREADS), 0, gq::as_stream(stream), seg_table, bucket_seg, nbuckets, n_bit,
                           gathered, user_stride_bytes, R, out, plain);
    } else if (bits != 4 && (user_stride_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(gathered) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
               (bits == 8 ? gq::launch_qb4_fixed_r<8>(R, seg_table, bucket_seg, nbuckets, nseg, n_bit, gathered, user_stride_bytes, out,
                                                     plain, gq::as_stream(stream), tail_or_null ? *tail_or_null : gq::StepTail{}, lpb_log2)
                          : gq::launch_qb4_fixed_r<16>(R, seg_table, bucket_seg, nbuckets, nseg, n_bit, gathered, user_stride_bytes, out,
                                                      plain, gq::as_stream(stream), tail_or_null ? *tail_or_null : gq::StepTail{}, lpb_log2))) {
        // the 16-lanes-per-bucket kernel on 8- / 16-bit codes (R <= 8, the table in LDS); everything else: the generic kernel below
        GQ_CHECK_LAUNCH("gq_qsgd_decode_sum_batched");
        if (tail_taken) *tail_taken = 1;
        return GQ_OK;
    } else {
        hipLaunchKernelGGL(gq::qsgd_decode_sum_batched_kernel, dim3((unsigned)gq::qb_grid(nbuckets)),
                           dim3(gq::QB_THREADS), 0, gq::as_stream(stream), seg_table, bucket_seg, nbuckets, n_bit, bits,
                           gathered, user_stride_bytes, R, out, plain);
    }
    GQ_CHECK_LAUNCH("gq_qsgd_decode_sum_batched");
    return GQ_OK;
}
