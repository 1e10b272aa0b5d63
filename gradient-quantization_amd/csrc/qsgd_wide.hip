// QSGD / TernGrad with WIDE buckets on the packed wire, multi-tensor (segment table) form.
//
// The reference's TernGrad command is `--quantizer qsgd --c-dim 0 --n-bit 1`: c_dim = 0 makes the whole
// tensor ONE bucket (qsgd_compressor.py:15-16), up to 2.4 M elements for ResNet-50, and any c_dim above a few
// thousand gives buckets that no single wave should walk.  Same arithmetic and the same wire as
// qsgd_batched.hip (norm f32[buckets] | sign<<(bits-1) | level, 4-bit codes two per byte), but the unit of
// work is a CHUNK of GQ_QSGD_WIDE_CHUNK = 1024 consecutive elements of one bucket, one wave per chunk:
//   pass 1  max |v| of the chunk -> integer atomic max on the bucket's word of `norm_bits` (|v| >= 0, so the
//           unsigned order of the bit patterns is the float order); with error feedback the chunk is read as
//           v = grad + ef_scale*error and written back (ps_quantizer.py:35)
//   pass 2  codes of the chunk from the finished norm (the second read comes out of the 256 MB MALL);
//           error = v - decode(code) with error feedback (ps_quantizer.py:39)
//   decode  mean over R users' wires, a chunk per wave.
// chunk_seg[chunk] names the tensor; seg_table[seg] = { grad ptr (8-byte aligned), d, first chunk, norm off,
// codes off (bytes inside ONE user's wire), out off (floats), first word in norm_bits, error ptr }.
// HBM-bound: 4 B read (+4 B from the MALL) and 0.5..1 B written per element.
#include "gq_common.hpp"

namespace gq {

constexpr int QW_THREADS = 256;
constexpr int QW_CHUNK = GQ_QSGD_WIDE_CHUNK;

typedef float v2f __attribute__((ext_vector_type(2)));

struct WideItem {
    const int64_t *rec;
    int64_t b;        // bucket inside the tensor
    int64_t first;    // first element of the chunk inside the bucket
    int n;            // elements in the chunk (even, <= QW_CHUNK)
    int d;
};

__device__ __forceinline__ WideItem wide_item(const int64_t *seg_table, const int32_t *chunk_seg, int64_t c) {
    WideItem it;
    const int seg = __builtin_amdgcn_readfirstlane(chunk_seg[c]);
    it.rec = seg_table + 8 * (int64_t)seg;
    it.d = (int)it.rec[1];
    const int64_t local = c - it.rec[2];
    const int64_t cpb = ((int64_t)it.d + QW_CHUNK - 1) / QW_CHUNK;
    it.b = local / cpb;
    it.first = (local - it.b * cpb) * QW_CHUNK;
    const int64_t left = (int64_t)it.d - it.first;
    it.n = left < QW_CHUNK ? (int)left : QW_CHUNK;
    return it;
}

// Every wave takes ONE contiguous run of chunks: its running max stays with a bucket (one atomic per run and
// bucket instead of one per chunk -- a grid-stride walk sent ~8000 atomics at the same few cache lines in one
// burst: 136 us for the ResNet-50 list), and consecutive chunks share their tensor's record.
__device__ __forceinline__ void wide_run(int64_t nchunks, int64_t &begin, int64_t &end) {
    const int64_t nw = (int64_t)gridDim.x * (QW_THREADS / 64);
    const int64_t w = (int64_t)blockIdx.x * (QW_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t per = (nchunks + nw - 1) / nw;
    begin = w * per;
    end = begin + per < nchunks ? begin + per : nchunks;
}

// a lane takes the element pairs [2*lane + 128*m, +2), m = 0 .. 7, of the chunk: every load -- and, with error feedback, every
// store of the updated gradient -- of a wave is 512 contiguous bytes.  (The quantise kernel's ownership, 8 consecutive elements
// per lane, made 8-byte pieces 32 bytes apart of them: the error-feedback form of this kernel took 107 us for the ResNet-50
// list, 2.2 x what its bytes cost.)
template <bool EF>
__global__ __launch_bounds__(QW_THREADS) void qsgd_wide_absmax_kernel(const int64_t *__restrict__ seg_table,
                                                                     const int32_t *__restrict__ chunk_seg,
                                                                     int64_t nchunks, float ef_scale,
                                                                     unsigned *__restrict__ norm_bits) {
    const int lane = threadIdx.x & 63;
    int64_t c_begin, c_end;
    wide_run(nchunks, c_begin, c_end);
    unsigned *cur = nullptr;     // the bucket word the running max belongs to
    float run_mx = 0.0f;
    auto flush = [&]() {
        const float m = wave_max_nan(run_mx);   // a NaN (positive: |.| cleared its sign) is the largest integer as well
        if (lane == 0 && cur) {
            // look before the atomic: the word only grows, so a value already as large makes ours redundant
            const unsigned mine = __float_as_uint(m);
            if (mine > __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(cur, mine);
        }
        run_mx = 0.0f;
    };
    // the chunk -> tensor -> record lookups of chunk c + 1 are issued before chunk c's data is touched: three
    // dependent round trips per chunk otherwise, and a wave only has a handful of chunks to hide them behind
    WideItem nxt = wide_item(seg_table, chunk_seg, c_begin < c_end ? c_begin : 0);
    for (int64_t c = c_begin; c < c_end; ++c) {
        const WideItem it = nxt;
        if (c + 1 < c_end) nxt = wide_item(seg_table, chunk_seg, c + 1);
        float *v = reinterpret_cast<float *>(it.rec[0]) + it.b * it.d + it.first;
        const float *err = (EF && it.rec[7]) ? reinterpret_cast<const float *>(it.rec[7]) + it.b * it.d + it.first : nullptr;
        unsigned *word = norm_bits + it.rec[6] + it.b;
        if (word != cur) {
            flush();
            cur = word;
        }
        float mx = run_mx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 2 * lane + 128 * (4 * i + k);
                if (e < it.n) {
                    v2f p = *reinterpret_cast<const v2f *>(v + e);
                    if (EF && err) {   // product rounded, then the add (in place, like the reference's add_)
                        const v2f q = *reinterpret_cast<const v2f *>(err + e);
                        const float p0 = ef_scale * q[0], p1 = ef_scale * q[1];
                        p[0] = p[0] + p0;
                        p[1] = p[1] + p1;
                        *reinterpret_cast<v2f *>(v + e) = p;
                    }
                    mx = absmax3_nan(mx, p[0], p[1]);   // NaN-propagating; as an integer a NaN's bits also win the atomic max below
                }
            }
        }
        run_mx = mx;
    }
    flush();
}

__device__ __forceinline__ unsigned wide_code(float v, float norm, float s, float smax, int random_mode, uint64_t seed,
                                              uint64_t gidx, int bits) {
    const float q = v / norm;
    const float x = fabsf(q) * s;
    unsigned l = 0, sgn = v > 0.0f ? 1u : 0u;
    if (x != x) {  // NaN (zero bucket) -> level 0 carrying the sign of the reference's INT_MIN level (qsgd_batched.hip: qsgd_code)
        sgn ^= 1u;
    } else {
        const float c = fminf(fmaxf(x, 0.0f), smax);
        l = (unsigned)(int)c;
        if (random_mode >= GQ_RANDOM_DEVICE) {   // DEVICE, or DEVICE_KEYED with the keyed seed handed in
            const float prob = x - (float)l;
            l += (prob > uniform01(seed, gidx)) ? 1u : 0u;
        }
    }
    return l | (sgn << (bits - 1));
}

template <bool EF, int BITS>
__global__ __launch_bounds__(QW_THREADS) void qsgd_wide_quantise_kernel(
    const int64_t *__restrict__ seg_table, const int32_t *__restrict__ chunk_seg, int64_t nchunks, int n_bit,
    int random_mode, uint64_t seed, const unsigned *__restrict__ norm_bits, uint8_t *__restrict__ wire, const int64_t *__restrict__ dense_table, int ndense) {
    resolve_seed(random_mode, seed);
    copy_dense_segments(dense_table, ndense, wire);
    const int lane = threadIdx.x & 63;
    const float s = (float)(1 << n_bit), smax = s - 1.0f;
    constexpr unsigned lmask = (1u << (BITS - 1)) - 1u;
    int64_t c_begin, c_end;
    wide_run(nchunks, c_begin, c_end);
    // the chunk -> tensor -> record lookups of chunk c + 1 are issued before chunk c's data is touched: three
    // dependent round trips per chunk otherwise, and a wave only has a handful of chunks to hide them behind
    WideItem nxt = wide_item(seg_table, chunk_seg, c_begin < c_end ? c_begin : 0);
    for (int64_t c = c_begin; c < c_end; ++c) {
        const WideItem it = nxt;
        if (c + 1 < c_end) nxt = wide_item(seg_table, chunk_seg, c + 1);
        const int64_t at = it.b * it.d + it.first;                 // first element of the chunk inside the tensor
        const float *v = reinterpret_cast<const float *>(it.rec[0]) + at;
        float *err = (EF && it.rec[7]) ? reinterpret_cast<float *>(it.rec[7]) + at : nullptr;
        const float norm = __uint_as_float(norm_bits[it.rec[6] + it.b]);
        if (it.first == 0 && lane == 0) reinterpret_cast<float *>(wire + it.rec[3])[it.b] = norm;
        uint8_t *dst = wire + it.rec[4] + ((at * BITS) >> 3);
        const bool dwords = ((at & 7) == 0);                       // code dwords of whole 8-element groups are aligned
        const bool quads = ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(err)) & 15) == 0;   // 16-byte accesses (a null err: aligned)
        const uint64_t g0 = ((uint64_t)(it.rec[6] + it.b) << 32) + (uint64_t)it.first;   // RNG index: (bucket, element)
        const uint64_t sd = random_mode == GQ_RANDOM_DEVICE_KEYED ? keyed_seed(seed, norm, norm) : seed;   // keyed by the bucket's norm
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e0 = 8 * lane + 512 * i;
            if (e0 >= it.n) continue;
            unsigned code[8];
            float val[8];
            const bool whole = e0 + 8 <= it.n;
            if (whole && quads) {   // the lane's 32 bytes as two 16-byte loads
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(v + e0), p1 = *reinterpret_cast<const f32x4 *>(v + e0 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    val[k] = p0[k];
                    val[4 + k] = p1[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + 2 * k;
                    v2f p = {0.0f, 0.0f};
                    if (e < it.n) p = *reinterpret_cast<const v2f *>(v + e);
                    val[2 * k] = p[0];
                    val[2 * k + 1] = p[1];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                code[k] = wide_code(val[k], norm, s, smax, random_mode, sd, g0 + (uint64_t)(e0 + k), BITS);
            if (BITS == 4) {
                if (whole && dwords) {
                    unsigned w = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) w |= code[k] << (4 * k);
                    *reinterpret_cast<unsigned *>(dst + (e0 >> 1)) = w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e0 + 2 * k < it.n) dst[(e0 >> 1) + k] = (uint8_t)(code[2 * k] | (code[2 * k + 1] << 4));
                }
            } else if (BITS == 16) {
                if (whole && dwords) {
                    *reinterpret_cast<uint4 *>(dst + 2 * e0) = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16),
                                                                          code[4] | (code[5] << 16), code[6] | (code[7] << 16));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e0 + 2 * k < it.n)
                            *reinterpret_cast<unsigned *>(dst + 2 * (e0 + 2 * k)) = code[2 * k] | (code[2 * k + 1] << 16);
                }
            } else {
                if (whole && dwords) {
                    const unsigned w0 = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
                    const unsigned w1 = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
                    *reinterpret_cast<uint2 *>(dst + e0) = make_uint2(w0, w1);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (e0 + k < it.n) dst[e0 + k] = (uint8_t)code[k];
                }
            }
            if (EF && err) {
                float res[8];   // qsgd_compressor.py:69-70 on the element's own code, then ps_quantizer.py:39
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float t = (float)(code[k] & lmask) * (2.0f * (float)(code[k] >> (BITS - 1)) - 1.0f);
                    t = t * norm;
                    t = t / s;
                    res[k] = val[k] - t;
                }
                if (whole && quads) {
                    *reinterpret_cast<f32x4 *>(err + e0) = f32x4{res[0], res[1], res[2], res[3]};
                    *reinterpret_cast<f32x4 *>(err + e0 + 4) = f32x4{res[4], res[5], res[6], res[7]};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int e = e0 + 2 * k;
                        if (e < it.n) *reinterpret_cast<v2f *>(err + e) = v2f{res[2 * k], res[2 * k + 1]};
                    }
                }
            }
        }
    }
}

// decode + mean: ((+-l) * norm) / 2^n_bit per payload (the division as an exact scaling), payloads in ascending
// order, then / R -- the arithmetic of qsgd_decode_sum_batched4_kernel.
template <int BITS>
__global__ __launch_bounds__(QW_THREADS) void qsgd_wide_decode_kernel(const int64_t *__restrict__ seg_table,
                                                                     const int32_t *__restrict__ chunk_seg,
                                                                     int64_t nchunks, int n_bit,
                                                                     const uint8_t *__restrict__ gathered,
                                                                     int64_t user_stride, int R,
                                                                     float *__restrict__ out, int plain, const StepTail tail) {
    step_tail_run(tail);      // the aggregate's small per-step work (gq_qsgd_decode_sum_batched_tail)
    const int lane = threadIdx.x & 63;
    const float inv_s = 1.0f / (float)(1 << n_bit);
    const MeanDiv md = mean_div_of(R, !plain);   // the aggregate of R users (ps_quantizer.py:48)
    constexpr unsigned lmask = (1u << (BITS - 1)) - 1u;
    int64_t c_begin, c_end;
    wide_run(nchunks, c_begin, c_end);
    // the chunk -> tensor -> record lookups of chunk c + 1 are issued before chunk c's data is touched: three
    // dependent round trips per chunk otherwise, and a wave only has a handful of chunks to hide them behind
    WideItem nxt = wide_item(seg_table, chunk_seg, c_begin < c_end ? c_begin : 0);
    for (int64_t c = c_begin; c < c_end; ++c) {
        const WideItem it = nxt;
        if (c + 1 < c_end) nxt = wide_item(seg_table, chunk_seg, c + 1);
        const int64_t at = it.b * it.d + it.first;
        float *o = out + it.rec[5] + at;
        const int64_t code_off = it.rec[4] + ((at * BITS) >> 3);
        const bool dwords = ((at & 7) == 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e0 = 8 * lane + 512 * i;
            if (e0 >= it.n) continue;
            const bool whole = e0 + 8 <= it.n;
            float acc[8];
            for (int r = 0; r < R; ++r) {
                const uint8_t *p = gathered + (int64_t)r * user_stride;
                const float norm = reinterpret_cast<const float *>(p + it.rec[3])[it.b];
                unsigned code[8];
                if (BITS == 4) {
                    unsigned w = 0;
                    if (whole && dwords) {
                        w = *reinterpret_cast<const unsigned *>(p + code_off + (e0 >> 1));
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (e0 + 2 * k < it.n) w |= (unsigned)p[code_off + (e0 >> 1) + k] << (8 * k);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) code[k] = (w >> (4 * k)) & 15u;
                } else if (BITS == 16) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned cc = (e0 + 2 * k < it.n)
                                                ? *reinterpret_cast<const unsigned *>(p + code_off + 2 * (e0 + 2 * k)) : 0u;
                        code[2 * k] = cc & 0xFFFFu;
                        code[2 * k + 1] = cc >> 16;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) code[k] = (e0 + k < it.n) ? (unsigned)p[code_off + e0 + k] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float lf = (float)(code[k] & lmask);
                    const unsigned neg = ((~code[k]) >> (BITS - 1)) & 1u;
                    float t = __uint_as_float(__float_as_uint(lf) | (neg << 31));   // l * (2*sign - 1), -0 for l = 0
                    t = t * norm;
                    t = t * inv_s;
                    acc[k] = (r == 0) ? t : acc[k] + t;
                }
            }
            if (md.apply) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = mean_div(acc[k], md);
            }
            if (whole && dwords) {   // tensors start 16-byte aligned in `out` and at % 8 == 0: two 16-byte stores
                *reinterpret_cast<f32x4 *>(o + e0) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f32x4 *>(o + e0 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + 2 * k;
                    if (e < it.n) *reinterpret_cast<v2f *>(o + e) = v2f{acc[2 * k], acc[2 * k + 1]};
                }
            }
        }
    }
}

static inline int64_t qw_grid(int64_t nchunks) {
    int64_t blocks = (nchunks + (QW_THREADS / 64) - 1) / (QW_THREADS / 64);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

}  // namespace gq

GQ_INTERNAL int gqi_qsgd_wide_compress(const int64_t *seg_table, const int32_t *chunk_seg, int nseg, int64_t nchunks,
                                 int n_bit, int random_mode, uint64_t seed, int ef, float ef_scale,
                                 uint32_t *norm_bits, uint8_t *wire, const int64_t *dense_table, int ndense, void *stream) {
    if (nseg < 1 || nchunks < 1 || n_bit < 1) return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress_batched (wide): bad sizes");
    if (!seg_table || !chunk_seg || !norm_bits || !wire)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress_batched (wide): null pointer");
    if (random_mode != GQ_RANDOM_OFF && random_mode != GQ_RANDOM_DEVICE && random_mode != GQ_RANDOM_DEVICE_KEYED &&
        random_mode != GQ_RANDOM_DEVICE_COUNTER)
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_qsgd_compress_batched (wide): random_mode must be OFF, DEVICE, DEVICE_KEYED or DEVICE_COUNTER");
    const int bits = gq_qsgd_code_bits(n_bit, random_mode);
    if (!bits) return gq::fail(GQ_ERR_UNSUPPORTED, "gq_qsgd_compress_batched (wide): n_bit %d has no packed format", n_bit);
    hipStream_t st = gq::as_stream(stream);
    const dim3 grid((unsigned)gq::qw_grid(nchunks)), block(gq::QW_THREADS);
    if (ef)
        hipLaunchKernelGGL(gq::qsgd_wide_absmax_kernel<true>, grid, block, 0, st, seg_table, chunk_seg, nchunks, ef_scale,
                           norm_bits);
    else
        hipLaunchKernelGGL(gq::qsgd_wide_absmax_kernel<false>, grid, block, 0, st, seg_table, chunk_seg, nchunks, 0.0f,
                           norm_bits);
#define GQ_QW_LAUNCH(EFV, BITSV)                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gq::qsgd_wide_quantise_kernel<EFV, BITSV>), grid, block, 0, st, seg_table, chunk_seg, \
                       nchunks, n_bit, random_mode, seed, norm_bits, wire, dense_table, ndense)
    if (ef && bits == 4) GQ_QW_LAUNCH(true, 4);
    else if (ef && bits == 8) GQ_QW_LAUNCH(true, 8);
    else if (ef) GQ_QW_LAUNCH(true, 16);
    else if (bits == 4) GQ_QW_LAUNCH(false, 4);
    else if (bits == 8) GQ_QW_LAUNCH(false, 8);
    else GQ_QW_LAUNCH(false, 16);
#undef GQ_QW_LAUNCH
    GQ_CHECK_LAUNCH("gq_qsgd_compress_batched (wide)");
    return GQ_OK;
}

GQ_INTERNAL int gqi_qsgd_wide_decode_sum(const int64_t *seg_table, const int32_t *chunk_seg, int nseg, int64_t nchunks,
                                         int n_bit, int bits, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                         float *out, int plain, const gq::StepTail *tail_or_null, int *tail_taken, void *stream) {
    plain = plain ? 1 : 0;
    if (tail_taken) *tail_taken = 0;
    const gq::StepTail tail = tail_or_null ? *tail_or_null : gq::StepTail{};
    if (nseg < 1 || nchunks < 1 || n_bit < 1 || R < 1 || (bits != 4 && bits != 8 && bits != 16))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum_batched (wide): bad sizes");
    if (!seg_table || !chunk_seg || !gathered || !out)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum_batched (wide): null pointer");
    if ((user_stride_bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(gathered) & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(out) & 15) != 0)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_decode_sum_batched (wide): wires must be 4-byte, out 16-byte aligned");
    const dim3 grid((unsigned)gq::qw_grid(nchunks)), block(gq::QW_THREADS);
    if (bits == 4)
        hipLaunchKernelGGL(gq::qsgd_wide_decode_kernel<4>, grid, block, 0, gq::as_stream(stream), seg_table, chunk_seg,
                           nchunks, n_bit, gathered, user_stride_bytes, R, out, plain, tail);
    else if (bits == 16)
        hipLaunchKernelGGL(gq::qsgd_wide_decode_kernel<16>, grid, block, 0, gq::as_stream(stream), seg_table, chunk_seg,
                           nchunks, n_bit, gathered, user_stride_bytes, R, out, plain, tail);
    else
        hipLaunchKernelGGL(gq::qsgd_wide_decode_kernel<8>, grid, block, 0, gq::as_stream(stream), seg_table, chunk_seg,
                           nchunks, n_bit, gathered, user_stride_bytes, R, out, plain, tail);
    GQ_CHECK_LAUNCH("gq_qsgd_decode_sum_batched (wide)");
    if (tail_taken) *tail_taken = 1;
    return GQ_OK;
}
