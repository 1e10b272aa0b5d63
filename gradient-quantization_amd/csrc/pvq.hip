// ProbabilisticVectorCompressor encode (unbiased vector quantisation) for gfx950.
//
// Replaces probabilistic_vector_compressor.py:42-63 of the reference.  Pinned by tests/golden/pvq_*.npz and
// residual_*.npz: the reference's own class produced them with ONE operation defined by the generator
// (torch.argmin of a bool tensor, :58, exists in no torch: defined as "first True - 1", the inverse-CDF sample
// the line is written for; tests/golden/make_golden.py), every other line unedited:
//     p     = C_dagger . v                    C_dagger = pinv(codewords^T),  [K, d]     (:47)  fmaf chain
//     l1    = sum_k |p_k|                     sequential f32 adds, k ascending               (:48)
//     cum_k = f32( sum_{i<=k} f64(|p_i| / l1) )   torch.cumsum on the CPU accumulates in DOUBLE   (:49,:57)
//     code  = first k with  cum_k  >=  r - 1e-5 ,   r ~ U[0,1) per subvector              (:52-58)
//     u     = sign(p_code) * l1                                                     (:60-61)
// so that E[ codewords[code] * u ] = sum_k p_k c_k = v  (unbiased for a full-rank codebook).
// pvq_encode_lds_kernel (below) runs it on the matrix cores for any d <= 104; pvq_encode_kernel is the VALU
// cross-check and the fallback for wider subvectors: one subvector per lane; C_dagger is broadcast from LDS (or
// read through L1 when it does not fit); two sweeps over the K codewords (l1, then the inverse-CDF walk) with the
// same fmaf chain as the NearestNeighbor encode, ~2*K*d FMAs per subvector.
#include "hsq_encode_common.hpp"

namespace gq {

constexpr int PV_THREADS = 256;
typedef float pv_f32x2 __attribute__((ext_vector_type(2)));

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
        double cum = 0.0;   // torch.cumsum's accumulator on the CPU (acc_type<float> = double), outputs rounded to f32
        float sel = 0.0f;
        int code = K - 1;
        bool found = false;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) acc = __fmaf_rn(cd[k * D + j], v[j], acc);
            cum = cum + (double)(fabsf(acc) / l1);   // cumsum(|p| / l1), as the reference divides first (:49,:57)
            const bool hit = !found && ((float)cum >= thr);
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

// ------------------------------------------------------------------------------------
// The same on the matrix cores, any d <= 104 and any K: both operands staged in LDS exactly as in
// hsq_encode_lds_kernel (hsq_encode.hip, where the layout is described); v_mfma_f32_32x32x2_f32 evaluates the
// oracle's fmaf chain bit for bit.  A wave owns a tile of 64 subvectors as two 32-column blocks; after a row
// block's two accumulators are exchanged across the wave halves (v_permlane32_swap), LANE L holds all 32
// scores of ITS subvector L: registers x[4q..4q+3] are codewords 8q..8q+3 of the block, y[4q..4q+3] codewords
// 8q+4..8q+7 -- so the two sequential sums of the oracle (l1, then the running inverse-CDF sum with its
// division per term) run lane-locally, in ascending k, with no cross-lane step.  Two sweeps over the codebook
// (l1 has to be complete before the walk); the code is the number of terms whose running sum stayed below the
// threshold (the sums never decrease, so "first k with cum >= thr" == that count; NaN -- an all-zero
// subvector -- counts every term and lands on K-1 like the oracle); the selected projection is recomputed as
// one explicit fmaf chain.  Padded codewords (K not a multiple of 32) score +0 and change neither sum.

// The double T with  (float)x >= thr  <=>  x >= T  for every double x (round to nearest even): the midpoint between
// thr and the float below it when the tie goes to thr (even mantissa), the next double above the midpoint otherwise.
__device__ __forceinline__ double rounds_up_to_threshold(float thr) {
    const uint32_t tb = __float_as_uint(thr);
    if ((tb & 0x7F800000u) == 0x7F800000u) return (double)thr;   // +-inf, NaN: the comparison is the same in double
    const float below = thr > 0.0f ? __uint_as_float(tb - 1u)
                                   : (thr < 0.0f ? __uint_as_float(tb + 1u) : __uint_as_float(0x80000001u));
    const double mid = 0.5 * ((double)thr + (double)below);
    if ((tb & 1u) == 0) return mid;
    const int64_t mb = __double_as_longlong(mid);
    return __longlong_as_double(mid > 0.0 ? mb + 1 : mb - 1);
}

// ------------------------------------------------------------------------------------
// Second stage of the ResidualCompressor (residual_compressor.py:15-24) without a residual tensor: the tile is
// staged as  v - codebook1[code1] * norm1  -- stage 1's decode (nearest_neighbor_compressor.py:85-89: gather x norm,
// the product rounded) subtracted from the gradient (`residuals -= decompressed`, :22), element by element, with the
// reference's roundings.  norm1 is the de-quantised stage-1 norm per subvector (M floats: 1/d of the gradient).
struct PvqResidual {
    const void *codes1;     // null: plain encode of `grad`
    int code1_bytes;        // 1 | 4
    const float *norm1;
    const float *cb1;       // stage 1's codebook [K1, d]
};

template <typename CodeT>
__global__ __launch_bounds__(ENC_THREADS) void pvq_encode_lds_kernel(const float *__restrict__ grad,
                                                                    const float *__restrict__ cdag, int64_t M, int d,
                                                                    int K, int random_mode,
                                                                    const float *__restrict__ r, uint64_t seed,
                                                                    CodeT *__restrict__ codes, float *__restrict__ u,
                                                                    float *__restrict__ partials, int dpad,
                                                                    int chunk_rows, PvqResidual rs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const int stride = dpad + LDS_ROW_PAD, half = dpad >> 1;
    float *const s_cb = lds;
    float *const s_v = lds + (size_t)chunk_rows * stride + (size_t)wave * 64 * stride;
    const int kpad = (K + 31) & ~31;
    const int nchunks = (kpad + chunk_rows - 1) / chunk_rows;
    const float inv_dpad = 1.0f / (float)dpad;
    auto stage_codebook = [&](int row0) {
        const int total = chunk_rows * dpad;
        for (int i = threadIdx.x; i < total; i += ENC_THREADS) {
            const int rr = (int)(((float)i + 0.5f) * inv_dpad);
            const int e = i - rr * dpad;
            const int row = row0 + rr;
            const float val = (row < K && e < d) ? cdag[(int64_t)row * d + e] : 0.0f;
            s_cb[rr * stride + (e & 1) * half + (e >> 1)] = val;
        }
    };
    auto stage_tile = [&](int64_t t) {
        const int total = 64 * dpad;
        const int64_t sv0 = t * 64;
        for (int i = lane; i < total; i += 64) {
            const int rr = (int)(((float)i + 0.5f) * inv_dpad);
            const int e = i - rr * dpad;
            float val = (sv0 + rr < M && e < d) ? grad[(sv0 + rr) * (int64_t)d + e] : 0.0f;
            if (rs.codes1 && sv0 + rr < M && e < d) {
                const int64_t m = sv0 + rr;
                const int c1 = rs.code1_bytes == 1 ? (int)static_cast<const uint8_t *>(rs.codes1)[m]
                                                   : static_cast<const int32_t *>(rs.codes1)[m];
                const float dec = rs.cb1[(int64_t)c1 * d + e] * rs.norm1[m];   // stage 1's decoded element (product rounded)
                val = val - dec;
            }
            s_v[rr * stride + (e & 1) * half + (e >> 1)] = val;
        }
    };
    // Rows that need no padding (d a multiple of 8, at most 32) are fetched as float4s, one tile AHEAD: the loads of
    // the next tile are in flight while this one is swept, so the sweeps never wait for HBM (the element-wise loop
    // above waits for every one of its loads).  Float4 q of a tile is row q / (d/4), elements 4 (q mod d/4) ...
    const bool vec = d == dpad && d <= 32 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0 &&
                     (!rs.codes1 || (reinterpret_cast<uintptr_t>(rs.cb1) & 15) == 0);
    const int nq = d >> 2;
    const float inv_nq = 1.0f / (float)nq;
    f32x4 pre[8];
    int pre_c1[8];
    float pre_n1[8];
    auto fetch_tile = [&](int64_t t) {
        const int64_t sv0 = t * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= nq) break;
            const int q = i * 64 + lane;
            const int rr = (int)(((float)q + 0.5f) * inv_nq);
            const int e0 = (q - rr * nq) * 4;
            const int64_t m = sv0 + rr;
            pre[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            pre_c1[i] = 0;
            pre_n1[i] = 0.0f;
            if (m < M) {
                pre[i] = *reinterpret_cast<const f32x4 *>(grad + m * (int64_t)d + e0);
                if (rs.codes1) {
                    pre_c1[i] = rs.code1_bytes == 1 ? (int)static_cast<const uint8_t *>(rs.codes1)[m]
                                                    : static_cast<const int32_t *>(rs.codes1)[m];
                    pre_n1[i] = rs.norm1[m];
                }
            }
        }
    };
    auto commit_tile = [&](int64_t t) {
        const int64_t sv0 = t * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= nq) break;
            const int q = i * 64 + lane;
            const int rr = (int)(((float)q + 0.5f) * inv_nq);
            const int e0 = (q - rr * nq) * 4;
            f32x4 val = pre[i];
            if (rs.codes1 && sv0 + rr < M) {
                const f32x4 c = *reinterpret_cast<const f32x4 *>(rs.cb1 + (int64_t)pre_c1[i] * d + e0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dec = c[e] * pre_n1[i];   // stage 1's decoded element (product rounded)
                    val[e] = val[e] - dec;
                }
            }
            float *row = s_v + rr * stride + (e0 >> 1);
            *reinterpret_cast<pv_f32x2 *>(row) = pv_f32x2{val[0], val[2]};
            *reinterpret_cast<pv_f32x2 *>(row + half) = pv_f32x2{val[1], val[3]};
        }
    };
    const int64_t ntiles = (M + 63) >> 6;
    const int64_t nw = (int64_t)gridDim.x * ENC_WAVES;
    const int64_t rounds = (ntiles + nw - 1) / nw;   // the same for every wave: barriers stay uniform
    float lmin = INFINITY, lmax = -INFINITY;
    if (nchunks == 1) {
        stage_codebook(0);
        __syncthreads();
    }
    if (vec && (int64_t)blockIdx.x * ENC_WAVES + wave < ntiles) fetch_tile((int64_t)blockIdx.x * ENC_WAVES + wave);
    for (int64_t round = 0; round < rounds; ++round) {
        const int64_t t = round * nw + (int64_t)blockIdx.x * ENC_WAVES + wave;
        const bool active = t < ntiles;
        const int64_t sv = t * 64 + lane;           // this lane's subvector
        if (active) {
            if (vec) {
                commit_tile(t);
                if (t + nw < ntiles) fetch_tile(t + nw);
            } else {
                stage_tile(t);
            }
        }
        float l1 = 0.0f, thr = 0.0f, amin = INFINITY;
        double cum = 0.0;   // torch.cumsum accumulates f32 input in double on the CPU; every output is rounded to f32
        int count = 0;
        // Sweep 1 in its fast form (see shared_quotient below): y = RN(1/l1), T = the double threshold that stands for
        // "(float)cum >= thr"; taken when every lane of the wave qualifies, otherwise the division is done term by term.
        float y = 0.0f;
        double T = 0.0;
        bool fast = false;
        auto sweep = [&](auto first, auto quick) {
            for (int chunk = 0; chunk < nchunks; ++chunk) {
                const int row0 = chunk * chunk_rows;
                if (nchunks > 1) {
                    __syncthreads();   // everyone is done with the previous chunk
                    stage_codebook(row0);
                    __syncthreads();
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (!active) continue;
                const int rblocks = min(chunk_rows, kpad - row0) >> 5;
                for (int rb = 0; rb < rblocks; ++rb) {
                    const float *arow = s_cb + (rb * 32 + j) * stride + h * half;
                    const float *b0 = s_v + j * stride + h * half;
                    const float *b1 = s_v + (32 + j) * stride + h * half;
                    // the first product of a chain takes a literal zero as its addend: no accumulator to clear
                    f32x16 acc0, acc1;
                    {
                        const f32x16 zero = {0};
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(arow);
                        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(b0);
                        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(b1);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x0[0], zero, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x1[0], zero, 0, 0, 0);
#pragma unroll
                        for (int q = 1; q < 4; ++q) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x0[q], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x1[q], acc1, 0, 0, 0);
                        }
                    }
                    for (int k4 = 4; k4 < half; k4 += 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + k4);
                        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(b0 + k4);
                        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(b1 + k4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x0[q], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x1[q], acc1, 0, 0, 0);
                        }
                    }
                    // lane L <- the 32 scores of subvector L: x = rows acc_row(r), y = rows acc_row(r) + 4
                    float x[16], yv[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        x[q] = acc0[q];
                        yv[q] = acc1[q];
                        swap32(x[q], yv[q]);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int part = 0; part < 2; ++part) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = fabsf(part ? yv[4 * g + e] : x[4 * g + e]);   // codeword 32 rb + 8 g + 4 part + e
                                if (decltype(first)::value) {
                                    l1 = l1 + a;
                                    if (e & 1) amin = fminf(fminf(amin, a), fabsf(part ? yv[4 * g + e - 1] : x[4 * g + e - 1]));
                                } else if (decltype(quick)::value) {
                                    cum = cum + (double)shared_quotient(a, l1, y);
                                    count += (cum >= T) ? 0 : 1;
                                } else {
                                    cum = cum + (double)(a / l1);   // the reference divides first (:49,:57)
                                    count += ((float)cum >= thr) ? 0 : 1;
                                }
                            }
                        }
                    }
                }
            }
        };
        sweep(std::true_type{}, std::false_type{});
        {
            const float rr = (active && sv < M) ? ((random_mode == GQ_RANDOM_GIVEN) ? r[sv] : uniform01(seed, (uint64_t)sv))
                                                : 0.0f;
            thr = rr - 1e-5f;
            // An all-zero subvector (l1 == 0) needs no exemption: 0 * (1/0) is the NaN that 0/0 gives, every term counts.
            const bool lane_ok = !active || sv >= M || l1 == 0.0f ||
                                 (K == kpad && l1 >= 0x1p-80f && l1 <= 0x1p20f && amin >= 0x1p-102f);
            fast = __builtin_amdgcn_ballot_w64(!lane_ok) == 0;
            y = 1.0f / l1;
            T = rounds_up_to_threshold(thr);
        }
        if (fast)
            sweep(std::false_type{}, std::true_type{});
        else
            sweep(std::false_type{}, std::false_type{});
        if (!active || sv >= M) continue;
        const int code = count < K - 1 ? count : K - 1;
        float sel = 0.0f;
        {
            const float *row = cdag + (int64_t)code * d;
            const float *v = s_v + lane * stride;
            for (int e = 0; e < d; ++e) sel = __fmaf_rn(row[e], v[(e & 1) * half + (e >> 1)], sel);
        }
        const float sg = (sel > 0.0f) ? 1.0f : ((sel < 0.0f) ? -1.0f : 0.0f);
        const float val = sg * l1;
        codes[sv] = (CodeT)code;
        u[sv] = val;
        lmin = fminf(lmin, val);
        lmax = fmaxf(lmax, val);
    }
    write_minmax_partials(lmin, lmax, partials);
}

// ------------------------------------------------------------------------------------
// ONE sweep + a lane-local walk (round 5; d in {8, 16, 32}, K a multiple of 32 up to 256).
//
// The exact f32 matrix instruction and the VALU never run together on a SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0,
// profiles/r02_b_pvq.txt): the two-sweep kernel above costs its 2 x 128 MFMAs PLUS its ~2,900 VALU instructions per tile.
// The second sweep exists because the walk's terms are |p_k| / l1 and l1 is complete only after the first.  This
// kernel keeps, in sweep 1, the running sum of the |p_k| at every 16-codeword boundary as a DOUBLE (P[b], from f32 sums
// of eight terms: one v_cvt + one v_add_f64 per eight scores).  With l1 known, P[b] / l1 is the walk's running sum at
// that boundary up to
//     eps >= |sum_k RN32(a_k / l1) - (sum_k a_k) / l1|  (<= 2^-24 of the sum)  +  the f32 group sums' rounding
//            (<= 3 * 2^-24 of the sum)  +  the double roundings (< 2^-44),            sum <= 1 + 2^-15:
// eps = 1.0625 * 2^-22.  The lane finds the run b* of 16 codewords whose boundaries bracket its threshold -- and, from the f32 sum
// of the run's first eight terms kept beside P[], the half of it --, recomputes THAT half's 8 projections itself (the same fmaf chain the matrix instruction evaluates, codeword rows read from the staged
// codebook at lane-dependent addresses: the rows of one in-run index lie d + 4 floats apart, sixteen runs on sixteen
// different bank quartets) and walks them from P[b*-1] / l1 with the reference's own terms.  Every comparison of the walk
// is made against T - eps and T + eps: when both agree for the start value and all 8 terms -- and the crossing is where the boundaries said -- the count
// equals the exact walk's (the running sums never decrease).  Otherwise (one lane in ~2^-13, and every lane whose l1
// or terms leave the range the three-operation quotient is proven for) the WAVE walks that lane's K codewords, four per
// lane, with a prefix sum in double that is exact in any order (see there); a term-by-term walk by the lane alone, as
// the VALU kernel does it, stays behind that for quotients below 2^-29 and non-finite l1.  (A first version sent every
// unsettled lane straight to the term-by-term walk: 0.8 % of the waves then ran ~17 us longer than the others, and the
// launch took 194 us instead of 153.)  $GQ_PVQ_EPS (tests) widens eps so that these paths run for most lanes.
// Per tile: 128 MFMAs + ~1,400 VALU instructions instead of 256 + ~2,900.
#ifndef PVQ_DIAG
#define PVQ_DIAG 0   // tools/pvq_variants.py (answers wrong; never shipped): 1 = no walk, 2 = no boundary sums, 4 = no l1 chain, 8 = no swaps,
                     // 16 = clock stamps, 32 = no MFMA, 64 = no run selection / division, 128 = no draw, 256 = no projection of the code, 512 = no staging
#endif
template <int D>
struct PwShape {
    static constexpr int HALF = D / 2;
    static constexpr int RS = D + 4;          // floats between staged rows (both operands): 16 consecutive rows cover the banks
    static constexpr int SI = 16 * RS + 4;    // floats between in-run indices i: codeword 16 hb + i at i * SI + hb * RS
    static constexpr int NQ = D / 4;          // float4s per row
    static constexpr int KQ = HALF / 4;       // float4s per half row
    static constexpr int CB_FLOATS = 16 * SI;
    static constexpr int TILE_FLOATS = 64 * RS;
    static constexpr size_t LDS_BYTES = (size_t)(CB_FLOATS + ENC_WAVES * TILE_FLOATS) * sizeof(float);
};

// Registers: d <= 16 is held to three waves per SIMD (168 VGPRs; the allocator took 169 on its own: 144.8 against 148.1 us);
// d = 32 needs its 200+ (held to three waves it spills: 175 against 129 us).
template <typename CodeT, int D>
__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(D <= 16 ? 3 : 1))) void pvq_encode_walk_kernel(const float *__restrict__ grad,
                                                                     const float *__restrict__ cdag, int64_t M, int K,
                                                                     int random_mode, const float *__restrict__ r,
                                                                     uint64_t seed, CodeT *__restrict__ codes,
                                                                     float *__restrict__ u, float *__restrict__ partials,
                                                                     double eps, PvqResidual rs) {
    using S = PwShape<D>;
    constexpr int HALF = S::HALF, RS = S::RS, SI = S::SI, NQ = S::NQ, KQ = S::KQ;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    float *const s_cb = lds;
    float *const s_v = lds + S::CB_FLOATS + wave * S::TILE_FLOATS;
    const int nb = K >> 5;
    const bool force_slow = eps < 0.0;   // tests: every unsettled lane walks term by term
    eps = fabs(eps);
    // codebook: codeword 16 hb + i as [even elements | odd elements] at i * SI + hb * RS
    for (int idx = threadIdx.x; idx < K * NQ; idx += ENC_THREADS) {
        const int row = idx / NQ, p = idx - row * NQ;
        const f32x4 c = *reinterpret_cast<const f32x4 *>(cdag + (int64_t)row * D + 4 * p);
        float *dst = s_cb + (row & 15) * SI + (row >> 4) * RS + 2 * p;
        *reinterpret_cast<pv_f32x2 *>(dst) = pv_f32x2{c[0], c[2]};
        *reinterpret_cast<pv_f32x2 *>(dst + HALF) = pv_f32x2{c[1], c[3]};
    }
    __syncthreads();
    f32x4 pre[NQ];
    int pre_c1[NQ];
    float pre_n1[NQ];
    auto fetch_tile = [&](int64_t t) {
        const int64_t sv0 = t * 64;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = i * 64 + lane;
            const int rr = q / NQ;
            const int e0 = (q - rr * NQ) * 4;
            const int64_t m = sv0 + rr;
            pre[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            pre_c1[i] = 0;
            pre_n1[i] = 0.0f;
            if (m < M) {
                pre[i] = *reinterpret_cast<const f32x4 *>(grad + m * (int64_t)D + e0);
                if (rs.codes1) {
                    pre_c1[i] = rs.code1_bytes == 1 ? (int)static_cast<const uint8_t *>(rs.codes1)[m]
                                                    : static_cast<const int32_t *>(rs.codes1)[m];
                    pre_n1[i] = rs.norm1[m];
                }
            }
        }
    };
    auto commit_tile = [&](int64_t t) {
        const int64_t sv0 = t * 64;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = i * 64 + lane;
            const int rr = q / NQ;
            const int e0 = (q - rr * NQ) * 4;
            f32x4 val = pre[i];
            if (rs.codes1 && sv0 + rr < M) {
                const f32x4 c = *reinterpret_cast<const f32x4 *>(rs.cb1 + (int64_t)pre_c1[i] * D + e0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dec = c[e] * pre_n1[i];   // stage 1's decoded element (product rounded)
                    val[e] = val[e] - dec;
                }
            }
            float *row = s_v + rr * RS + (e0 >> 1);
            *reinterpret_cast<pv_f32x2 *>(row) = pv_f32x2{val[0], val[2]};
            *reinterpret_cast<pv_f32x2 *>(row + HALF) = pv_f32x2{val[1], val[3]};
        }
    };
    // the lane's own subvector against codeword (b, i): the fmaf chain over ascending elements
    auto project = [&](const float *crow, const f32x4 (&ve)[KQ], const f32x4 (&vo)[KQ]) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const f32x4 ce = *reinterpret_cast<const f32x4 *>(crow + 4 * k);
            const f32x4 co = *reinterpret_cast<const f32x4 *>(crow + HALF + 4 * k);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = __fmaf_rn(ce[q], ve[k][q], acc);
                acc = __fmaf_rn(co[q], vo[k][q], acc);
            }
        }
        return acc;
    };
    const int64_t ntiles = (M + 63) >> 6;
    const int64_t nw = (int64_t)gridDim.x * ENC_WAVES;
    float lmin = INFINITY, lmax = -INFINITY;
    int64_t t = (int64_t)blockIdx.x * ENC_WAVES + wave;
#if PVQ_DIAG & 16
    const uint64_t diag_c0 = __builtin_readcyclecounter(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (t < ntiles) fetch_tile(t);
    for (; t < ntiles; t += nw) {
        const int64_t sv = t * 64 + lane;           // this lane's subvector
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous tile's reads are done
        __builtin_amdgcn_wave_barrier();
        if (!(PVQ_DIAG & 512)) {
            commit_tile(t);
            if (t + nw < ntiles) fetch_tile(t + nw);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- sweep 1: l1 (sequential f32, the reference's sum) and the block boundaries' running sums in double
        float l1 = 0.0f;
        double P[16];
        float S8[16];   // the first eight terms of every run of 16, as they entered the running sum
        {
            f32x4 xb0[KQ], xb1[KQ];
            const float *b0 = s_v + j * RS + h * HALF;
            const float *b1 = s_v + (32 + j) * RS + h * HALF;
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                xb0[k] = *reinterpret_cast<const f32x4 *>(b0 + 4 * k);
                xb1[k] = *reinterpret_cast<const f32x4 *>(b1 + 4 * k);
            }
            double run = 0.0;
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                P[2 * rb] = P[2 * rb + 1] = INFINITY;
                S8[2 * rb] = S8[2 * rb + 1] = 0.0f;
                if (rb < nb) {
                    const float *arow = s_cb + (j & 15) * SI + (2 * rb + (j >> 4)) * RS + h * HALF;
                    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
                    for (int k = 0; k < KQ; ++k) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + 4 * k);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (PVQ_DIAG & 32) {
                                acc0[4 * k + q] = a[q] * xb0[k][q];
                                acc1[4 * k + q] = a[q] * xb1[k][q];
                                continue;
                            }
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], xb0[k][q], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], xb1[k][q], acc1, 0, 0, 0);
                        }
                    }
                    // lane L <- the 32 scores of subvector L: x[4g..] = codewords 8g..8g+3 of the block, yv[4g..] = 8g+4..8g+7
                    float x[16], yv[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        x[q] = acc0[q];
                        yv[q] = acc1[q];
                        if (!(PVQ_DIAG & 8)) swap32(x[q], yv[q]);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (PVQ_DIAG & 4) {
                            l1 = l1 + fabsf(x[4 * g]) + fabsf(yv[4 * g + 3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) l1 = l1 + fabsf(x[4 * g + e]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) l1 = l1 + fabsf(yv[4 * g + e]);
                        }
                        if (PVQ_DIAG & 2) {
                            if (g & 1) P[2 * rb + (g >> 1)] = (double)l1;
                            continue;
                        }
                        const float s8 = ((fabsf(x[4 * g]) + fabsf(x[4 * g + 1])) + (fabsf(x[4 * g + 2]) + fabsf(x[4 * g + 3]))) +
                                         ((fabsf(yv[4 * g]) + fabsf(yv[4 * g + 1])) + (fabsf(yv[4 * g + 2]) + fabsf(yv[4 * g + 3])));
                        run = run + (double)s8;
                        if (g & 1) P[2 * rb + (g >> 1)] = run;
                        else S8[2 * rb + (g >> 1)] = s8;
                    }
                }
            }
        }
        const bool mine = sv < M;
        const float rr = (PVQ_DIAG & 128) ? 0.5f : mine ? ((random_mode == GQ_RANDOM_GIVEN) ? r[sv] : uniform01(seed, (uint64_t)sv)) : 0.0f;
        const float thr = rr - 1e-5f;
        const double T = (PVQ_DIAG & 128) ? 0.5 : rounds_up_to_threshold(thr);   // (float)x >= thr  <=>  x >= T
        // ---- the block whose boundaries bracket the threshold, and the walk's value at its start
        const double U = T * (double)l1;
        double startP = 0.0;
        float mid = S8[0];
        int bstar = 0;
#pragma unroll
        for (int b = 0; b < ((PVQ_DIAG & 64) ? 1 : 15); ++b) {
            const bool below = (b < 2 * nb - 1) && (P[b] < U);
            startP = below ? P[b] : startP;
            mid = below ? S8[b + 1] : mid;
            bstar = below ? b + 1 : bstar;
        }
        // ... and the half of that run: the running sum after its first eight terms is startP + mid, as sweep 1 formed it
        {
            const double midP = startP + (double)mid;
            const bool second = midP < U;
            startP = second ? midP : startP;
            bstar = 2 * bstar + (second ? 1 : 0);   // from here on: a run of EIGHT codewords
        }
        f32x4 ve[KQ], vo[KQ];
        {
            const float *v = s_v + lane * RS;
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                ve[k] = *reinterpret_cast<const f32x4 *>(v + 4 * k);
                vo[k] = *reinterpret_cast<const f32x4 *>(v + HALF + 4 * k);
            }
        }
        const double Tlo = T - eps, Thi = T + eps;
        double cum = (PVQ_DIAG & 64) ? startP * (double)l1 : startP / (double)l1;
        const bool start_ok = cum < Tlo;
        const float y = 1.0f / l1;
        int cnt_lo = 0, cnt_hi = 0;
        float amin = INFINITY;
        {
            // the run's 8 codeword rows, each fetched one projection ahead of its use
            const float *crow = s_cb + (bstar >> 1) * RS + (bstar & 1) * 8 * SI;
            f32x4 ce[2][KQ], co[2][KQ];
            auto fetch_row = [&](int i, int buf) {
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    ce[buf][k] = *reinterpret_cast<const f32x4 *>(crow + i * SI + 4 * k);
                    co[buf][k] = *reinterpret_cast<const f32x4 *>(crow + i * SI + HALF + 4 * k);
                }
            };
            fetch_row(0, 0);
#pragma unroll
            for (int i = 0; i < ((PVQ_DIAG & 1) ? 1 : 8); ++i) {
                if (i + 1 < 8) fetch_row(i + 1, (i + 1) & 1);
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc = __fmaf_rn(ce[i & 1][k][q], ve[k][q], acc);
                        acc = __fmaf_rn(co[i & 1][k][q], vo[k][q], acc);
                    }
                }
                const float a = fabsf(acc);
                amin = fminf(amin, a);
                cum = cum + (double)shared_quotient(a, l1, y);
                cnt_lo += (cum < Tlo) ? 1 : 0;
                cnt_hi += (cum < Thi) ? 1 : 0;
            }
        }
        int count = bstar * 8 + cnt_lo;
        bool settled = l1 >= 0x1p-80f && l1 <= 0x1p20f && amin >= 0x1p-102f && start_ok && cnt_lo == cnt_hi &&
                       (cnt_lo < 8 || bstar == 4 * nb - 1);
        if (l1 == 0.0f) {   // an all-zero subvector: every term is 0 / 0, every comparison with NaN fails, all K terms count
            count = K;
            settled = true;
        }
        // ---- the unsettled lanes (one in ~2^-13), one at a time by the whole wave: lane L takes codewords 4L .. 4L+3 of
        // the lane's subvector, divides as the reference does, and the running sums come from a wave-wide prefix sum in
        // double.  That equals the sequential sum bit for bit when every non-zero quotient is at least 2^-29: all terms
        // and all partial sums are then multiples of 2^-52 below 2, every double addition is exact in any order.
        // Otherwise (and for a non-finite l1) the lane walks its K terms alone, one after the other.
        uint64_t todo = (PVQ_DIAG & ~16) ? 0 : __builtin_amdgcn_ballot_w64(mine && !settled);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const float l1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l1), src));
            const float thrs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(thr), src));
            f32x4 se[KQ], so[KQ];
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float e_own = ve[k][q], o_own = vo[k][q];   // (scalars first: a bit_cast of the vector element
                    se[k][q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_own), src));   // read element 0 four times)
                    so[k][q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o_own), src));
                }
            }
            double inc[4];
            double tot = 0.0;
            bool exact = !force_slow;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = 4 * lane + c;
                float q = 0.0f;
                if (k < K) q = fabsf(project(s_cb + (k & 15) * SI + (k >> 4) * RS, se, so)) / l1s;
                exact = exact && (q == 0.0f || (q >= 0x1p-29f && q <= 2.0f));
                tot = tot + (double)q;
                inc[c] = tot;
            }
            double before = tot;   // inclusive prefix over the lanes, then made exclusive
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double up = __shfl_up(before, off, 64);
                if (lane >= off) before = before + up;
            }
            before = before - tot;   // exact: both are multiples of 2^-52 below 2
            int n = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) n += (4 * lane + c < K && !((float)(before + inc[c]) >= thrs)) ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
            if (__builtin_amdgcn_ballot_w64(!exact) == 0) {
                if (lane == src) count = n;
            } else if (lane == src) {
                // term by term over all K codewords, the reference's own arithmetic (as pvq_encode_kernel)
                double c2 = 0.0;
                int n2 = 0;
                for (int k = 0; k < K; ++k) {
                    const float a = fabsf(project(s_cb + (k & 15) * SI + (k >> 4) * RS, ve, vo));
                    c2 = c2 + (double)(a / l1);
                    n2 += ((float)c2 >= thr) ? 0 : 1;
                }
                count = n2;
            }
        }
        if (!mine) continue;
        const int code = count < K - 1 ? count : K - 1;
        const float sel = (PVQ_DIAG & 256) ? ve[0][0] : project(s_cb + (code & 15) * SI + (code >> 4) * RS, ve, vo);
        const float sg = (sel > 0.0f) ? 1.0f : ((sel < 0.0f) ? -1.0f : 0.0f);
        const float val = sg * l1;
        codes[sv] = (CodeT)code;
        u[sv] = val;
        lmin = fminf(lmin, val);
        lmax = fmaxf(lmax, val);
    }
    write_minmax_partials(lmin, lmax, partials);
#if PVQ_DIAG & 16
    if (blockIdx.x == 1 && threadIdx.x == 0) {   // shader cycles and 100 MHz ticks of this workgroup's run, for tools/pvq_ab.py
        u[0] = (float)(__builtin_readcyclecounter() - diag_c0);
        u[1] = (float)(__builtin_amdgcn_s_memrealtime() - diag_r0);
    }
#endif
}

template <typename CodeT, int D>
static int launch_pvq_walk(const float *grad, const float *cdag, int64_t M, int K, int random_mode, const float *r,
                           uint64_t seed, CodeT *codes, float *u, float *ws, hipStream_t st, PvqResidual rs) {
    static const double eps = [] {
        const char *e = getenv("GQ_PVQ_EPS");   // tests: a wide window sends most lanes through the term-by-term walk
        const double v = e ? atof(e) : 0.0;   // (negative: ... and those walk term by term instead of as a wave)
        return fabs(v) > 0x1.1p-22 ? v : 0x1.1p-22;
    }();
    constexpr size_t lds_bytes = PwShape<D>::LDS_BYTES;
    static const int bpc = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pvq_encode_walk_kernel<CodeT, D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
        return resident_blocks_per_cu(pvq_encode_walk_kernel<CodeT, D>, ENC_THREADS, lds_bytes);
    }();
    const int64_t ntiles = (M + 63) / 64;
    int64_t blocks = (ntiles + ENC_WAVES - 1) / ENC_WAVES;
    static const int bpc_env = getenv("GQ_PVQ_BPC") ? atoi(getenv("GQ_PVQ_BPC")) : 0;   // experiments: workgroups per CU
    int64_t cap = (int64_t)cu_count() * (bpc_env > 0 ? bpc_env : bpc);
    if (getenv("GQ_PVQ_BPC")) fprintf(stderr, "pvq walk: occupancy API says %d workgroups per CU\n", bpc);
    if (cap > GQ_MAIN_PARTIALS) cap = GQ_MAIN_PARTIALS;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(pvq_encode_walk_kernel<CodeT, D>), dim3((unsigned)blocks), dim3(ENC_THREADS),
                       lds_bytes, st, grad, cdag, M, K, random_mode, r, seed, codes, u, ws, eps, rs);
    GQ_CHECK_LAUNCH("gq_pvq_encode (one sweep + walk)");
    return GQ_OK;
}

// The one-sweep kernel serves d in {8, 16, 32} with K = 32 ... 256 in whole blocks and 16-byte aligned operands.
static bool walk_serves(const float *grad, const float *cdag, int d, int K, const PvqResidual &rs) {
    static const bool off = getenv("GQ_PVQ_TWO_SWEEPS") != nullptr;   // A/B and cross-check: the two-sweep kernel
    if (off || (d != 8 && d != 16 && d != 32) || K < 32 || K > 256 || (K & 31)) return false;
    if ((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(cdag)) & 15) return false;
    if (rs.codes1 && (reinterpret_cast<uintptr_t>(rs.cb1) & 15)) return false;
    return true;
}

template <typename CodeT>
static int launch_pvq_lds(const float *grad, const float *cdag, int64_t M, int d, int K, int random_mode, const float *r,
                          uint64_t seed, CodeT *codes, float *u, float *ws, hipStream_t st, bool *done,
                          PvqResidual rs = PvqResidual{nullptr, 0, nullptr, nullptr}) {
    int dpad = 0, chunk_rows = 0;
    size_t lds_bytes = 0;
    *done = false;
    if (walk_serves(grad, cdag, d, K, rs)) {
        *done = true;
        switch (d) {
            case 8: return launch_pvq_walk<CodeT, 8>(grad, cdag, M, K, random_mode, r, seed, codes, u, ws, st, rs);
            case 16: return launch_pvq_walk<CodeT, 16>(grad, cdag, M, K, random_mode, r, seed, codes, u, ws, st, rs);
            default: return launch_pvq_walk<CodeT, 32>(grad, cdag, M, K, random_mode, r, seed, codes, u, ws, st, rs);
        }
    }
    if (!lds_plan(d, K, &dpad, &chunk_rows, &lds_bytes)) return GQ_OK;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pvq_encode_lds_kernel<CodeT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
    const int bpc = resident_blocks_per_cu(pvq_encode_lds_kernel<CodeT>, ENC_THREADS, lds_bytes);
    const int64_t ntiles = (M + 63) / 64;
    int64_t blocks = (ntiles + ENC_WAVES - 1) / ENC_WAVES;
    int64_t cap = (int64_t)cu_count() * bpc;
    if (cap > GQ_MAIN_PARTIALS) cap = GQ_MAIN_PARTIALS;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(pvq_encode_lds_kernel<CodeT>), dim3((unsigned)blocks), dim3(ENC_THREADS), lds_bytes,
                       st, grad, cdag, M, d, K, random_mode, r, seed, codes, u, ws, dpad, chunk_rows, rs);
    GQ_CHECK_LAUNCH("gq_pvq_encode (mfma)");
    *done = true;
    return GQ_OK;
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
    static const bool valu_only = getenv("GQ_PVQ_VALU") != nullptr;   // tests: the VALU cross-check kernel
    if (!valu_only) {
        bool done = false;
        const int rc = launch_pvq_lds<CodeT>(grad, cdag, M, d, K, random_mode, r, seed, codes, u, ws, st, &done);
        if (rc != GQ_OK || done) return rc;
    }
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
                         const gq_pvq_stage1 *stage1, void *stream) {
    if (M < 1 || d < 1 || K < 1) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: bad sizes");
    if (!grad || !c_dagger || !codes || !u || !workspace) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: null pointer");
    if (random_mode != GQ_RANDOM_GIVEN && random_mode != GQ_RANDOM_DEVICE)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: random_mode must be GIVEN or DEVICE (the sampler needs draws)");
    if (random_mode == GQ_RANDOM_GIVEN && !r) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: r is null");
    if (code_bytes != 1 && code_bytes != 4) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: code_bytes must be 1 or 4");
    if (code_bytes == 1 && K > 256) return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: uint8 codes need K <= 256");
    hipStream_t st = gq::as_stream(stream);
    if (!stage1) {
        if (code_bytes == 1)
            return gq::dispatch_pvq<uint8_t>(grad, c_dagger, M, d, K, random_mode, r, seed, static_cast<uint8_t *>(codes), u,
                                             workspace, st);
        return gq::dispatch_pvq<int32_t>(grad, c_dagger, M, d, K, random_mode, r, seed, static_cast<int32_t *>(codes), u,
                                         workspace, st);
    }
    // second stage of the ResidualCompressor: the tile is staged as grad - codebook1[codes1] * norm1
    if (!stage1->codes1 || !stage1->norm1 || !stage1->codebook1)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: stage1 holds a null pointer");
    if (stage1->code1_bytes != 1 && stage1->code1_bytes != 4)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_pvq_encode: stage1->code1_bytes must be 1 or 4");
    const gq::PvqResidual rs = {stage1->codes1, stage1->code1_bytes, stage1->norm1, stage1->codebook1};
    bool done = false;
    int rc;
    if (code_bytes == 1)
        rc = gq::launch_pvq_lds<uint8_t>(grad, c_dagger, M, d, K, random_mode, r, seed, static_cast<uint8_t *>(codes), u,
                                         workspace, st, &done, rs);
    else
        rc = gq::launch_pvq_lds<int32_t>(grad, c_dagger, M, d, K, random_mode, r, seed, static_cast<int32_t *>(codes), u,
                                         workspace, st, &done, rs);
    if (rc != GQ_OK) return rc;
    if (!done) return gq::fail(GQ_ERR_UNSUPPORTED, "gq_pvq_encode: d = %d does not fit the LDS-staged kernel (stage1 form)", d);
    return GQ_OK;
}
