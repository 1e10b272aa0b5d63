// HSQ encode kernels for gfx950 (MI355X).
//
// Replaces nearest_neighbor_compressor.py:65-73 of the reference
// (view(-1,d) -> torch.mm -> abs -> argmax -> gather).  The reference's inner
// products are, bit for bit, the ascending chain acc = fmaf(c[j], v[j], acc) from
// acc = 0.  gfx950's f32-input MFMA (v_mfma_f32_32x32x2_f32)
// evaluates exactly that chain over its k index -- one rounding per product, no
// wider internal accumulator -- so the scores come out of the matrix cores
// bit-identical to the reference while running at the full f32 rate.
//
// Implementations, all exact (hsq_encode_pf.hip adds the bf16 prefilter path for d16/K256):
//   d16k256  codebook (256x16) held in 64 VGPRs per lane as MFMA A fragments;
//            gradient subvectors are the B operand (subvector = MFMA column = lane),
//            so the 256-way argmax is lane-local plus ONE cross-half exchange.
//   lds      same MFMA formulation for any (d <= 128, K); codebook (chunked if need be) and tiles in LDS.
//   generic  same MFMA formulation for any (d, K); fragments come from L1/L2 (fallback for d > 128).
//   valu     one subvector per lane, codebook broadcast from LDS, __fmaf_rn chain,
//            wave-level min/max by shuffles; kept as the cross-check of the MFMA path.
#include "hsq_encode_common.hpp"
#include <type_traits>

namespace gq {

// ------------------------------------------------------------------------------------
// d = 16, K = 256: the BASELINE configuration.
//
// One wave handles a tile of 64 subvectors (4 KiB of gradient) per iteration as two
// 32-column MFMA blocks.  Lane (j = lane&31, h = lane>>5) loads floats [8h, 8h+8) of
// subvector j (two dwordx4; the wave covers the 2 KiB block exactly once), four
// v_permlane32_swap turn that into the B fragments b[ks] = v[2ks + h].  For each of the
// 8 row blocks of 32 codewords the 8 chained MFMAs produce, in every lane, 16 finished
// scores of ITS subvector; the running best is updated in registers.
// HBM traffic per subvector: 64 B read, 1 B code + 4 B u written.
// ------------------------------------------------------------------------------------
// 8 chained MFMAs = the 32 scores x 32 subvectors of one row block (ascending k: the fmaf chain).
__device__ __forceinline__ f32x16 score_chain(const float (&a)[8], const float (&b)[8]) {
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[ks], acc, 0, 0, 0);
    return acc;
}

// First-max over the 16 finished scores of one row block, then into the running best.
// Inside a row block the candidates carry their ROW (0..27: inline constants, no VGPRs for
// index literals); the block base rb*32 is added once per block.  Two independent chains
// (registers 0-7 / 8-15; every index of the first is below every index of the second) keep
// the compare -> select dependency from serialising the VALU.
struct Best {
    float v;
    int i;
};
__device__ __forceinline__ void argmax_block(Best &best, const f32x16 &acc, int rb) {
    float lv = acc[0], hv = acc[8];
    int li = acc_row(0), hi = acc_row(8);
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        take_if_greater(lv, li, acc[r], acc_row(r));
        take_if_greater(hv, hi, acc[r + 8], acc_row(r + 8));
    }
    take_if_greater(lv, li, hv, hi);            // strict: ties stay with the lower rows
    take_if_greater(best.v, best.i, lv, li + rb * 32);  // strict: ties stay with earlier blocks
}

template <typename CodeT>
__global__ __launch_bounds__(ENC_THREADS, 2) void hsq_encode_d16k256_kernel(const float *__restrict__ grad,
                                                                           const float *__restrict__ cb, int64_t M,
                                                                           CodeT *__restrict__ codes,
                                                                           float *__restrict__ u,
                                                                           float *__restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;

    // A fragments: a[rb][ks] = codebook[rb*32 + j][2*ks + h]   (64 VGPRs)
    float a[8][8];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a[rb][ks] = cb[(rb * 32 + j) * 16 + 2 * ks + h];
    }

    const int64_t ntiles = (M + 63) >> 6;
    const int64_t nw = (int64_t)gridDim.x * ENC_WAVES;
    int64_t t = (int64_t)blockIdx.x * ENC_WAVES + wave;

    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)
    f32x4 cur[4], nxt[4];

    auto load_tile = [&](int64_t tile, f32x4(&dst)[4]) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            int64_t sv = tile * 64 + blk * 32 + j;
            sv = sv < M ? sv : M - 1;  // tail: re-read the last subvector, result is masked
            const f32x4 *p = reinterpret_cast<const f32x4 *>(grad + sv * 16 + 8 * h);
            dst[2 * blk] = p[0];
            dst[2 * blk + 1] = p[1];
        }
    };

    if (t < ntiles) load_tile(t, cur);
    for (; t < ntiles; t += nw) {
        const int64_t tn = t + nw;
        if (tn < ntiles) load_tile(tn, nxt);  // prefetch the next tile under this tile's MFMAs

        // B fragments of both 32-subvector blocks: b[blk][ks] = v[2*ks + h]
        float b[2][8];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float r0 = cur[2 * blk][0], r1 = cur[2 * blk][1], r2 = cur[2 * blk][2], r3 = cur[2 * blk][3];
            float r4 = cur[2 * blk + 1][0], r5 = cur[2 * blk + 1][1], r6 = cur[2 * blk + 1][2],
                  r7 = cur[2 * blk + 1][3];
            swap32(r0, r1);
            swap32(r2, r3);
            swap32(r4, r5);
            swap32(r6, r7);
            b[blk][0] = r0; b[blk][1] = r2; b[blk][2] = r4; b[blk][3] = r6;
            b[blk][4] = r1; b[blk][5] = r3; b[blk][6] = r5; b[blk][7] = r7;
        }

        // Software pipeline over the 16 (block, row block) chains: the MFMAs of chain c+1 are
        // issued between the VALU argmax instructions of chain c (two accumulator sets).
        Best best[2] = {{0.0f, 0}, {0.0f, 0}};
        f32x16 acc = score_chain(a[0], b[0]);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            f32x16 nacc;
            if (c + 1 < 16) nacc = score_chain(a[(c + 1) & 7], b[(c + 1) >> 3]);
            argmax_block(best[c >> 3], acc, c & 7);
            if (c + 1 < 16) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // six VALU (2 scores)
                }
                acc = nacc;
            }
        }

        float bv[2];
        int bi[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            bv[blk] = best[blk].v;
            bi[blk] = best[blk].i + 4 * h;
        }

        // Cross-half exchange: afterwards lane L holds both half-candidates of subvector
        // t*64 + L  (lanes 0..31: first block, lanes 32..63: second block).
        swap32(bv[0], bv[1]);
        swap32(bi[0], bi[1]);
        const float a0 = fabsf(bv[0]), a1 = fabsf(bv[1]);
        const bool take1 = (a1 > a0) || (a1 == a0 && bi[1] < bi[0]);
        float val = take1 ? bv[1] : bv[0];
        int idx = take1 ? bi[1] : bi[0];

        const int64_t sv = t * 64 + lane;
        // non-finite inputs (hsq_encode_common.hpp): x * 0 is NaN exactly for NaN and the infinities
        {
            float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __fmaf_rn(cur[0][e], 0.0f, z0);
                z0 = __fmaf_rn(cur[1][e], 0.0f, z0);
                z1 = __fmaf_rn(cur[2][e], 0.0f, z1);
                z1 = __fmaf_rn(cur[3][e], 0.0f, z1);
            }
            const uint64_t b0 = __ballot(z0 != z0), b1 = __ballot(z1 != z1);   // lanes (j, h): halves of subvector j
            uint64_t todo = ((b0 | (b0 >> 32)) & 0xFFFFFFFFull) | (((b1 | (b1 >> 32)) & 0xFFFFFFFFull) << 32);
            todo &= __ballot(sv < M);
            while (todo) {
                const int fl = __builtin_ctzll(todo);
                todo &= todo - 1;
                const float *v = grad + (t * 64 + fl) * 16;
                float bvv;
                int bii;
                nonfinite_argmax(256, 16, [&](int k, int e) { return cb[k * 16 + e]; }, [&](int e) { return v[e]; }, bvv, bii);
                if (lane == fl) {
                    val = bvv;
                    idx = bii;
                }
                sawnan = sawnan || (bvv != bvv);
            }
        }
        if (sv < M) {
            codes[sv] = (CodeT)idx;
            u[sv] = val;
            lmin = fminf(lmin, val);
            lmax = fmaxf(lmax, val);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
    write_minmax_partials(lmin, lmax, partials, sawnan);
}

// ------------------------------------------------------------------------------------
// Generic (d, K): same MFMA formulation; A/B fragments are fetched per MFMA from
// global memory (the codebook and the 64-subvector tile stay L1/L2 resident).
// k beyond d and codewords beyond K are fed as zeros: fma(0,0,acc) == acc exactly
// (acc is never -0 in this chain), and padded codewords are excluded from the argmax.
// ------------------------------------------------------------------------------------
template <typename CodeT>
__global__ __launch_bounds__(ENC_THREADS) void hsq_encode_generic_kernel(const float *__restrict__ grad,
                                                                        const float *__restrict__ cb, int64_t M,
                                                                        int d, int K, CodeT *__restrict__ codes,
                                                                        float *__restrict__ u,
                                                                        float *__restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ksteps = (d + 1) >> 1;
    const int rblocks = (K + 31) >> 5;

    const int64_t ntiles = (M + 63) >> 6;
    const int64_t nw = (int64_t)gridDim.x * ENC_WAVES;
    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)

    for (int64_t t = (int64_t)blockIdx.x * ENC_WAVES + wave; t < ntiles; t += nw) {
        float bv[2];
        int bi[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            int64_t sv = t * 64 + blk * 32 + j;
            sv = sv < M ? sv : M - 1;
            const float *v = grad + sv * (int64_t)d;
            float best_v = 0.0f;
            int best_i = 0;
            for (int rb = 0; rb < rblocks; ++rb) {
                const int row = rb * 32 + j;
                const float *c = cb + (int64_t)(row < K ? row : K - 1) * d;
                const bool row_ok = row < K;
                f32x16 acc = {0};
                for (int ks = 0; ks < ksteps; ++ks) {
                    const int k = 2 * ks + h;
                    const bool k_ok = k < d;
                    const float av = (row_ok && k_ok) ? c[k_ok ? k : 0] : 0.0f;
                    const float bvv = k_ok ? v[k_ok ? k : 0] : 0.0f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = rb * 32 + acc_row(r);
                    if (idx + 4 * h < K) take_if_greater(best_v, best_i, acc[r], idx);
                }
            }
            bv[blk] = best_v;
            bi[blk] = best_i + 4 * h;
        }
        swap32(bv[0], bv[1]);
        swap32(bi[0], bi[1]);
        // a half whose candidate index is out of range (K < 8) never saw a valid codeword
        const bool v0 = bi[0] < K, v1 = bi[1] < K;
        const float a0 = fabsf(bv[0]), a1 = fabsf(bv[1]);
        const bool take1 = v1 && (!v0 || (a1 > a0) || (a1 == a0 && bi[1] < bi[0]));
        float val = take1 ? bv[1] : bv[0];
        int idx = take1 ? bi[1] : bi[0];
        const int64_t sv = t * 64 + lane;
        {   // non-finite inputs (hsq_encode_common.hpp): this lane's own subvector, re-read (L1 / L2)
            float z = 0.0f;
            if (sv < M)
                for (int e = 0; e < d; ++e) z = __fmaf_rn(grad[sv * (int64_t)d + e], 0.0f, z);
            uint64_t todo = __ballot(z != z);
            while (todo) {
                const int fl = __builtin_ctzll(todo);
                todo &= todo - 1;
                const float *v = grad + (t * 64 + fl) * (int64_t)d;
                float bvv;
                int bii;
                nonfinite_argmax(K, d, [&](int k, int e) { return cb[(int64_t)k * d + e]; }, [&](int e) { return v[e]; }, bvv, bii);
                if (lane == fl) {
                    val = bvv;
                    idx = bii;
                }
                sawnan = sawnan || (bvv != bvv);
            }
        }
        if (sv < M) {
            codes[sv] = (CodeT)idx;
            u[sv] = val;
            lmin = fminf(lmin, val);
            lmax = fmaxf(lmax, val);
        }
    }
    write_minmax_partials(lmin, lmax, partials, sawnan);
}

// ------------------------------------------------------------------------------------
// Any (d <= 128, K): exact f32 MFMA with BOTH operands staged in LDS.
//
// The codebook (or a chunk of `chunk_rows` rows of it when K*d*4 does not fit) and each wave's
// 64-subvector tile are written to LDS once, de-interleaved -- a row is stored as
// [ even k | odd k ], each run padded with zeros to dpad/2 floats (dpad = d rounded up to 8) --
// so that lane (j, h) of v_mfma_f32_32x32x2_f32 finds its operands a[ks] = row[2ks + h] as ONE
// contiguous run and fetches four k-steps per ds_read_b128.  Rows are dpad + 4 floats apart:
// (dpad+4)/4 is odd, so the 16 lanes of a b128 phase hit 16 different bank groups.
// Zero padding is exact: fma(0, 0, acc) == acc (acc is never -0 in this chain); padded codewords
// are excluded from the argmax.  Two independent accumulator chains (the tile's two 32-column
// blocks) share every A fragment.  When the codebook is chunked the workgroup re-stages the next
// chunk between barriers and each wave carries its tile's running (best, index) in registers.
// ------------------------------------------------------------------------------------

// BATCHED: the tiles are those of a segment table (include/gq_hsq.h, gq_hsq_encode_batched_any): tile t belongs to
// tensor tile_seg[t]; codes go into the tensor's section of `wire`, u into the padded u_flat, and (min,max)
// of u is folded into seg_minmax per tensor.  EF: a tile is read as grad + ef_scale*error and written back.
struct LdsBatch {
    const int64_t *seg_table;
    const int32_t *tile_seg;
    uint8_t *wire;
    unsigned *seg_minmax;
    float ef_scale;
};

__device__ __forceinline__ unsigned lds_order_map(float f) {
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

template <typename CodeT, bool BATCHED = false, bool EF = false>
__global__ __launch_bounds__(ENC_THREADS) void hsq_encode_lds_kernel(const float *__restrict__ grad,
                                                                    const float *__restrict__ cb, int64_t M, int d,
                                                                    int K, CodeT *__restrict__ codes,
                                                                    float *__restrict__ u,
                                                                    float *__restrict__ partials, int dpad,
                                                                    int chunk_rows, LdsBatch bt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = (int)(blockDim.x >> 6);   // 4; 2 or 1 when a 4-wave workgroup's tiles do not fit the LDS (d > ~110)
    const int j = lane & 31, h = lane >> 5;
    const int stride = dpad + LDS_ROW_PAD, half = dpad >> 1;
    float *const s_cb = lds;
    float *const s_v = lds + (size_t)chunk_rows * stride + (size_t)wave * 64 * stride;
    const int kpad = (K + 31) & ~31;
    const int nchunks = (kpad + chunk_rows - 1) / chunk_rows;
    const float inv_dpad = 1.0f / (float)dpad;

    // rows [row0, row0 + chunk_rows) of the codebook -> s_cb (all threads)
    auto stage_codebook = [&](int row0) {
        const int total = chunk_rows * dpad;
        for (int i = threadIdx.x; i < total; i += (int)blockDim.x) {
            const int r = (int)(((float)i + 0.5f) * inv_dpad);   // exact for these ranges (i < 2^20, dpad <= 128)
            const int e = i - r * dpad;
            const int row = row0 + r;
            const float val = (row < K && e < d) ? cb[(int64_t)row * d + e] : 0.0f;
            s_cb[r * stride + (e & 1) * half + (e >> 1)] = val;
        }
    };
    // this wave's tile (64 subvectors) -> s_v; subvectors beyond M are zeros (masked at the store)
    // BATCHED: the tile's tensor (wave-uniform): its record, the tile's first subvector inside it
    const int64_t *rec = nullptr;
    int64_t seg_m = M, seg_sv0 = 0;
    int seg = -1, cur_seg = -1;
    auto stage_tile = [&](int64_t t) {
        const int total = 64 * dpad;
        int64_t sv0 = t * 64, m = M;
        const float *g = grad;
        float *gw = nullptr;
        const float *err = nullptr;
        if (BATCHED) {
            seg = bt.tile_seg[t];
            rec = bt.seg_table + 8 * (int64_t)seg;
            m = seg_m = rec[1];
            sv0 = seg_sv0 = (t - rec[2]) * 64;
            g = gw = reinterpret_cast<float *>(rec[0]);
            if (EF) err = reinterpret_cast<const float *>(rec[7]);
        }
        if ((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | (EF && err ? reinterpret_cast<uintptr_t>(err) : 0)) & 15) == 0) {
            // the tile is 64*d contiguous floats: 16-byte loads, four in flight per lane (one element at a time
            // the staging was a chain of dependent 4-byte round trips and the whole kernel waited on it).
            // A dwordx4 holds elements e..e+3 of one row: the even ones go to [e/2, e/2+1] of the row's first
            // run, the odd ones to the same place in its second run.  The zero padding [d, dpad) is never
            // written here (zeroed once at kernel start).
            const int upr = d >> 2, units = 64 * upr;          // 16-byte units per row / per tile
            const float inv_upr = 1.0f / (float)upr;
            const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g + sv0 * (int64_t)d);
            f32x4 *gw4 = reinterpret_cast<f32x4 *>(gw + sv0 * (int64_t)d);
            const f32x4 *e4 = reinterpret_cast<const f32x4 *>(err + sv0 * (int64_t)d);
            const int64_t rows_in = m - sv0;                    // rows of the tile inside the tensor (may exceed 64)
            for (int i0 = lane; i0 < units; i0 += 256) {
                f32x4 p[4];
                int r[4], q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + 64 * k;
                    r[k] = (int)(((float)i + 0.5f) * inv_upr);
                    q[k] = i - r[k] * upr;
                    const bool in = i < units && r[k] < rows_in;
                    p[k] = in ? g4[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (EF && err && in) {   // ps_quantizer.py:35, in place: product rounded, then the add
                        const f32x4 prod = e4[i] * bt.ef_scale;
                        p[k] = p[k] + prod;
                        gw4[i] = p[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (i0 + 64 * k < units) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        float *row = s_v + r[k] * stride + 2 * q[k];
                        *reinterpret_cast<f32x2 *>(row) = f32x2{p[k][0], p[k][2]};
                        *reinterpret_cast<f32x2 *>(row + half) = f32x2{p[k][1], p[k][3]};
                    }
                }
            }
            return;
        }
        for (int i = lane; i < total; i += 64) {
            const int r = (int)(((float)i + 0.5f) * inv_dpad);
            const int e = i - r * dpad;
            const bool in = sv0 + r < m && e < d;
            const int64_t at = (sv0 + r) * (int64_t)d + e;
            float val = in ? g[at] : 0.0f;
            if (EF && err && in) {   // ps_quantizer.py:35, in place: product rounded, then the add
                const float p = bt.ef_scale * err[at];
                val = val + p;
                gw[at] = val;
            }
            s_v[r * stride + (e & 1) * half + (e >> 1)] = val;
        }
    };
    // the padding [d, dpad) of this wave's tile rows: zero, once (the 16-byte staging path never writes it)
    for (int i = lane; i < 64 * stride; i += 64) s_v[i] = 0.0f;
    auto flush_minmax = [&](float &lmin, float &lmax) {   // this wave's running (min,max) -> its tensor's words
        const float lo = wave_min(lmin), hi = wave_max(lmax);
        if (lane == 0 && cur_seg >= 0 && lo <= hi) {
            // look before the atomic (hsq_encode_pf.hip): the words only move towards the extremes
            unsigned *mm = bt.seg_minmax + 2 * cur_seg;
            const unsigned mlo = lds_order_map(lo), mhi = lds_order_map(hi);
            if (mlo < __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(mm, mlo);
            if (mhi > __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mm + 1, mhi);
        }
        lmin = INFINITY;
        lmax = -INFINITY;
    };

    const int64_t ntiles = (M + 63) >> 6;
    const int64_t nw = (int64_t)gridDim.x * nwaves;
    const int64_t rounds = (ntiles + nw - 1) / nw;   // the same for every wave of the grid: barriers stay uniform
    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)
    if (nchunks == 1) {
        stage_codebook(0);
        __syncthreads();
    }
    for (int64_t round = 0; round < rounds; ++round) {
        // BATCHED: a wave takes ONE contiguous run of `rounds` tiles, so that its running (min,max) stays with a
        // tensor (visited round-robin every tile changed tensor and paid two atomics on the same few cache
        // lines: 46,000 tiles of the ResNet-50 list at d = 8 took 350 us, most of it queueing there)
        const int64_t wid = (int64_t)blockIdx.x * nwaves + wave;
        const int64_t t = BATCHED ? wid * rounds + round : round * nw + wid;
        const bool active = t < ntiles;
        if (active) stage_tile(t);
        Best best[2] = {{0.0f, 0}, {0.0f, 0}};
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int row0 = chunk * chunk_rows;
            if (nchunks > 1) {
                __syncthreads();   // everyone is done with the previous chunk
                stage_codebook(row0);
                __syncthreads();   // chunk visible
            } else {
                // the tile is private to the wave and LDS operations of one wave complete in order:
                // only the compiler has to be kept from moving the reads above the staging writes
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (!active) continue;
            const int rblocks = min(chunk_rows, kpad - row0) >> 5;
            for (int rb = 0; rb < rblocks; ++rb) {
                const float *arow = s_cb + (rb * 32 + j) * stride + h * half;
                const float *b0 = s_v + j * stride + h * half;
                const float *b1 = s_v + (32 + j) * stride + h * half;
                f32x16 acc0 = {0}, acc1 = {0};
                for (int k4 = 0; k4 < half; k4 += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + k4);
                    const f32x4 x0 = *reinterpret_cast<const f32x4 *>(b0 + k4);
                    const f32x4 x1 = *reinterpret_cast<const f32x4 *>(b1 + k4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x0[q], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x1[q], acc1, 0, 0, 0);
                    }
                }
                // Padded codewords (rows >= K) score exactly +0 and never win a strict '>' against the
                // running best (which starts at |0|), so no per-score validity test is needed.
                argmax_block(best[0], acc0, (row0 >> 5) + rb);
                argmax_block(best[1], acc1, (row0 >> 5) + rb);
            }
        }
        if (!active) continue;
        float bv[2] = {best[0].v, best[1].v};
        int bi[2] = {best[0].i + 4 * h, best[1].i + 4 * h};
        swap32(bv[0], bv[1]);
        swap32(bi[0], bi[1]);
        // a half whose candidate index is out of range (K < 8) never saw a valid codeword
        const bool v0 = bi[0] < K, v1 = bi[1] < K;
        const float a0 = fabsf(bv[0]), a1 = fabsf(bv[1]);
        const bool take1 = v1 && (!v0 || (a1 > a0) || (a1 == a0 && bi[1] < bi[0]));
        float val = take1 ? bv[1] : bv[0];
        int idx = take1 ? bi[1] : bi[0];
        {   // non-finite inputs (hsq_encode_common.hpp): this lane's own subvector from the staged tile (rows beyond
            // the tensor are zeros); the rare scan reads the codebook from global memory (it may be chunked in LDS)
            const float *row = s_v + lane * stride;
            float z = 0.0f;
            for (int e = 0; e < dpad; ++e) z = __fmaf_rn(row[e], 0.0f, z);
            uint64_t todo = __ballot(z != z);
            bool nanhere = false;
            while (todo) {
                const int fl = __builtin_ctzll(todo);
                todo &= todo - 1;
                const float *v = s_v + fl * stride;
                float bvv;
                int bii;
                nonfinite_argmax(K, d, [&](int k, int e) { return cb[(int64_t)k * d + e]; },
                                 [&](int e) { return v[(e & 1) * half + (e >> 1)]; }, bvv, bii);
                if (lane == fl) {
                    val = bvv;
                    idx = bii;
                }
                nanhere = nanhere || (bvv != bvv);
            }
            if (nanhere) {
                sawnan = true;
                if (BATCHED && lane == 0) {   // this tensor's (lb, ub) become NaN
                    atomicMin(bt.seg_minmax + 2 * seg, MAPPED_NAN_LO);
                    atomicMax(bt.seg_minmax + 2 * seg + 1, MAPPED_NAN_HI);
                }
            }
        }
        if (BATCHED) {
            if (seg != cur_seg) {
                flush_minmax(lmin, lmax);
                cur_seg = seg;
            }
            const int64_t local = seg_sv0 + lane;
            u[t * 64 + lane] = val;   // padded index space; lanes beyond the tensor write their (unused) slot
            if (local < seg_m) {
                reinterpret_cast<CodeT *>(bt.wire + rec[3])[local] = (CodeT)idx;
                lmin = fminf(lmin, val);
                lmax = fmaxf(lmax, val);
            }
            continue;
        }
        const int64_t sv = t * 64 + lane;
        if (sv < M) {
            codes[sv] = (CodeT)idx;
            u[sv] = val;
            lmin = fminf(lmin, val);
            lmax = fmaxf(lmax, val);
        }
    }
    if (BATCHED) {
        flush_minmax(lmin, lmax);
        return;
    }
    write_minmax_partials(lmin, lmax, partials, sawnan);
}

// LDS plan of hsq_encode_lds_kernel: false if (d, K) does not fit
bool lds_plan(int d, int K, int *dpad, int *chunk_rows, size_t *bytes, int *waves) {
    static const int limit = [] {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0) != hipSuccess || v < 65536) v = 65536;
        // a dynamic allocation of the full 160 KiB is refused at dispatch (HSA_STATUS_ERROR_INVALID_ALLOCATION,
        // measured); 128 KiB leaves room and costs the chunked shapes next to nothing
        if (v > 128 * 1024) v = 128 * 1024;
        const char *e = getenv("GQ_LDS_LIMIT");   // tests: force the chunked path on small codebooks
        if (e && atoi(e) >= 16384 && atoi(e) < v) v = atoi(e);
        return v;
    }();
    if (d > 128) return false;
    const int dp = (d + 7) & ~7, stride = dp + LDS_ROW_PAD;
    const size_t row = (size_t)stride * sizeof(float);
    // waves per workgroup: 4; callers that pass `waves` accept 2 or 1 when four tiles of 64 rows do not fit next to 32
    // codebook rows (d > ~110: d = 128 ran on the unstaged generic kernel before, 790 us per 25 M elements)
    int w = ENC_WAVES;
    while (w > 1 && waves && (size_t)w * 64 * row + 32 * row > (size_t)limit) w >>= 1;
    const size_t tile = (size_t)w * 64 * row;
    if (tile + 32 * row > (size_t)limit) return false;
    if (waves) *waves = w;
    const int kpad = (K + 31) & ~31;
    int rows = (int)(((size_t)limit - tile) / row) & ~31;
    // the whole codebook when it fits in half the LDS (two workgroups per CU); otherwise the largest chunk
    if ((size_t)kpad * row + tile <= (size_t)limit / 2 || rows >= kpad) rows = kpad < rows ? kpad : rows;
    *dpad = dp;
    *chunk_rows = rows;
    *bytes = tile + (size_t)rows * row;
    return true;
}

// ------------------------------------------------------------------------------------
// VALU cross-check: one subvector per lane, codebook staged in LDS (broadcast reads),
// explicit __fmaf_rn chain.  d <= 64, K*d*4 bytes <= 64 KiB.
// ------------------------------------------------------------------------------------
template <typename CodeT, int D>
__global__ __launch_bounds__(ENC_THREADS) void hsq_encode_valu_kernel(const float *__restrict__ grad,
                                                                     const float *__restrict__ cb, int64_t M, int K,
                                                                     CodeT *__restrict__ codes,
                                                                     float *__restrict__ u,
                                                                     float *__restrict__ partials) {
    extern __shared__ float s_cb[];
    for (int i = threadIdx.x; i < K * D; i += blockDim.x) s_cb[i] = cb[i];
    __syncthreads();
    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
        float v[D];
#pragma unroll
        for (int jj = 0; jj < D; ++jj) v[jj] = grad[m * D + jj];
        float best_v = 0.0f, z = 0.0f;
        int best_i = 0;
#pragma unroll
        for (int jj = 0; jj < D; ++jj) z = __fmaf_rn(v[jj], 0.0f, z);
        const bool odd = z != z;   // a NaN or an infinity in this subvector: torch.argmax's order (hsq_encode_common.hpp)
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int jj = 0; jj < D; ++jj) acc = __fmaf_rn(s_cb[k * D + jj], v[jj], acc);
            if (odd) {
                if (k == 0) {
                    best_v = acc;
                } else {
                    take_if_greater_nan(best_v, best_i, acc, k);
                }
            } else {
                take_if_greater(best_v, best_i, acc, k);
            }
        }
        codes[m] = (CodeT)best_i;
        u[m] = best_v;
        sawnan = sawnan || (__ballot(best_v != best_v) != 0);
        lmin = fminf(lmin, best_v);
        lmax = fmaxf(lmax, best_v);
    }
    write_minmax_partials(lmin, lmax, partials, sawnan);
}

template <typename CodeT>
static int launch_encode(const float *grad, const float *codebook, int64_t M, int d, int K, CodeT *codes, float *u,
                         float *partials, int impl, hipStream_t st, int profile_slot) {
    const int cus = cu_count();
    const int64_t ntiles = (M + 63) / 64;
    static const int bpc_d16 = resident_blocks_per_cu(hsq_encode_d16k256_kernel<CodeT>, ENC_THREADS, 0);
    static const int bpc_gen = resident_blocks_per_cu(hsq_encode_generic_kernel<CodeT>, ENC_THREADS, 0);
    auto grid_for = [&](int bpc) {
        int64_t blocks = (ntiles + ENC_WAVES - 1) / ENC_WAVES;
        int64_t cap = (int64_t)cus * bpc;
        if (cap > GQ_MAIN_PARTIALS) cap = GQ_MAIN_PARTIALS;
        if (blocks > cap) blocks = cap;
        return blocks < 1 ? (int64_t)1 : blocks;
    };
    const int64_t cap = (int64_t)cus * 4 > GQ_MAIN_PARTIALS ? GQ_MAIN_PARTIALS : (int64_t)cus * 4;

    int dpad = 0, chunk_rows = 0;
    size_t lds_bytes = 0;
    int lds_waves = ENC_WAVES;
    const bool lds_ok = lds_plan(d, K, &dpad, &chunk_rows, &lds_bytes, &lds_waves);
    const bool pf_ok = K >= 4 && K <= 256 && (K & 3) == 0 && (d == 8 || d == 12 || d == 16 || d == 24 || d == 32) && (reinterpret_cast<uintptr_t>(grad) & 15) == 0;   // (12 / 24: the reference's repaired dimensions, on the d = 16 / 32 kernels)
    // larger codebooks of the d = 16 family: the prefilter kernel once per page of 256 codewords
    const bool paged_ok = std::is_same<CodeT, int32_t>::value && (d == 8 || d == 16 || d == 32) && K > 256 &&
                          (K & 255) == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0;
    if (impl == 0) impl = (pf_ok || paged_ok) ? 4 : (lds_ok ? 5 : 2);
    if (impl == 4 && K > 256) {
        if constexpr (std::is_same<CodeT, int32_t>::value) {
            if (!paged_ok)
                return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: impl 4 with K > 256 needs d in {8, 16, 32}, K %% 256 == 0, aligned grad");
            return launch_encode_pfd_paged(grad, codebook, M, d, K, codes, u, partials, st);
        } else {
            return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: K > 256 needs int32 codes");
        }
    }
    if (impl == 5) {
        if (!lds_ok) return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode: impl 5 needs d <= 128 (d=%d K=%d)", d, K);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_encode_lds_kernel<CodeT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
            attr_set = true;
        }
        const int bpc = resident_blocks_per_cu(hsq_encode_lds_kernel<CodeT>, lds_waves * 64, lds_bytes);
        int64_t lds_blocks = (ntiles + lds_waves - 1) / lds_waves;
        int64_t lds_cap = (int64_t)cus * bpc;
        if (lds_cap > GQ_MAIN_PARTIALS) lds_cap = GQ_MAIN_PARTIALS;
        if (lds_blocks > lds_cap) lds_blocks = lds_cap;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_lds_kernel<CodeT>), dim3((unsigned)lds_blocks),
                           dim3(lds_waves * 64), lds_bytes, st, grad, codebook, M, d, K, codes, u, partials, dpad,
                           chunk_rows, LdsBatch{});
        GQ_CHECK_LAUNCH("gq_hsq_encode (lds)");
        return GQ_OK;
    }
    if (impl == 4) {
        if (!(d == 8 || d == 12 || d == 16 || d == 24 || d == 32) || K < 4 || K > 256 || (K & 3)) return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: impl 4 needs K <= 256 (a multiple of 4) and d in {8, 12, 16, 24, 32}");
        if ((reinterpret_cast<uintptr_t>(grad) & 15) != 0)
            return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: grad must be 16-byte aligned");
        return launch_encode_pf<CodeT>(grad, codebook, M, d, K, codes, u, partials, st, profile_slot);
    }
    if (impl == 1) {
        if (d != 16 || K != 256) return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: impl 1 needs d=16, K=256");
        if ((reinterpret_cast<uintptr_t>(grad) & 15) != 0)
            return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: grad must be 16-byte aligned");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_d16k256_kernel<CodeT>), dim3((unsigned)grid_for(bpc_d16)),
                           dim3(ENC_THREADS), 0, st, grad, codebook, M, codes, u, partials);
    } else if (impl == 2) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_generic_kernel<CodeT>), dim3((unsigned)grid_for(bpc_gen)),
                           dim3(ENC_THREADS), 0, st, grad, codebook, M, d, K, codes, u, partials);
    } else if (impl == 3) {
        const size_t lds = (size_t)K * d * sizeof(float);
        if (lds > 64 * 1024) return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode: valu impl needs K*d*4 <= 64 KiB");
        int64_t vb = (M + ENC_THREADS - 1) / ENC_THREADS;
        if (vb > cap) vb = cap;
        if (vb < 1) vb = 1;
#define GQ_VALU_CASE(DD)                                                                                        \
    case DD:                                                                                                    \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_valu_kernel<CodeT, DD>), dim3((unsigned)vb),              \
                           dim3(ENC_THREADS), lds, st, grad, codebook, M, K, codes, u, partials);               \
        break;
        switch (d) {
            GQ_VALU_CASE(8)
            GQ_VALU_CASE(12)
            GQ_VALU_CASE(16)
            GQ_VALU_CASE(24)
            GQ_VALU_CASE(32)
            default:
                return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode: valu impl is built for d in {8,12,16,24,32}");
        }
#undef GQ_VALU_CASE
    } else {
        return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: unknown impl %d", impl);
    }
    GQ_CHECK_LAUNCH("gq_hsq_encode");
    return GQ_OK;
}

// Multi-tensor launch of the LDS kernel (any d <= 128, any K): gq_hsq_encode_batched_any.
template <typename CodeT, bool EF>
static int launch_encode_lds_batched(const int64_t *seg_table, const int32_t *tile_seg, int64_t ntiles,
                                     const float *codebook, int d, int K, float ef_scale, uint8_t *wire, float *u_flat,
                                     uint32_t *seg_minmax, hipStream_t st) {
    int dpad = 0, chunk_rows = 0;
    size_t lds_bytes = 0;
    if (!lds_plan(d, K, &dpad, &chunk_rows, &lds_bytes, nullptr))
        return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode_batched (exact): needs d <= 128 (d=%d K=%d)", d, K);
    auto kernel = hsq_encode_lds_kernel<CodeT, true, EF>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  128 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
    const int bpc = resident_blocks_per_cu(kernel, ENC_THREADS, lds_bytes);
    int64_t blocks = (ntiles + ENC_WAVES - 1) / ENC_WAVES;
    const int64_t cap = (int64_t)cu_count() * bpc;
    if (blocks > cap) blocks = cap;
    LdsBatch bt = {seg_table, tile_seg, wire, seg_minmax, ef_scale};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_lds_kernel<CodeT, true, EF>), dim3((unsigned)blocks),
                       dim3(ENC_THREADS), lds_bytes, st, (const float *)nullptr, codebook, ntiles * 64, d, K,
                       (CodeT *)nullptr, u_flat, (float *)nullptr, dpad, chunk_rows, bt);
    GQ_CHECK_LAUNCH("gq_hsq_encode_batched_any");
    return GQ_OK;
}

}  // namespace gq

GQ_INTERNAL int gqi_hsq_batched_any_supported(int d, int K) {
    int dpad = 0, chunk_rows = 0;
    size_t bytes = 0;
    return (d >= 1 && K >= 1 && K <= 65536 && gq::lds_plan(d, K, &dpad, &chunk_rows, &bytes, nullptr)) ? 1 : 0;
}

GQ_INTERNAL int gqi_hsq_encode_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                     const float *codebook, int d, int K, int code_bytes, int ef, float ef_scale,
                                     uint8_t *wire, float *u_flat, uint32_t *seg_minmax, void *stream) {
    if (nseg < 1 || ntiles < 1 || ntiles * 64 > 0x7FFFFFFFLL || d < 1 || K < 1 || K > 65536)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched (exact): bad sizes nseg=%d ntiles=%lld d=%d K=%d", nseg,
                        (long long)ntiles, d, K);
    if (!seg_table || !tile_seg || !codebook || !wire || !u_flat || !seg_minmax)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched (exact): null pointer");
    if (code_bytes != 1 && code_bytes != 4)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched (exact): code_bytes must be 1 or 4");
    if (code_bytes == 1 && K > 256)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched (exact): uint8 codes need K <= 256");
    hipStream_t st = gq::as_stream(stream);
    if (code_bytes == 1)
        return ef ? gq::launch_encode_lds_batched<uint8_t, true>(seg_table, tile_seg, ntiles, codebook, d, K, ef_scale,
                                                                 wire, u_flat, seg_minmax, st)
                  : gq::launch_encode_lds_batched<uint8_t, false>(seg_table, tile_seg, ntiles, codebook, d, K, 0.0f, wire,
                                                                  u_flat, seg_minmax, st);
    return ef ? gq::launch_encode_lds_batched<int32_t, true>(seg_table, tile_seg, ntiles, codebook, d, K, ef_scale, wire,
                                                             u_flat, seg_minmax, st)
              : gq::launch_encode_lds_batched<int32_t, false>(seg_table, tile_seg, ntiles, codebook, d, K, 0.0f, wire,
                                                              u_flat, seg_minmax, st);
}

GQ_API size_t gq_hsq_workspace_bytes(int64_t M) {
    if (M < 0) M = 0;
    return (size_t)(2 * GQ_MAX_PARTIALS) * sizeof(float) + 16 + (size_t)M * sizeof(int32_t);
}

GQ_API int gq_hsq_encode_ex(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes,
                            int code_bytes, float *u, float *minmax_partials, int impl, int profile_slot, void *stream) {
    if (M < 1 || d < 1 || d > 512 || K < 1 || K > 65536)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: bad sizes M=%lld d=%d K=%d", (long long)M, d, K);
    if (!grad || !codebook || !codes || !u || !minmax_partials)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: null pointer");
    if (profile_slot >= GQ_PROFILE_SLOTS) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_ex: profile_slot %d", profile_slot);
    if (code_bytes == 1) {
        if (K > 256) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: uint8 codes need K <= 256");
        return gq::launch_encode<uint8_t>(grad, codebook, M, d, K, static_cast<uint8_t *>(codes), u, minmax_partials,
                                          impl, gq::as_stream(stream), profile_slot);
    }
    if (code_bytes == 4)
        return gq::launch_encode<int32_t>(grad, codebook, M, d, K, static_cast<int32_t *>(codes), u, minmax_partials,
                                          impl, gq::as_stream(stream), profile_slot);
    return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: code_bytes must be 1 or 4");
}

GQ_API int gq_hsq_encode(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes,
                         int code_bytes, float *u, float *minmax_partials, void *stream) {
    return gq_hsq_encode_ex(grad, codebook, M, d, K, codes, code_bytes, u, minmax_partials, GQ_ENCODE_AUTO, -1, stream);
}
