// HSQ encode, sub-dimension D = 8 or 32: round 3's bf16x3 matrix-core prefilter + exact f32 rescoring (three bf16 MFMAs per
// chain and k-step: hi x hi, hi x lo, lo x hi; keys and exactness argument as in hsq_encode_pf.hip's header, its error
// bound the bf16x3 one stated below).  Since round 5 the product path for K = 256 is hsq_encode_pf.hip's one-f16-MFMA kernel,
// a template over D = 8 / 16 / 32; what this file serves is the PAGED form, K = 512 ... 65536 (codebook pages of 256 rows,
// int32 codes), single- and multi-tensor.  (Its single-page K = 256 form was gq_hsq_encode_ex's impl 6 until round 6, a
// cross-check of the f16 kernel that the exact kernels -- impl 1 / 2 / 5 -- cover: d = 32 / 8 ran 42.1 / 68.7 us against 30.5 / 54.6.)
// What differs from the d = 16 kernel of round 3:
//   * a chain is 3 * KS v_mfma_f32_32x32x16_bf16 with KS = ceil(D / 16) k-steps; D = 8 feeds zeros for the
//     upper half of the one k-step;
//   * the A fragments (codebook hi / lo bf16 parts) do not fit the register file next to everything else
//     for D = 32, so they live in LDS as ready-made fragments -- s_a[(rb * KS + s) * 2 + part][lane], one
//     conflict-free ds_read_b128 each -- and are fetched one row block ahead; the chains run
//     (rb, block 0), (rb, block 1) so that both blocks share them;
//   * the error bound E = 2^-15 max||c||_1 max|v_j| still holds: dropped terms 3 * 2^-18 and, for the 96
//     accumulated products of D = 32, accumulate roundings <= ~2^-17.4, both relative to sum |c_j||v_j|.
// A tile is 64 subvectors (64 * D floats); scores per gradient element are 256 / D, so D = 32 costs half of
// D = 16 per element and D = 8 twice as much.
#include "hsq_pf_common.hpp"
// waves per workgroup (one workgroup per CU): D = 32 two per SIMD (<= 251 VGPRs per lane since the library is built
// without packed f32 operations; with them the kernel wanted ~330 and ran one wave per SIMD: 59 -> 52 us),
// D = 8 three per SIMD (<= 168 VGPRs; 78 -> 76 us)
#ifndef GQ_W32
#define GQ_W32 8
#endif
#ifndef GQ_W8
#define GQ_W8 12
#endif
// tiles at the end of a workgroup's run that the second wave of a SIMD leaves to the first (hsq_encode_pf.hip's tail rule):
// D = 32 has only ~48 tiles of 5.5 us per workgroup -- 2 measured best (0: same, 4: +0.4 %, 6: +2 %); D = 8 is flat from 0 to 6
#ifndef GQ_PFD_TAIL
#define GQ_PFD_TAIL (D == 32 ? 2 : 6)
#endif

namespace gq {

struct PfdArgs {
    const float *grad;
    int64_t M;
    void *codes;
    float *u;
    const float *cb;
    float *ws;
    // batched form (gq_hsq_encode_batched_d): as PfArgs in hsq_encode_pf.hip; M = ntiles * 64
    const int64_t *seg_table;
    const int32_t *tile_seg;
    uint8_t *wire;
    unsigned *seg_minmax;
    int64_t ntiles;
    int nseg;
    float ef_scale;   // EF form: grad <- grad + ef_scale * error (error pointer = seg_table[seg][7], 0 = none)
    int code_base;    // PAGED (hsq_encode_pf.hip): index of this page's first codeword, and whether to keep
    int merge;        // the (code, u) already in the output unless this page beats it
    int last_page;    // PAGED: this launch produces the final projections (fold their min / max)
    int tiles_q, tiles_r;   // tiles per workgroup, split on the host (hsq_encode_pf.hip): the first tiles_r take one more
    int npages;             // PAGED: pages of 256 codewords THIS launch scores (all of them resident in LDS)
};

static void pfd_split(PfdArgs &a, int64_t ntiles, int64_t blocks) {
    a.tiles_q = (int)(ntiles / blocks);
    a.tiles_r = (int)(ntiles % blocks);
}

constexpr int PFD_LDS_SEGS = 384;   // batched form: segment records kept in LDS (24 KiB); longer lists are refused

// D = 32 holds two tiles of 64 x 32 floats, the fragments of both and two accumulators: 205-251 VGPRs per lane
// (built with -amdgpu-mfma-vgpr-form so that the accumulators stay in VGPRs: build.py).  While packed f32
// operations were in the build the same source wanted ~330, spilled 112 at two waves per SIMD (101 us per 25 M
// elements) and ran at one wave per SIMD (61 us).
// BATCHED: the multi-tensor form -- segment table, one contiguous run of tiles per wave, per-tensor (min,max)
// by look-before-you-leap atomics, table-derived addresses as global address-space pointers: everything as in
// hsq_encode_pf.hip, where the reasons are written down.
// EF (batched only): error feedback folded into the load, as in hsq_encode_pf.hip -- the tile is read as
// v = grad + ef_scale * error (product rounded, then the add), v is written back over grad, and the level
// kernel (gq_hsq_levels_batched_ef_d) later writes error = v - decoded.
template <typename CodeT, int D, bool BATCHED = false, bool EF = false, bool PAGED = false>
__global__ __launch_bounds__((D == 32 ? GQ_W32 : (D == 16 ? 8 : GQ_W8)) * 64, 1) void hsq_encode_pfd_kernel(const PfdArgs a) {
    // one workgroup per CU: 12 waves (three per SIMD) for D = 8, 8 waves (two per SIMD) for D = 32; the waves share
    // the workgroup's contiguous run of tiles through an LDS counter (hsq_encode_pf.hip)
    constexpr int WAVES = D == 32 ? GQ_W32 : (D == 16 ? 8 : GQ_W8);   // D = 16: two waves per SIMD, as hsq_encode_pf.hip
    constexpr int THREADS = WAVES * 64;
    static_assert(D == 8 || D == 16 || D == 32, "built for D = 8, 16 (PAGED only; K = 256: hsq_encode_pf.hip) and 32");
    // PAGED: K = 256 * pages.  A launch keeps `a.npages` pages of the codebook in LDS (bf16 fragments and the f32
    // image for the exact rescoring: as many as fit), a tile's subvectors are loaded and split ONCE and scored
    // against one page after the other (per-page top-2 trackers, folded into running ones that remember the
    // page), and the ONE exact rescoring / fix-up at the end of the tile covers all of them.  Before, every page
    // was a launch of its own that re-read the gradient (K = 1024: 4x, K = 4096: 16x the traffic and 4x / 16x the
    // per-tile overhead).  Codebooks beyond the LDS take several such launches, merged in place as before.
    const int npages = PAGED ? a.npages : 1;
    constexpr int KS = D > 16 ? D / 16 : 1;    // MFMA k-steps per chain
    constexpr int QS = 4 * D + 4;              // LDS floats per group of 4 codewords: an odd number of 16-byte units
    const float *__restrict__ cb = a.cb;
    float *__restrict__ ws = a.ws;
    float *__restrict__ u = a.u;
    const float *__restrict__ grad = a.grad;
    const int64_t M = a.M;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const s_cb = lds;                                                         // [npages * 64 groups][QS]
    bf16x8 *const s_a = reinterpret_cast<bf16x8 *>(lds + npages * 64 * QS);          // [npages * 8 * KS * 2][64 lanes]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;

    __shared__ int s_next;
    __shared__ int64_t s_seg[BATCHED ? PFD_LDS_SEGS * 8 : 1];
    __shared__ float s_c1[WAVES];
    // batched: the workgroup's (min, max) per tensor for the PFD_MM_SEGS tensors from its first tile's on, folded by LDS
    // atomics; one pair of global atomics per (workgroup, tensor) at the end (hsq_encode_pf.hip, s_mm)
    constexpr int PFD_MM_SEGS = 64;
    __shared__ unsigned s_mm[BATCHED ? 2 * PFD_MM_SEGS : 2];

    const int b = (int)blockIdx.x;
    const int64_t lo_tile = (int64_t)b * a.tiles_q + (b < a.tiles_r ? b : a.tiles_r);
    const int64_t tile_end = lo_tile + a.tiles_q + (b < a.tiles_r ? 1 : 0);
    // the slower wave of a SIMD (waves 4-7 of an 8-wave workgroup) leaves the last tiles of the run to the faster one
    const int tail_from = (int)(tile_end - lo_tile) - ((WAVES >= 8 && wave >= 4) ? GQ_PFD_TAIL : 0);
    auto draw = [&]() {   // the next tile of this workgroup's run (may lie beyond tile_end)
        int k = 0x3FFFFFFF;
        if (lane == 0) {
            if (__hip_atomic_load(&s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tail_from)
                k = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return lo_tile + __builtin_amdgcn_readfirstlane(k);
    };
    int64_t t = lo_tile + wave;
    int *const worklist = ws_worklist(ws);
    // (the tensor of the workgroup's first tile = entry 0 of s_mm; the table is cleared here, every fold lies behind the
    // kernel's first barrier)
    const int seg_first = (BATCHED && lo_tile < tile_end) ? __builtin_amdgcn_readfirstlane(a.tile_seg[lo_tile]) : 0;
    if (BATCHED && threadIdx.x < PFD_MM_SEGS) {
        s_mm[2 * threadIdx.x] = 0xFFFFFFFFu;
        s_mm[2 * threadIdx.x + 1] = 0u;
    }

    typedef const float __attribute__((address_space(1))) *gcf_ptr;
    typedef const f32x4 __attribute__((address_space(1))) *gcv_ptr;
    typedef CodeT __attribute__((address_space(1))) *gcode_ptr;
    typedef f32x4 __attribute__((address_space(1))) *gv_ptr;
    struct Tile {
        gcf_ptr base;      // the tensor's first float
        int64_t m, sv0;    // its subvector count; this tile's first subvector inside it
        int seg;
        gcode_ptr codes;
        gcf_ptr err;       // EF: the tensor's error buffer or null
    };
    auto uniform64 = [](int64_t v) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto tile_info = [&](int64_t tile, int seg) {
        Tile ti;
        if (BATCHED) {
            const int64_t *rec = s_seg + 8 * seg;
            ti.seg = seg;
            ti.base = (gcf_ptr)(uintptr_t)uniform64(rec[0]);
            ti.m = uniform64(rec[1]);
            ti.sv0 = (tile - uniform64(rec[2])) * 64;
            ti.codes = (gcode_ptr)((uintptr_t)a.wire + (uintptr_t)uniform64(rec[3]));
            ti.err = EF ? (gcf_ptr)(uintptr_t)uniform64(rec[7]) : (gcf_ptr)0;
        } else {
            ti.seg = 0;
            ti.base = (gcf_ptr)grad;
            ti.m = M;
            ti.sv0 = tile * 64;
            ti.codes = (gcode_ptr)static_cast<CodeT *>(a.codes);
            ti.err = (gcf_ptr)0;
        }
        return ti;
    };
    // tile data: lane (j, h) holds, for block b and k-step s, floats [16 s + 8 h, + 8) of subvector 32 b + j
    auto load_tile = [&](const Tile &ti, f32x4 (&dst)[2][KS][2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int64_t sv = ti.sv0 + b * 32 + j;
            sv = sv < ti.m ? sv : ti.m - 1;  // tail: re-read the last subvector, result is masked
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (16 * s + 8 * h < D) {
                    const gcv_ptr p = (gcv_ptr)(ti.base + sv * D + 16 * s + 8 * h);
                    dst[b][s][0] = p[0];
                    dst[b][s][1] = p[1];
                } else {
                    dst[b][s][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    dst[b][s][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
    };
    auto load_err = [&](const Tile &ti, f32x4 (&dst)[2][KS][2]) {
        if (EF && ti.err) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int64_t sv = ti.sv0 + b * 32 + j;
                sv = sv < ti.m ? sv : ti.m - 1;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (16 * s + 8 * h < D) {
                        const gcv_ptr p = (gcv_ptr)(ti.err + sv * D + 16 * s + 8 * h);
                        dst[b][s][0] = p[0];
                        dst[b][s][1] = p[1];
                    }
                }
            }
        }
    };
    // v = grad + scale * error, written back over grad (valid subvectors only)
    auto fold_err = [&](const Tile &ti, f32x4 (&g)[2][KS][2], const f32x4 (&e)[2][KS][2]) {
        if (EF && ti.err) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int64_t sv = ti.sv0 + b * 32 + j;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (16 * s + 8 * h < D) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float prod = a.ef_scale * e[b][s][q][c];
                                g[b][s][q][c] = g[b][s][q][c] + prod;
                            }
                        if (sv < ti.m) {
                            const gv_ptr p = (gv_ptr)(uintptr_t)(ti.base + sv * D + 16 * s + 8 * h);
                            p[0] = g[b][s][0];
                            p[1] = g[b][s][1];
                        }
                    }
                }
            }
        }
    };
    auto load_a = [&](int rb, bf16x8 (&hi)[KS], bf16x8 (&lo)[KS]) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            hi[s] = s_a[((rb * KS + s) * 2 + 0) * 64 + lane];
            lo[s] = s_a[((rb * KS + s) * 2 + 1) * 64 + lane];
        }
    };

    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)
    int cur_seg = -1;  // batched: segment the running (lmin, lmax) belongs to
    auto flush_minmax = [&]() {  // batched: fold this wave's running (min,max) into its segment
        const float lo = wave_min(lmin), hi = wave_max(lmax);
        if (lane == 0 && cur_seg >= 0 && lo <= hi) {
            const unsigned mlo = order_map(lo), mhi = order_map(hi);
            const unsigned idx = (unsigned)(cur_seg - seg_first);
            if (idx < (unsigned)PFD_MM_SEGS) {
                atomicMin(&s_mm[2 * idx], mlo);
                atomicMax(&s_mm[2 * idx + 1], mhi);
            } else {   // beyond the table: the tensor's global words, looking before the atomic (hsq_encode_pf.hip)
                unsigned *mm = a.seg_minmax + 2 * cur_seg;
                if (mlo < __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(mm, mlo);
                if (mhi > __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mm + 1, mhi);
            }
        }
        lmin = INFINITY;
        lmax = -INFINITY;
    };
    auto seg_of = [&](int64_t tile) {   // batched: tile -> tensor, one global word (0 beyond the end)
        return (BATCHED && tile < tile_end) ? a.tile_seg[tile] : 0;
    };
    f32x4 cur[2][KS][2], nxt[2][KS][2];
    f32x4 nxte[2][KS][2];   // EF: the error tile that goes with nxt (dead otherwise)
    bf16x8 vh[2][KS], vl[2][KS];
    Tile ti = {};
    // ---- prologue.  K = 256 (one page): everything the workgroup needs from memory is requested first, in the order
    // it is used -- the codebook words for the LDS image, this wave's row block of the codebook (every wave splits ONE
    // of the 8 row blocks into bf16 hi / lo fragments and shares it through LDS; ||c||_1 comes from the same
    // registers), the wave's first tile (single tensor), whose HBM latency then hides behind the staging -- and there
    // is one barrier: hsq_encode_pf.hip's prologue (D = 32: 8.0 us before, encode 47.5 -> 44.2 us).  PAGED keeps the plain loops below.
    bool first_loaded = false;
    if constexpr (!PAGED) {
        constexpr int NCB = (256 * D + THREADS - 1) / THREADS;
        float cbv[NCB];
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            const int i = threadIdx.x + n * THREADS;
            cbv[n] = (256 * D % THREADS == 0 || i < 256 * D) ? cb[i] : 0.0f;
        }
        f32x4 aq[KS][2];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            aq[s][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            aq[s][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (wave < 8 && 16 * s + 8 * h < D) {
                aq[s][0] = *reinterpret_cast<const f32x4 *>(cb + (wave * 32 + j) * D + 16 * s + 8 * h);
                aq[s][1] = *reinterpret_cast<const f32x4 *>(cb + (wave * 32 + j) * D + 16 * s + 8 * h + 4);
            }
        }
        if (!BATCHED && t < tile_end) {
            ti = tile_info(t, 0);
            load_tile(ti, cur);
            first_loaded = true;
        }
        if (threadIdx.x == 0) s_next = WAVES;
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            const int i = threadIdx.x + n * THREADS;
            if (256 * D % THREADS == 0 || i < 256 * D) {
                const int k = i / D, e = i % D;
                s_cb[(k >> 2) * QS + 4 * e + (k & 3)] = cbv[n];
            }
        }
        if (BATCHED) {
            for (int i = threadIdx.x; i < a.nseg * 8; i += THREADS) s_seg[i] = a.seg_table[i];
        }
        float l1 = 0.0f;
        if (wave < 8) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 hi, lo;
                split8(aq[s][0], aq[s][1], hi, lo);
                s_a[((wave * KS + s) * 2 + 0) * 64 + lane] = hi;
                s_a[((wave * KS + s) * 2 + 1) * 64 + lane] = lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) l1 += fabsf(aq[s][0][e]) + fabsf(aq[s][1][e]);
            }
            l1 += __shfl_xor(l1, 32, 64);   // row wave*32 + j: this lane's elements + its partner's
            l1 = wave_max(l1);
        }
        if (lane == 0) s_c1[wave] = l1;
        __syncthreads();
    } else {
        // exact f32 codebook, groups of 4 codewords interleaved: s_cb[(k>>2)*QS + 4*e + (k&3)] = c[k][e]
        if (threadIdx.x == 0) s_next = WAVES;
        for (int i = threadIdx.x; i < npages * 256 * D; i += THREADS) {
            const int k = i / D, e = i % D;
            s_cb[(k >> 2) * QS + 4 * e + (k & 3)] = cb[i];
        }
        // A fragments of v_mfma_f32_32x32x16_bf16: lane (row j, half h) of k-step s holds
        // c[rb*32 + j][16 s + 8 h .. + 7] (zeros beyond D), split into bf16 hi and lo
        for (int i = threadIdx.x; i < npages * 8 * KS * 64; i += THREADS) {
            const int l = i & 63, s = (i >> 6) % KS, rb = i / (64 * KS);
            const int row = rb * 32 + (l & 31), e0 = 16 * s + 8 * (l >> 5);
            f32x4 q0 = {0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0;
            if (e0 < D) {
                q0 = *reinterpret_cast<const f32x4 *>(cb + row * D + e0);
                q1 = *reinterpret_cast<const f32x4 *>(cb + row * D + e0 + 4);
            }
            bf16x8 hi, lo;
            split8(q0, q1, hi, lo);
            s_a[((rb * KS + s) * 2 + 0) * 64 + l] = hi;
            s_a[((rb * KS + s) * 2 + 1) * 64 + l] = lo;
        }
        if (BATCHED) {
            for (int i = threadIdx.x; i < a.nseg * 8; i += THREADS) s_seg[i] = a.seg_table[i];
        }
        __syncthreads();
        // the error bound scales with max_k ||c_k||_1: measured, not assumed
        {
            float l1 = 0.0f;
            for (int k = threadIdx.x & 255; k < npages * 256; k += 256) {
                float rowsum = 0.0f;
#pragma unroll
                for (int e = 0; e < D; ++e) rowsum += fabsf(s_cb[(k >> 2) * QS + 4 * e + (k & 3)]);
                l1 = fmaxf(l1, rowsum);
            }
            l1 = wave_max(l1);
            if (lane == 0) s_c1[wave] = l1;
        }
        __syncthreads();
    }
    float c1 = s_c1[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) c1 = fmaxf(c1, s_c1[w]);
    const float err_scale = c1 * ERR_SCALE;  // E = max|v_j| * err_scale

    int64_t tn = draw();
    int seg_n = seg_of(tn);
    int seg_next = 0;
    if (t < tile_end) {
        seg_next = BATCHED ? __builtin_amdgcn_readfirstlane(seg_n) : 0;
        if (!first_loaded) {
            ti = tile_info(t, BATCHED ? __builtin_amdgcn_readfirstlane(a.tile_seg[t]) : 0);
            load_tile(ti, cur);
        }
        load_err(ti, nxte);
        fold_err(ti, cur, nxte);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < KS; ++s) split8(cur[b][s][0], cur[b][s][1], vh[b][s], vl[b][s]);
    }
    while (t < tile_end) {
        const int64_t tnn = BATCHED ? draw() : 0;   // batched: one more tile ahead, for its tile -> tensor word
        Tile tin = ti;
        if (tn < tile_end) {
            tin = tile_info(tn, seg_next);
            load_tile(tin, nxt);  // prefetch the next tile
            load_err(tin, nxte);
        }
        seg_n = seg_of(tnn);
        if (BATCHED && ti.seg != cur_seg) {
            flush_minmax();
            cur_seg = ti.seg;
        }
        // PAGED: what the earlier pages left for this lane's subvector, requested a whole tile before its use
        float prev_u = 0.0f;
        int prev_idx = 0;
        if (PAGED && a.merge && ti.sv0 + lane < ti.m) {
            prev_u = u[BATCHED ? t * 64 + lane : ti.sv0 + lane];
            prev_idx = (int)ti.codes[ti.sv0 + lane];
        }

        // ---- prefilter: 16 chains in the order (rb, block 0), (rb, block 1); top-2 GROUP keys per
        // (block, row-block half).  The MFMAs of chain c+1 sit between the key operations of chain c.
        unsigned best[4] = {0, 0, 0, 0}, second[4] = {0, 0, 0, 0};
        int bpage[4] = {0, 0, 0, 0};     // PAGED: the page each tracker's best group belongs to
        unsigned vmask = KEY_MASK;
        asm volatile("" : "+v"(vmask));  // keep the mask in a VGPR: v_and_or with an SGPR operand issues slower
        for (int page = 0; page < npages; ++page) {
            unsigned pbest[4] = {0, 0, 0, 0}, psecond[4] = {0, 0, 0, 0};   // this page's top-2 group keys
            auto group_key = [&](const f32x16 &x, int rb, int q) {   // group q = registers 4q..4q+3 = four consecutive rows
                return and_or(__float_as_uint(absmax4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3])), vmask,
                              (unsigned)((rb & 3) * 4 + q));
            };
            auto track = [&](int trk, unsigned k0, unsigned k1) {
                psecond[trk] = max(psecond[trk], med3u(pbest[trk], k0, k1));
                pbest[trk] = max3u(pbest[trk], k0, k1);
            };
            const int rb0 = page * 8;        // the page's first row block in s_a
            bf16x8 ahi[2][KS], alo[2][KS];   // A fragments, double-buffered by row-block parity
            load_a(rb0, ahi[0], alo[0]);
            f32x16 acc = {0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[0][s], vh[0][s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[0][s], vl[0][s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[0][s], vh[0][s], acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int rb = c >> 1, blk = c & 1, trk = blk * 2 + (rb >> 2);
                if (blk == 0 && rb + 1 < 8) load_a(rb0 + rb + 1, ahi[(rb + 1) & 1], alo[(rb + 1) & 1]);   // a row block ahead
                if (c + 1 < 16) {
                    const int nr = (c + 1) >> 1, nb = (c + 1) & 1, ab = nr & 1;
                    f32x16 nacc = {0};
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < KS; ++s) nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ab][s], vh[nb][s], nacc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    track(trk, group_key(acc, rb, 0), group_key(acc, rb, 1));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < KS; ++s) nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ab][s], vl[nb][s], nacc, 0, 0, 0);
#pragma unroll
                    for (int s = 0; s < KS; ++s) nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ab][s], vh[nb][s], nacc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    track(trk, group_key(acc, rb, 2), group_key(acc, rb, 3));
                    __builtin_amdgcn_sched_barrier(0);
                    acc = nacc;
                } else {
                    track(trk, group_key(acc, rb, 0), group_key(acc, rb, 1));
                    track(trk, group_key(acc, rb, 2), group_key(acc, rb, 3));
                }
            }
            // fold the page's trackers into the running ones (a later page wins only with a strictly larger key)
#pragma unroll
            for (int trk = 0; trk < 4; ++trk) {
                if (!PAGED) {
                    best[trk] = pbest[trk];
                    second[trk] = psecond[trk];
                } else {
                    const bool take = (pbest[trk] & KEY_MASK) > (best[trk] & KEY_MASK);
                    second[trk] = max3u(second[trk], psecond[trk], take ? best[trk] : pbest[trk]);
                    bpage[trk] = take ? page : bpage[trk];
                    best[trk] = take ? pbest[trk] : best[trk];
                }
            }
        }

        // ---- per block: merge the two trackers; best group's first codeword and the bound on the rest ----
        int k1[2];
        unsigned s2[2], bk[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const unsigned bA = best[2 * blk], bB = best[2 * blk + 1];
            const bool useB = (bB & KEY_MASK) > (bA & KEY_MASK);
            const unsigned bw = useB ? bB : bA, bl = useB ? bA : bB;
            const int gid = (int)(bw & 31u);
            const int pg = PAGED ? (useB ? bpage[2 * blk + 1] : bpage[2 * blk]) : 0;
            k1[blk] = pg * 256 + ((gid >> 2) + (useB ? 4 : 0)) * 32 + 8 * (gid & 3) + 4 * h;   // rows k1 .. k1+3 (registers 4q..4q+3)
            s2[blk] = max3u(second[2 * blk], second[2 * blk + 1], bl) | 31u;         // upper end of its bucket
            bk[blk] = bw;
        }

        // ---- this lane's own full subvector (tile subvector `lane`): swaps of the B loads ----
        float vf[D];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = cur[0][s][e >> 2][e & 3];   // block 0: floats 16 s + 8 h + e of subvector j
                float y = cur[1][s][e >> 2][e & 3];   // block 1
                swap32(x, y);                          // x = floats 16 s + e, y = floats 16 s + 8 + e of subvector `lane`
                vf[(16 * s + e) % D] = x;
                if (16 * s + 8 + e < D) vf[(16 * s + 8 + e) % D] = y;
            }
        }
        // cross-half exchange of the candidates: [0] = lower-half rows, [1] = upper-half rows
        swap32(k1[0], k1[1]);
        {
            int a0 = (int)s2[0], a1 = (int)s2[1];
            swap32(a0, a1);
            s2[0] = (unsigned)a0;
            s2[1] = (unsigned)a1;
            int b0 = (int)bk[0], b1 = (int)bk[1];
            swap32(b0, b1);
            bk[0] = (unsigned)b0;
            bk[1] = (unsigned)b1;
        }

        // ---- exact rescoring of the better of the two halves' best groups (4 codewords; the
        // reference's fmaf chain); the other half's best group joins the bound on the rest
        const bool pick1 = (bk[1] & KEY_MASK) > (bk[0] & KEY_MASK);
        const int kc = pick1 ? k1[1] : k1[0];
        const unsigned rest = max3u(s2[0], s2[1], (pick1 ? bk[0] : bk[1]) | 31u);
        const f32x4 p4 = exact_score_quad<D>(s_cb + (kc >> 2) * QS, vf);   // kc is a multiple of 4: one group
        float val = p4[0];
        int idx = kc;
        take_if_greater(val, idx, p4[1], kc + 1);
        take_if_greater(val, idx, p4[2], kc + 2);
        take_if_greater(val, idx, p4[3], kc + 3);

        float vmax = 0.0f;
#pragma unroll
        for (int e = 0; e < D; e += 2) vmax = fmaxf(fmaxf(fabsf(vf[e]), fabsf(vf[e + 1])), vmax);
        const float E = vmax * err_scale;
        const float others = __uint_as_float(rest);  // >= every s~ outside the rescored group
        bool safe = (others + E < fabsf(val)) && (vmax >= 8.27e-25f) && (vmax <= 1.0e30f);
        // a NaN score: the float comparisons above are compiled for NaN-free operands (-fno-honor-nans: the
        // complement `others + E >= |val|` is what is evaluated, false for NaN), so the bits decide
        if (nan_bits(val)) safe = false;
        if (vmax == 0.0f && !nan_bits(val)) {  // all-zero subvector: every score is +0 -> first index, u = +0
            safe = true;
            val = 0.0f;
            idx = 0;
        }
        // NaN anywhere makes the comparisons false -> not safe -> exact path

        const int64_t sv = ti.sv0 + lane;                   // index inside this tile's tensor
        const bool valid = sv < ti.m;
        const int64_t gsv = BATCHED ? t * 64 + lane : sv;   // index into u / the fix-up log

        // consume the prefetched tile BEFORE this tile's stores are issued (hsq_encode_pf.hip)
        if (BATCHED) seg_next = __builtin_amdgcn_readfirstlane(seg_n);
        if (EF && tn < tile_end) fold_err(tin, nxt, nxte);
        if (tn < tile_end) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    split8(nxt[b][s][0], nxt[b][s][1], vh[b][s], vl[b][s]);
                    cur[b][s][0] = nxt[b][s][0];
                    cur[b][s][1] = nxt[b][s][1];
                    if (KS > 1) __builtin_amdgcn_sched_barrier(0);   // one fragment's temporaries at a time (register pressure)
                }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- exact fix-up, in place and wave-wide, for the few subvectors the bound could not settle
        uint64_t todo = __ballot(valid && !safe);
        while (todo) {
            const int fl = __builtin_ctzll(todo);
            todo &= todo - 1;
            float bv = 0.0f;
            int bi = lane;
            // rare path: keep it light on registers (one codeword at a time, 8 LDS reads in flight)
#pragma unroll 1
            for (int q = 0; q < 4 * npages; ++q) {
                const int k = q * 64 + lane;
                const float *row = s_cb + (k >> 2) * QS + (k & 3);
                float sc = 0.0f;
#pragma unroll
                for (int e = 0; e < D; ++e) {
                    if (e % 8 == 0) __builtin_amdgcn_sched_barrier(0);
                    const float we = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vf[e]), fl));
                    sc = __fmaf_rn(row[4 * e], we, sc);
                }
                if (q == 0) {
                    bv = sc;
                } else {
                    take_if_greater_nan(bv, bi, sc, k);   // torch.argmax's order: NaN is the largest, the first one wins
                }
            }
            wave_first_max_nan(bv, bi);
            if (lane == fl) {
                val = bv;
                idx = bi;
            }
            if (nan_bits(bv)) {   // (lb, ub) of this tensor become NaN (torch.min / torch.max propagate it)
                sawnan = true;
                if (BATCHED && lane == 0) {
                    atomicMin(a.seg_minmax + 2 * ti.seg, MAPPED_NAN_LO);
                    atomicMax(a.seg_minmax + 2 * ti.seg + 1, MAPPED_NAN_HI);
                }
            }
            if (!BATCHED && lane == 0) worklist[t * 64 + fl] = (int)(t * 64 + fl);   // diagnostics only: which subvectors took this path
        }

        if (PAGED && valid) {
            idx += a.code_base;
            if (a.merge && !(nan_rank(val) > nan_rank(prev_u))) {   // strict: ties (and an earlier NaN) stay with the earlier page
                val = prev_u;
                idx = prev_idx;
            }
        }
        if (PAGED && __ballot(valid && nan_bits(val)) != 0) {   // a NaN kept from an earlier page counts as well
            sawnan = true;
            if (BATCHED && lane == 0) {
                atomicMin(a.seg_minmax + 2 * ti.seg, MAPPED_NAN_LO);
                atomicMax(a.seg_minmax + 2 * ti.seg + 1, MAPPED_NAN_HI);
            }
        }
        if (valid) {
            ti.codes[sv] = (CodeT)idx;
            u[gsv] = val;
            if (!PAGED || a.last_page) {   // (min,max) of the FINAL projections only: earlier pages' values may be replaced
                lmin = fminf(lmin, val);
                lmax = fmaxf(lmax, val);
            }
        }
        ti = tin;
        t = tn;
        tn = BATCHED ? tnn : draw();
    }
    if (BATCHED) {
        flush_minmax();
        __syncthreads();
        if (threadIdx.x < PFD_MM_SEGS) {   // the workgroup's table into the tensors' global words
            const unsigned lo = s_mm[2 * threadIdx.x], hi = s_mm[2 * threadIdx.x + 1];
            if (lo != 0xFFFFFFFFu || hi != 0u) {
                unsigned *mm = a.seg_minmax + 2 * (seg_first + (int)threadIdx.x);
                atomicMin(mm, lo);
                atomicMax(mm + 1, hi);
            }
        }
        return;
    }
    write_minmax_partials<WAVES>(lmin, lmax, ws, sawnan);   // per-workgroup (min,max); the level kernel folds them
}

static int64_t pfd_grid(int64_t ntiles, int bpc, int waves) {
    int64_t blocks = (ntiles + waves - 1) / waves;
    int64_t cap = (int64_t)cu_count() * bpc;
    if (cap > GQ_MAIN_PARTIALS) cap = GQ_MAIN_PARTIALS;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

// LDS of a launch that keeps `pages` pages resident: the f32 image for the exact rescoring + the bf16 fragments
template <int D>
static constexpr size_t pfd_lds_bytes(int pages) {
    constexpr int KS = D > 16 ? D / 16 : 1;
    return (size_t)pages * ((size_t)64 * (4 * D + 4) * sizeof(float) + (size_t)8 * KS * 2 * 64 * 16);
}
// pages per launch: what fits beside the kernel's static LDS (the batched form keeps 24 KiB of segment records there);
// a dynamic allocation that fills the CU's 160 KiB to the last byte is refused at dispatch
template <int D>
static int pfd_pages_per_launch(bool batched) {
    const size_t budget = batched ? (size_t)132 * 1024 : (size_t)150 * 1024;
    int p = (int)(budget / pfd_lds_bytes<D>(1));
    return p < 1 ? 1 : p;
}

// K = 256 * pages (PAGED): as many pages per launch as the LDS holds, launches merged in place (hsq_encode_pf.hip)
template <int D>
static int launch_pfd_paged(const float *grad, const float *codebook, int64_t M, int K, int32_t *codes, float *u, float *ws,
                            hipStream_t st) {
    constexpr int WAVES = D == 32 ? GQ_W32 : (D == 16 ? 8 : GQ_W8), THREADS = WAVES * 64;
    auto kernel = hsq_encode_pfd_kernel<int32_t, D, false, false, true>;
    const int per = pfd_pages_per_launch<D>(false), npages = K / 256;
    static const int bpc = [] {
        const size_t lds = pfd_lds_bytes<D>(pfd_pages_per_launch<D>(false));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_encode_pfd_kernel<int32_t, D, false, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
        return resident_blocks_per_cu(hsq_encode_pfd_kernel<int32_t, D, false, false, true>, THREADS, lds);
    }();
    const int64_t blocks = pfd_grid((M + 63) / 64, bpc, WAVES);
    for (int page = 0; page < npages; page += per) {
        PfdArgs a = {};
        a.grad = grad;
        a.M = M;
        a.codes = codes;
        a.u = u;
        a.cb = codebook + (size_t)page * 256 * D;
        a.ws = ws;
        a.npages = npages - page < per ? npages - page : per;
        a.code_base = page * 256;
        a.merge = page > 0;
        a.last_page = page + a.npages >= npages;
        pfd_split(a, (M + 63) / 64, blocks);
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(THREADS), pfd_lds_bytes<D>(a.npages), st, a);
    }
    GQ_CHECK_LAUNCH("gq_hsq_encode (paged prefilter)");
    return GQ_OK;
}

int launch_encode_pfd_paged(const float *grad, const float *codebook, int64_t M, int d, int K, int32_t *codes, float *u,
                            float *ws, hipStream_t st) {
    if (M > 0x7FFFFFFFLL) return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode: prefilter path needs M < 2^31");
    if (d == 8) return launch_pfd_paged<8>(grad, codebook, M, K, codes, u, ws, st);
    if (d == 16) return launch_pfd_paged<16>(grad, codebook, M, K, codes, u, ws, st);
    if (d == 32) return launch_pfd_paged<32>(grad, codebook, M, K, codes, u, ws, st);
    return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: the paged prefilter kernel was asked for d = %d", d);
}

template <int D>
static int pfd_batched_paged(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                             const float *codebook, int K, int ef, float ef_scale, uint8_t *wire, float *u_flat,
                             uint32_t *seg_minmax, float *ws, hipStream_t st) {
    constexpr int WAVES = D == 32 ? GQ_W32 : (D == 16 ? 8 : GQ_W8), THREADS = WAVES * 64;
    const int per = pfd_pages_per_launch<D>(true), npages = K / 256;
    static const int bpc = [] {
        const size_t lds = pfd_lds_bytes<D>(pfd_pages_per_launch<D>(true));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_encode_pfd_kernel<int32_t, D, true, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_encode_pfd_kernel<int32_t, D, true, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
        return resident_blocks_per_cu(hsq_encode_pfd_kernel<int32_t, D, true, true, true>, THREADS, lds);
    }();
    const int64_t blocks = pfd_grid(ntiles, bpc, WAVES);
    for (int page = 0; page < npages; page += per) {
        PfdArgs a = {};
        a.M = ntiles * 64;
        a.u = u_flat;
        a.cb = codebook + (size_t)page * 256 * D;
        a.ws = ws;
        a.seg_table = seg_table;
        a.tile_seg = tile_seg;
        a.wire = wire;
        a.seg_minmax = seg_minmax;
        a.ntiles = ntiles;
        a.nseg = nseg;
        a.ef_scale = ef_scale;
        a.npages = npages - page < per ? npages - page : per;
        a.code_base = page * 256;
        a.merge = page > 0;
        a.last_page = page + a.npages >= npages;
        pfd_split(a, ntiles, blocks);
        if (ef && page == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pfd_kernel<int32_t, D, true, true, true>), dim3((unsigned)blocks),
                               dim3(THREADS), pfd_lds_bytes<D>(a.npages), st, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pfd_kernel<int32_t, D, true, false, true>), dim3((unsigned)blocks),
                               dim3(THREADS), pfd_lds_bytes<D>(a.npages), st, a);
    }
    GQ_CHECK_LAUNCH("gq_hsq_encode_batched_paged");
    return GQ_OK;
}

int launch_pfd_batched_paged(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                             const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire, float *u_flat,
                             uint32_t *seg_minmax, float *ws, hipStream_t st) {
    if (d == 8)
        return pfd_batched_paged<8>(seg_table, tile_seg, nseg, ntiles, codebook, K, ef, ef_scale, wire, u_flat, seg_minmax, ws, st);
    if (d == 16)
        return pfd_batched_paged<16>(seg_table, tile_seg, nseg, ntiles, codebook, K, ef, ef_scale, wire, u_flat, seg_minmax, ws, st);
    return pfd_batched_paged<32>(seg_table, tile_seg, nseg, ntiles, codebook, K, ef, ef_scale, wire, u_flat, seg_minmax, ws, st);
}

}  // namespace gq
