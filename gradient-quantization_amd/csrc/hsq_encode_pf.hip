// HSQ encode, d = 16, K = 256: bf16x3 matrix-core prefilter + exact f32 rescoring.
//
// The exact f32 MFMA kernel (hsq_encode.hip) is bound by the f32 matrix rate, and on gfx950
// the f32 MFMA shares its datapath with the VALU (measured, tools/enc_probe.hip: their times
// ADD), so the 256-way argmax cannot hide behind it.  This kernel gets the same bits out
// with ~4x less time:
//
//  1. APPROXIMATE scores on the bf16 matrix pipe (which does overlap with the VALU):
//       c = ch + cl + O(2^-18 c),  v = vh + vl + O(2^-18 v)      (bf16 hi/lo splits)
//       s~_k = ch.vh + ch.vl + cl.vh     (3 x v_mfma_f32_32x32x16_bf16, f32 accumulate)
//     |s~_k - p_k| <= E := 2^-15 * max_k||c_k||_1 * max_j |v_j|   for the reference's p_k (fmaf chain):
//     dropped terms 3*2^-18, chain/accumulate roundings ~2^-20, all relative to
//     sum_j |c_kj||v_j| <= ||c_k||_1 max|v_j| <= 4 (1+eps) max|v_j|  (rows are unit L2 norm).
//  2. The 16 scores a lane gets per row block are 8 GROUPS of 2 consecutive codewords
//     (accumulator registers 2p, 2p+1).  Per group the VALU takes g = max |s~| (one v_max_f32
//     with |.| modifiers) and forms ONE key
//       key = (bits(g) & 0x7FFFFFE0) | group_id ,
//     and keeps the TOP-2 group keys per lane (v_med3_u32 / v_max3_u32): ~1.6 VALU ops per
//     score instead of ~3 for a compare/select argmax.  (On gfx950 min/max/med3 and v_and_or
//     issue ~1.6x slower than add/fma -- tools/valu_probe.hip -- so op COUNT is what matters.)
//  3. EXACT rescoring: both codewords of the best group of each half (4 per subvector) are
//     recomputed with the reference's arithmetic, acc = fmaf(c[j], v[j], acc) for j ascending,
//     codebook rows from LDS.  The largest |p| (lowest index on a tie) is the answer IF every
//     other candidate is provably smaller:  upper(second-best group key of either half) + E < |u|.
//     Every candidate that was not rescored lies in a group whose key is <= that bound, so it
//     cannot reach |u| even after the approximation error: code and u are exactly the
//     reference's first-max argmax and projection.
//  4. Otherwise (top-2 gap below ~2e-4 relative, ~5e-4 of random subvectors; tiny / huge / non-finite
//     inputs) the wave stops for a moment and recomputes that subvector exactly against all 256
//     codewords (lane k takes codewords k, k+64, k+128, k+192; wave-wide first-max reduction).
//     Correctness never depends on E being tight -- only on it being an upper bound.
//  5. The last workgroup to finish folds the per-workgroup (min,max) of u into the final (lb, ub):
//     the whole encode is ONE launch.
#include <hip/hip_ext.h>

#include <type_traits>

#include "hsq_pf_common.hpp"

namespace gq {

// The d = 16 kernels run ONE workgroup of 8 waves per CU (two waves per SIMD, as before, but in one workgroup):
// the waves of a workgroup share their tiles through an LDS counter (see the kernel).
constexpr int PF_WAVES = 8;
constexpr int PF_THREADS = PF_WAVES * 64;
constexpr int PF_TAIL = 4;   // swept 2..12 in round 1 (52.3 us at 4..8, 54 at 2 and 12); again at the end of round 3: 3-4 40.35, 6 40.48, 8 40.6, 10 40.8 us
constexpr int PF_LDS_SEGS = 384;            // batched form: tensors whose segment records are kept in LDS (24 KiB)
constexpr int QUAD_STRIDE = 68;             // LDS floats per GROUP of 4 codewords (64 used, 272 B = 17 x 16 B: random groups spread over the banks)

// Arguments of the prefilter kernels.  Single-tensor form: grad/M/codes/u.  Batched form
// (gq_hsq_encode_batched): a segment table describes many tensors that share the codebook; tiles
// never straddle tensors (every tensor starts on a 64-subvector tile boundary of the padded
// index space), so one launch serves all of them with per-tensor (min,max).
struct PfArgs {
    const float *grad;        // single
    int64_t M;                // single: subvectors; batched: ntiles * 64 (padded index space)
    void *codes;              // single
    float *u;                 // single: u[M]; batched: u_flat[ntiles*64]
    const float *cb;
    float *ws;
    const int64_t *seg_table; // batched: int64[8] per segment (include/gq_hsq.h)
    const int32_t *tile_seg;  // batched
    uint8_t *wire;            // batched
    unsigned *seg_minmax;     // batched: order-mapped (min, max) per segment
    int64_t ntiles;
    int nseg;                 // batched: segments in seg_table
    float ef_scale;           // EF kernels: grad <- grad + ef_scale * error (error pointer = seg_table[seg][7], 0 = none)
    int tiles_q, tiles_r;     // tiles per workgroup: the first tiles_r workgroups take tiles_q + 1, the others tiles_q
    int tiles_skew, skew_blocks;   // ... and of the first skew_blocks (even) workgroups the even ones take tiles_skew more, the odd ones as many fewer
};

// The split of the tiles over the grid is made on the host: dividing 64-bit integers in the kernel's prologue was
// ~300 scalar instructions (0.8 us) in front of the first load.
#ifndef GQ_PF_SKEW_PERMILLE
#define GQ_PF_SKEW_PERMILLE 10
#endif
static void pf_split(PfArgs &a, int64_t ntiles, int64_t blocks) {
    a.tiles_q = (int)(ntiles / blocks);
    a.tiles_r = (int)(ntiles % blocks);
    // Workgroups go to the eight XCDs in turn, and on every MI355X measured (six boxes, profiles/r0*_pf_kernel_stamps.txt)
    // the odd XCDs finish the same number of tiles 1.2-1.5 us later than the even ones (same cycles per tile: a lower
    // clock).  The odd workgroups hand ~1 % of their tiles to their even neighbours.
    a.skew_blocks = (int)(blocks & ~(int64_t)1);
    a.tiles_skew = (int)((a.tiles_q * (int64_t)GQ_PF_SKEW_PERMILLE + 500) / 1000);
    if (a.tiles_skew >= a.tiles_q) a.tiles_skew = 0;
}

// SEGLDS (batched only): the segment records are read from their LDS copy (nseg <= PF_LDS_SEGS) or, for
// longer tensor lists, from global memory -- as two instantiations, because a run-time choice between
// the two sources turns the record pointer into a flat pointer (see tile_info).
template <typename CodeT, bool BATCHED, bool EF = false, bool SEGLDS = true>
__global__ __launch_bounds__(PF_THREADS, 1) void hsq_encode_pf_kernel(const PfArgs a) {
    const float *__restrict__ cb = a.cb;
    float *__restrict__ ws = a.ws;
    float *__restrict__ u = a.u;
    const int64_t M = a.M;
    // f32 codebook for the exact rescoring, codeword pairs interleaved element by element:
    // s_cb[(k>>2)*QUAD_STRIDE + 4*j + (k&3)] = c[k][j]
    __shared__ __attribute__((aligned(16))) float s_cb[64 * QUAD_STRIDE];
    __shared__ int s_next;                     // tile counter of this workgroup's run (see below)
    // Batched form: the segment table (64 B per tensor) goes to LDS once.  Looking a tile's tensor up in
    // global memory costs two dependent round trips at the top of every tile (tile -> segment -> record),
    // which made this form 3x slower than the single-tensor one; from LDS the record is ~100 cycles away,
    // and the tile -> segment word is fetched one tile ahead.
    __shared__ int64_t s_seg[(BATCHED && SEGLDS) ? PF_LDS_SEGS * 8 : 1];
    // bf16 hi / lo A fragments of the 8 row blocks as the waves of the workgroup produce them (one row block each)
    __shared__ __attribute__((aligned(16))) u32x4 s_frag[8 * 2 * 64];
    __shared__ float s_c1[PF_WAVES];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;

    const int64_t ntiles = BATCHED ? a.ntiles : ((M + 63) >> 6);
    // Tiles: every workgroup owns one contiguous run [lo, hi) of them and its 8 waves draw from it through an
    // LDS counter (ds_add_rtn: ~100 cycles, so a wave is committed to one tile beyond the one it works on --
    // two in the batched form, which wants the tile -> tensor word a tile early).  The two waves of a SIMD do
    // not run at one speed (the older one wins VALU arbitration, ~1.6x; profiles/r01_e_pf_kernel_stamps.txt)
    // and a static split between them, however tuned, left the slowest wave ~15 % behind the mean; drawn from
    // a shared counter the tiles go to whoever is free.  (Ticket counters in global memory cost an atomic round
    // trip per tile, ~0.7 us, and commit a wave two tiles ahead: measured slower than the static split.)
    // Contiguous runs also keep a wave's running (min,max) with one tensor for many tiles in the batched form.
    const int b = (int)blockIdx.x;
    const bool skewed = b < a.skew_blocks;
    const int64_t lo_tile = (int64_t)b * a.tiles_q + (b < a.tiles_r ? b : a.tiles_r) + ((skewed && (b & 1)) ? a.tiles_skew : 0);
    const int64_t tile_end = lo_tile + a.tiles_q + (b < a.tiles_r ? 1 : 0) + (skewed ? ((b & 1) ? -a.tiles_skew : a.tiles_skew) : 0);
    // The second wave of a SIMD (waves 4-7: the slower of the pair) leaves the last PF_TAIL tiles of the run to
    // the first one: a tile it started that late would finish ~1.5 us after everybody else.
    const int tail_from = (int)(tile_end - lo_tile) - (wave >= PF_WAVES / 2 ? PF_TAIL : 0);
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

    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)
    int cur_seg = -1;  // batched: segment the running (lmin, lmax) belongs to
    f32x4 cur[4], nxt[4];
    f32x4 nxte[4];   // EF: the error tile that goes with nxt (dead otherwise)

    // where tile `tile` lives: base pointer, subvector count of its tensor, local index of its first subvector
    // Pointers that come out of the segment table are integers to the compiler: cast to plain pointers
    // they are FLAT (generic address space) and every access becomes a flat_load / flat_store whose wait
    // is vmcnt(0) AND lgkmcnt(0) -- behind the previous tile's stores, a store round trip per tile (the
    // batched form ran 70 % slower than the single-tensor one).  Address space 1 = global memory.
    typedef const float __attribute__((address_space(1))) *gcf_ptr;
    typedef float __attribute__((address_space(1))) *gf_ptr;
    typedef const f32x4 __attribute__((address_space(1))) *gcv_ptr;
    typedef f32x4 __attribute__((address_space(1))) *gv_ptr;
    typedef CodeT __attribute__((address_space(1))) *gcode_ptr;
    struct Tile {
        gcf_ptr base;
        int64_t m, sv0;
        int rem;       // index of the tile's last valid subvector (0..63): m - 1 - sv0, capped at 63
        int seg;
        gcode_ptr codes;
        gcf_ptr err;   // EF: this tensor's error buffer or nullptr
    };
    auto uniform64 = [](int64_t v) {   // a wave-uniform value read through a vector path -> SGPRs
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    // FROM_GLOBAL: the record comes from global memory even in the SEGLDS form (the first tile is set up
    // before the LDS copy of the table exists)
    auto tile_info = [&](int64_t tile, int seg, auto from_global) {   // seg: the tile's tensor (batched; fetched ahead by the caller)
        constexpr bool FROM_GLOBAL = decltype(from_global)::value;
        Tile ti;
        if (BATCHED) {
            ti.seg = seg;
            // One pointer that may point into LDS or into global memory is a FLAT pointer, and the wait for a
            // flat load (vmcnt AND lgkmcnt) sits right behind the previous tile's stores: a store round trip
            // per tile.  Hence the compile-time choice of the source.
            int64_t r0, r1, r2, r3, r7 = 0;
            if constexpr (SEGLDS && !FROM_GLOBAL) {
                const int64_t *rec = s_seg + 8 * seg;
                r0 = rec[0];
                r1 = rec[1];
                r2 = rec[2];
                r3 = rec[3];
                if (EF) r7 = rec[7];
            } else {
                typedef const int64_t __attribute__((address_space(1))) *grec_ptr;
                const grec_ptr rec = (grec_ptr)(a.seg_table + 8 * (int64_t)seg);
                r0 = rec[0];
                r1 = rec[1];
                r2 = rec[2];
                r3 = rec[3];
                if (EF) r7 = rec[7];
            }
            ti.base = (gcf_ptr)(uintptr_t)uniform64(r0);
            ti.m = uniform64(r1);
            ti.sv0 = (tile - uniform64(r2)) * 64;
            ti.codes = (gcode_ptr)((uintptr_t)a.wire + (uintptr_t)uniform64(r3));
            ti.err = EF ? (gcf_ptr)(uintptr_t)uniform64(r7) : (gcf_ptr)0;
            const int64_t rem = ti.m - 1 - ti.sv0;
            ti.rem = rem > 63 ? 63 : (int)rem;
        } else {
            ti.seg = 0;
            ti.base = (gcf_ptr)a.grad;
            ti.m = M;
            ti.sv0 = tile * 64;
            ti.codes = (gcode_ptr)static_cast<CodeT *>(a.codes);
            ti.err = (gcf_ptr)0;
            const int64_t rem = M - 1 - ti.sv0;
            ti.rem = rem > 63 ? 63 : (int)rem;
        }
        return ti;
    };
    // Addresses inside a tile: a wave-uniform 64-bit base (the tile's first subvector: scalar arithmetic) plus a 32-bit
    // lane offset, so that the loads and stores take an SGPR base and the per-lane part is two VALU operations per
    // block (clamp, scale) instead of ~10 of 64-bit compare / select / shift / add.  lane_sv: this lane's subvector of
    // each block, clamped to the tile's last valid one (tail: re-read it, the result is masked).
    auto lane_off = [&](const Tile &ti, int blk) {   // float offset of v[8h..] of subvector blk*32+j from the tile's first float
        const unsigned li = min((unsigned)(blk * 32 + j), (unsigned)ti.rem);
        return li * 16u + 8u * (unsigned)h;
    };
    auto load_tile = [&](const Tile &ti, f32x4(&dst)[4]) {
        const gcf_ptr tb = ti.base + ti.sv0 * 16;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const gcv_ptr p = (gcv_ptr)(tb + lane_off(ti, blk));
            dst[2 * blk] = p[0];
            dst[2 * blk + 1] = p[1];
        }
    };
    auto load_err = [&](const Tile &ti, f32x4(&dst)[4]) {
        if (EF && ti.err) {
            const gcf_ptr tb = ti.err + ti.sv0 * 16;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const gcv_ptr p = (gcv_ptr)(tb + lane_off(ti, blk));
                dst[2 * blk] = p[0];
                dst[2 * blk + 1] = p[1];
            }
        }
    };
    // v = grad + scale*error, written back over grad (valid subvectors only)
    auto fold_err = [&](const Tile &ti, f32x4(&g)[4], const f32x4(&e)[4]) {
        if (EF && ti.err) {
            const gf_ptr tb = (gf_ptr)(ti.base + ti.sv0 * 16);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 &x = g[2 * blk + q];
                    const f32x4 &y = e[2 * blk + q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float prod = a.ef_scale * y[c];
                        x[c] = x[c] + prod;
                    }
                }
                if (blk * 32 + j <= ti.rem) {
                    const gv_ptr p = (gv_ptr)(tb + (unsigned)((blk * 32 + j) * 16 + 8 * h));
                    p[0] = g[2 * blk];
                    p[1] = g[2 * blk + 1];
                }
            }
        }
    };
    auto flush_minmax = [&]() {  // batched: fold this wave's running (min,max) into its segment
        const float lo = wave_min(lmin), hi = wave_max(lmax);
        if (lane == 0 && cur_seg >= 0 && lo <= hi) {
            // Look before the atomic: a tensor's (min,max) words are hit by every wave that touched the
            // tensor (all ~2000 of them for one big tensor, ~90 atomics/us per address: a 23 us tail),
            // but only the first few still improve them.  The words only ever move towards the extremes,
            // so a value that is already as good as ours -- however stale -- makes ours redundant.
            unsigned *mm = a.seg_minmax + 2 * cur_seg;
            const unsigned mlo = order_map(lo), mhi = order_map(hi);
            const unsigned seen_lo = __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned seen_hi = __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mlo < seen_lo) atomicMin(mm, mlo);
            if (mhi > seen_hi) atomicMax(mm + 1, mhi);
        }
        lmin = INFINITY;
        lmax = -INFINITY;
    };

    // B fragments: lane (col j, half h) holds v[8h .. 8h+7] of subvector j of each block
    bf16x8 vh[2], vl[2];
    Tile ti = {};
    auto seg_of = [&](int64_t tile) {   // batched: tile -> tensor, one global word (0 beyond the end)
        return (BATCHED && tile < tile_end) ? a.tile_seg[tile] : 0;
    };

    // ---- prologue.  Everything the workgroup needs from memory is requested first, in the order it is used
    // (vector-memory results return in order): the codebook words for the LDS image, this wave's row block of
    // the codebook (A fragments: every wave splits ONE of the 8 row blocks into bf16 hi / lo and shares it
    // through LDS -- each wave splitting all 8 cost 0.8 us of VALU time per SIMD), and the wave's first tile,
    // whose HBM latency then hides behind the staging.  One barrier.  (Before: staging, fragments, barrier,
    // ||c||_1, barrier and only then the first tile's loads: 5.2 us; profiles/r02_b_pf_prologue_stamps.txt.)
    float cbv[256 * 16 / PF_THREADS];
#pragma unroll
    for (int n = 0; n < 256 * 16 / PF_THREADS; ++n) cbv[n] = cb[threadIdx.x + n * PF_THREADS];
    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(cb + (wave * 32 + j) * 16 + 8 * h);
    const f32x4 q1 = *reinterpret_cast<const f32x4 *>(cb + (wave * 32 + j) * 16 + 8 * h + 4);
    if (t < tile_end) {
        ti = tile_info(t, BATCHED ? __builtin_amdgcn_readfirstlane(a.tile_seg[t]) : 0, std::true_type{});
        load_tile(ti, cur);
        load_err(ti, nxte);
    }
    if (threadIdx.x == 0) s_next = PF_WAVES;   // the first PF_WAVES tiles of the run go to the waves by index
#pragma unroll
    for (int n = 0; n < 256 * 16 / PF_THREADS; ++n) {
        const int i = threadIdx.x + n * PF_THREADS, k = i >> 4, jj = i & 15;
        s_cb[(k >> 2) * QUAD_STRIDE + 4 * jj + (k & 3)] = cbv[n];
    }
    if (BATCHED && SEGLDS) {
        const int n = (a.nseg < PF_LDS_SEGS ? a.nseg : PF_LDS_SEGS) * 8;
        for (int i = threadIdx.x; i < n; i += PF_THREADS) s_seg[i] = a.seg_table[i];
    }
    {
        bf16x8 fh, fl;
        split8(q0, q1, fh, fl);
        s_frag[(wave * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, fh);
        s_frag[(wave * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, fl);
        // The error bound scales with max_k ||c_k||_1 (<= 4 for unit-L2 rows); measure it instead of
        // trusting the caller's codebook to be normalised.  Row wave*32 + j: this lane's 8 elements + its partner's.
        float l1 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) l1 += fabsf(q0[e]) + fabsf(q1[e]);
        l1 += __shfl_xor(l1, 32, 64);
        l1 = wave_max(l1);
        if (lane == 0) s_c1[wave] = l1;
    }
    __syncthreads();
    // A fragments of v_mfma_f32_32x32x16_bf16: lane (row j, half h) holds c[rb*32+j][8h .. 8h+7];
    // hi and lo bf16 parts, 64 VGPRs, resident for the kernel's lifetime.  (Keeping them in LDS
    // instead and running 4 waves/SIMD measured slower: the kernel is bound by VALU issue, not
    // by latency.)
    bf16x8 ch[8], cl[8];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        ch[rb] = __builtin_bit_cast(bf16x8, s_frag[(rb * 2 + 0) * 64 + lane]);
        cl[rb] = __builtin_bit_cast(bf16x8, s_frag[(rb * 2 + 1) * 64 + lane]);
    }
    float c1 = s_c1[0];
#pragma unroll
    for (int w = 1; w < PF_WAVES; ++w) c1 = fmaxf(c1, s_c1[w]);
    const float err_scale = c1 * ERR_SCALE;  // E = max|v_j| * err_scale

    int64_t tn = draw();                // the tile after this wave's first one
    int seg_n = seg_of(tn);             // in flight while the first tile is set up
    int seg_next = 0;                   // its value, read back BEFORE a tile's stores (see the consume point)
    if (t < tile_end) {
        seg_next = BATCHED ? __builtin_amdgcn_readfirstlane(seg_n) : 0;
        fold_err(ti, cur, nxte);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) split8(cur[2 * blk], cur[2 * blk + 1], vh[blk], vl[blk]);
    }
    while (t < tile_end) {
        // single tensor: tn was drawn at the end of the previous tile; batched: a whole tile ago
        const int64_t tnn = BATCHED ? draw() : 0;
        Tile tin = ti;
        if (tn < tile_end) {
            tin = tile_info(tn, seg_next, std::false_type{});
            load_tile(tin, nxt);  // prefetch the next tile
            load_err(tin, nxte);
        }
        seg_n = seg_of(tnn);
        if (BATCHED && ti.seg != cur_seg) {
            flush_minmax();
            cur_seg = ti.seg;
        }
        // ---- prefilter: 16 (block, row block) chains; top-2 GROUP keys per (block, row-block half) ----
        // The three MFMAs of chain c+1 depend on each other and issue is in order, so they are
        // placed one by one BETWEEN the key operations of chain c (sched_barrier pins the order):
        // the matrix pipe runs under the VALU stream.
        unsigned best[4] = {0, 0, 0, 0}, second[4] = {0, 0, 0, 0};
        unsigned vmask = KEY_MASK;
        asm volatile("" : "+v"(vmask));  // keep the mask in a VGPR: v_and_or with an SGPR operand issues slower
        auto group_key = [&](const f32x16 &a, int rb, int q) {   // group q = registers 4q..4q+3 = four consecutive rows
            return and_or(__float_as_uint(absmax4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3])), vmask,
                          (unsigned)((rb & 3) * 4 + q));
        };
        // Top-2 of the tracker and the FOUR keys of a chain in five operations (two steps of three were six):
        //   t = max3(best, k0, k1)   a = med3(best, k0, k1)      -- first and second of {best, k0, k1}
        //   b = med3(t, k2, k3)      best' = max3(t, k2, k3)     -- second of {t, k2, k3}: k2 / k3 if they stay below t, else t or the smaller of them
        //   second' = max3(second, a, b)
        // (the second largest of {best, k0..k3} is max(a, b): both of the top two lie in {t, a, k2, k3}.)
        unsigned trk_t = 0, trk_a = 0;
        auto track_lo = [&](int trk, unsigned k0, unsigned k1) {
            trk_t = max3u(best[trk], k0, k1);
            trk_a = med3u(best[trk], k0, k1);
        };
        auto track_hi = [&](int trk, unsigned k2, unsigned k3) {
            const unsigned b = med3u(trk_t, k2, k3);
            best[trk] = max3u(trk_t, k2, k3);
            second[trk] = max3u(second[trk], trk_a, b);
        };
        f32x16 acc = {0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[0], vh[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[0], vl[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[0], vh[0], acc, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int rb = c & 7, trk = c >> 2;
            if (c + 1 < 16) {
                const int nb = (c + 1) >> 3, nr = (c + 1) & 7;
                f32x16 nacc = {0};
                __builtin_amdgcn_sched_barrier(0);
                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[nr], vh[nb], nacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                track_lo(trk, group_key(acc, rb, 0), group_key(acc, rb, 1));
                __builtin_amdgcn_sched_barrier(0);
                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vl[nb], nacc, 0, 0, 0);
                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vh[nb], nacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                track_hi(trk, group_key(acc, rb, 2), group_key(acc, rb, 3));
                __builtin_amdgcn_sched_barrier(0);
                acc = nacc;
            } else {
                track_lo(trk, group_key(acc, rb, 0), group_key(acc, rb, 1));
                track_hi(trk, group_key(acc, rb, 2), group_key(acc, rb, 3));
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
            k1[blk] = ((gid >> 2) + (useB ? 4 : 0)) * 32 + 8 * (gid & 3) + 4 * h;   // rows k1 .. k1+3 (registers 4q..4q+3)
            s2[blk] = max3u(second[2 * blk], second[2 * blk + 1], bl) | 31u;         // upper end of its bucket
            bk[blk] = bw;
        }

        // ---- this lane's own full subvector (tile subvector `lane`): 8 swaps of the B loads ----
        float vf[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = cur[e >> 2][e & 3];        // block 0: floats 8h+e of subvector j
            float y = cur[2 + (e >> 2)][e & 3];  // block 1
            swap32(x, y);                        // x = floats e (0..7), y = floats 8+e of subvector `lane`
            vf[e] = x;
            vf[8 + e] = y;
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

        // ---- exact rescoring of the better of the two halves' best groups (2 codewords; the
        // reference's fmaf chain).  The other half's best group joins the bound on everything that
        // was not rescored.
        const bool pick1 = (bk[1] & KEY_MASK) > (bk[0] & KEY_MASK);
        const int kc = pick1 ? k1[1] : k1[0];
        const unsigned rest = max3u(s2[0], s2[1], (pick1 ? bk[0] : bk[1]) | 31u);
        const f32x4 p4 = exact_score_quad<16>(s_cb + (kc >> 2) * QUAD_STRIDE, vf);   // kc is a multiple of 4: one quad
        float val = p4[0];
        int idx = kc;
        take_if_greater(val, idx, p4[1], kc + 1);
        take_if_greater(val, idx, p4[2], kc + 2);
        take_if_greater(val, idx, p4[3], kc + 3);

        float vmax = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e += 2) vmax = fmaxf(fmaxf(fabsf(vf[e]), fabsf(vf[e + 1])), vmax);
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
        // NaN anywhere makes the comparisons false -> not safe -> exact fix-up path

        const bool valid = lane <= ti.rem;         // this lane's subvector exists (the tensor's last tile may be short)

        // Consume the prefetched tile (convert it to the next B fragments) BEFORE this tile's
        // stores are issued: the wait for the prefetch then sees only long-finished memory ops.
        // Done the other way round, the compiler's vmcnt wait at the first use of `nxt` sits right
        // behind the just-issued stores and every tile eats a store round trip.
        bf16x8 nvh[2], nvl[2];
        // the tile -> tensor word of the tile after next was requested at the top of this tile: read it
        // back here, with the prefetch, not behind the stores
        if (BATCHED) seg_next = __builtin_amdgcn_readfirstlane(seg_n);
        if (EF && tn < tile_end) fold_err(tin, nxt, nxte);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) split8(nxt[2 * blk], nxt[2 * blk + 1], nvh[blk], nvl[blk]);
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            vh[blk] = nvh[blk];
            vl[blk] = nvl[blk];
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- exact fix-up, in place and wave-wide, for the few subvectors the bound could not settle
        // (~5e-4 of random ones: a wave meets one every ~30 tiles).  The flagged lane's subvector is
        // broadcast through SGPRs; lane k scores codewords k, k+64, k+128, k+192 with the reference's
        // fmaf chain from the LDS codebook; a wave-wide first-max reduction picks the winner.
        uint64_t todo = __ballot(valid && !safe);
        while (todo) {
            const int fl = __builtin_ctzll(todo);
            todo &= todo - 1;
            float w[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                w[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vf[e]), fl));
            float bv = 0.0f;
            int bi = lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = q * 64 + lane;
                const float *row = s_cb + (k >> 2) * QUAD_STRIDE + (k & 3);
                float acc = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc = __fmaf_rn(row[4 * e], w[e], acc);
                if (q == 0) {
                    bv = acc;
                } else {
                    take_if_greater_nan(bv, bi, acc, k);   // torch.argmax's order: NaN is the largest, the first one wins
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
            if (!BATCHED && lane == 0) worklist[ti.sv0 + fl] = (int)(ti.sv0 + fl);   // diagnostics only: which subvectors took this path
        }

        if (valid) {   // uniform bases (the tile's first code / projection) + the lane index
            (ti.codes + ti.sv0)[(unsigned)lane] = (CodeT)idx;
            ((gf_ptr)u + (BATCHED ? t * 64 : ti.sv0))[(unsigned)lane] = val;
            lmin = fminf(lmin, val);
            lmax = fmaxf(lmax, val);
        }
        ti = tin;
        t = tn;
        tn = BATCHED ? tnn : draw();
    }
    if (BATCHED) {
        flush_minmax();
        return;
    }
    write_minmax_partials<PF_WAVES>(lmin, lmax, ws, sawnan);   // per-workgroup (min,max); the level kernel folds them
}

// one resident wave of 8-wave workgroups (the (min,max) slots of the workspace cap the grid)
static int64_t pf16_grid(int64_t ntiles, int bpc) {
    int64_t blocks = (ntiles + PF_WAVES - 1) / PF_WAVES;
    int64_t cap = (int64_t)cu_count() * bpc;
    if (cap > GQ_MAIN_PARTIALS) cap = GQ_MAIN_PARTIALS;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

template <typename CodeT>
int launch_encode_pf(const float *grad, const float *codebook, int64_t M, CodeT *codes, float *u, float *ws,
                     hipStream_t st, int profile_slot) {
    if (M > 0x7FFFFFFFLL) return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode: prefilter path needs M < 2^31");
    static const int bpc = resident_blocks_per_cu(hsq_encode_pf_kernel<CodeT, false>, PF_THREADS, 0);
    PfArgs a = {};
    a.grad = grad;
    a.M = M;
    a.codes = codes;
    a.u = u;
    a.cb = codebook;
    a.ws = ws;
    const int64_t blocks = pf16_grid((M + 63) / 64, bpc);
    pf_split(a, (M + 63) / 64, blocks);
    hipEvent_t ev_start, ev_stop;
    if (profile_events(profile_slot, &ev_start, &ev_stop)) {   // events attached to this dispatch (gq_profile_read)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<CodeT, false>), dim3((unsigned)blocks),
                              dim3(PF_THREADS), 0, st, ev_start, ev_stop, 0, a);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<CodeT, false>), dim3((unsigned)blocks),
                           dim3(PF_THREADS), 0, st, a);
    }
    GQ_CHECK_LAUNCH("gq_hsq_encode (prefilter)");
    return GQ_OK;
}

template int launch_encode_pf<uint8_t>(const float *, const float *, int64_t, uint8_t *, float *, float *, hipStream_t, int);
template int launch_encode_pf<int32_t>(const float *, const float *, int64_t, int32_t *, float *, float *, hipStream_t, int);

}  // namespace gq

namespace gq {
template <bool EF>
static int encode_batched(const char *what, const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                          const float *codebook, uint8_t *wire, float *u_flat, uint32_t *seg_minmax, float *workspace,
                          float ef_scale, int profile_slot, void *stream) {
    if (nseg < 1 || ntiles < 1 || ntiles * 64 > 0x7FFFFFFFLL)
        return fail(GQ_ERR_INVALID_ARG, "%s: bad sizes nseg=%d ntiles=%lld", what, nseg, (long long)ntiles);
    if (!seg_table || !tile_seg || !codebook || !wire || !u_flat || !seg_minmax || !workspace)
        return fail(GQ_ERR_INVALID_ARG, "%s: null pointer", what);
    static const int bpc = resident_blocks_per_cu(hsq_encode_pf_kernel<uint8_t, true, EF, true>, PF_THREADS, 0);
    PfArgs a = {};
    a.M = ntiles * 64;
    a.u = u_flat;
    a.cb = codebook;
    a.ws = workspace;
    a.seg_table = seg_table;
    a.tile_seg = tile_seg;
    a.wire = wire;
    a.seg_minmax = seg_minmax;
    a.ntiles = ntiles;
    a.nseg = nseg;
    a.ef_scale = ef_scale;
    hipStream_t st = as_stream(stream);
    const int64_t blocks = pf16_grid(ntiles, bpc);
    pf_split(a, ntiles, blocks);
    hipEvent_t ev_start, ev_stop;
    if (nseg <= PF_LDS_SEGS && profile_events(profile_slot, &ev_start, &ev_stop)) {   // events attached to this dispatch
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<uint8_t, true, EF, true>), dim3((unsigned)blocks),
                              dim3(PF_THREADS), 0, st, ev_start, ev_stop, 0, a);
    } else if (nseg <= PF_LDS_SEGS) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<uint8_t, true, EF, true>), dim3((unsigned)blocks),
                           dim3(PF_THREADS), 0, st, a);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<uint8_t, true, EF, false>), dim3((unsigned)blocks),
                           dim3(PF_THREADS), 0, st, a);
    }
    GQ_CHECK_LAUNCH(what);
    return GQ_OK;
}
}  // namespace gq

namespace gq {
int launch_pfd_batched_paged(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                             const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire, float *u_flat,
                             uint32_t *seg_minmax, float *ws, hipStream_t st);   // hsq_encode_pfd.hip

}  // namespace gq

GQ_INTERNAL int gqi_hsq_encode_batched_paged(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                       const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire,
                                       float *u_flat, uint32_t *seg_minmax, float *workspace, void *stream) {
    if (nseg < 1 || ntiles < 1 || ntiles * 64 > 0x7FFFFFFFLL)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched (paged): bad sizes nseg=%d ntiles=%lld", nseg,
                        (long long)ntiles);
    if (!seg_table || !tile_seg || !codebook || !wire || !u_flat || !seg_minmax || !workspace)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched (paged): null pointer");
    if ((d != 8 && d != 16 && d != 32) || K <= 256 || (K & 255) != 0 || K > 65536)
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode_batched (paged): d must be 8, 16 or 32 and K a multiple of 256 above 256");
    if (nseg > (d == 16 ? gq::PF_LDS_SEGS : 384))
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode_batched (paged): at most 384 tensors per launch");
    hipStream_t st = gq::as_stream(stream);
    return gq::launch_pfd_batched_paged(seg_table, tile_seg, nseg, ntiles, codebook, d, K, ef, ef_scale, wire, u_flat,
                                        seg_minmax, workspace, st);
}

GQ_INTERNAL int gqi_hsq_encode_batched_d16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                           const float *codebook, int ef, float ef_scale, uint8_t *wire, float *u_flat,
                                           uint32_t *seg_minmax, float *workspace, int profile_slot, void *stream) {
    if (ef)
        return gq::encode_batched<true>("gq_hsq_encode_batched", seg_table, tile_seg, nseg, ntiles, codebook, wire, u_flat,
                                        seg_minmax, workspace, ef_scale, profile_slot, stream);
    return gq::encode_batched<false>("gq_hsq_encode_batched", seg_table, tile_seg, nseg, ntiles, codebook, wire, u_flat,
                                     seg_minmax, workspace, 0.0f, profile_slot, stream);
}
