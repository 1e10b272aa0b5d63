// HSQ encode, K <= 256 (256 in the text below; round 6: fewer codewords as zero rows, K <= 32 / <= 64 on one / two of the eight
// row blocks -- template parameter NRB), sub-dimension D = 8, 16 or 32 (12 / 24 as padded 16 / 32 -- DR): f16 matrix-core prefilter
// (ONE MFMA per chain and 16 dimensions) +
// exact f32 rescoring + deferred exact scans.  Same bits as the exact f32 kernel (hsq_encode.hip) in about a quarter of
// its time.  Written for D = 16 (BASELINE's shape; the text below describes that instantiation); round 5 made D a template
// parameter (main.py:90's default is --c-dim 32): a chain is KS = ceil(D / 16) MFMAs, D = 8 feeds zeros for the upper half
// of its one k-step, D = 32 keeps the codebook's f16 fragments in LDS (64 VGPRs otherwise) and a 32-entry ring per wave.
//
// The exact f32 MFMA is bound by the f32 matrix rate and shares its datapath with the VALU (tools/enc_probe.hip: their
// times ADD), so the 256-way argmax cannot hide behind it.  This kernel scores approximately on the f16 matrix pipe
// (which overlaps with the VALU), settles nearly every subvector from those scores plus FOUR exact scores, and hands
// the rest to an exact scan:
//
//  1. APPROXIMATE scores, one v_mfma_f32_32x32x16_f16 per chain of 32 codewords x 32 subvectors:
//       ch = f16(c),   v' = v * sigma,   vh = f16(v'),   s~_k = ch_k . vh      (products exact in f32, f32 accumulate)
//     sigma is a power of two, ONE per wave and tile (a scalar, so the conversion is one v_fma_mix per element): it
//     keeps the tile's values in the middle of the f16 range and follows the data from tile to tile (below).
//     Why one MFMA: round 3 ran three bf16 MFMAs per chain (ch.vh + ch.vl + cl.vh, error 2^-15).  Measured in round 4
//     (profiles/r04_encode_dvfs.txt): what an MFMA costs this kernel is not its issue slot or its 32 cycles of matrix
//     pipe but the CLOCK -- with the B operands zeroed the same instruction stream ran 18 % faster (2.06 GHz against
//     1.83); dropping MFMAs gave 3 -> 2 per chain -13 %, 2 -> 1 another -16 % (an 8-bit lo term: -6 %).  One f16 MFMA
//     rounds both operands ONCE to 11 bits; the error bound is ~30 x round 3's, ~1 % of N(0,1) subvectors are not
//     settled by it, and what makes that affordable is the batched exact scan of (4).
//     Error against the reference's p_k (fmaf chain), in scaled units:
//       s'_k - s~_k = (c_k - ch_k) . v'  +  ch_k . (v' - vh)
//       |ch_k . (v' - vh)| <= ||ch_k||_2 * 2^-11 sqrt(n2),  n2 = ||vh||_2^2: the rounding error of element j is at most half an
//                             ulp <= 2^-11 |vh_j| (subnormal or zero vh_j: 2^-25, in err_abs)
//       |(c_k - ch_k) . v'| <= dc ||v'||_2,   dc = max_k ||c_k - ch_k||_2 -- the codebook's rounding residuals as they are,
//                             measured in the prologue (~0.6 x 2^-11 for unit rows) --,  ||v'||_2 <= (1 + 2^-11) sqrt(n2)
//       E' := sqrt(n2) * (1.03 * 2^-11 * c2 + 1.001 dc) + 2^-21 * c2,    c2 = max_k ||c_k||_2 (measured too);
//     the 1.03 covers the accumulation roundings of the MFMA, of the reference's chain and of n2 itself (< 1 % of the first term).
//     Valid while 2^-12 <= n2 <= 2^24 (every |vh_j| <= 2^12: no f16 overflow -- an overflowed element makes n2 infinite):
//     outside that window -- a subvector far off the wave's scale -- the subvector goes to the exact scan.  n2 costs four
//     v_dot2_f32_f16 per fragment.  (Rounds 4-5: n2 = sum of 4^exponent, four v_and_b32 more; see scale8_f16.)
//  2. The 16 scores a lane gets per chain are 4 GROUPS of 4 consecutive codewords (accumulator registers 4q..4q+3).
//     Per group the VALU takes g = max |s~| (v_max3_f32 + v_max_f32 with |.| modifiers) and forms ONE key
//       key = (bits(g) & 0x7FFFFFE0) | group_id ,
//     and keeps the TOP-2 group keys per lane and half tile in five operations per chain (v_max3_u32 / v_med3_u32).
//     (On gfx950 min / max / med3 / and_or issue ~1.6x slower than add / fma -- tools/oprate.hip -- so op COUNT matters.)
//  3. EXACT rescoring: the four codewords of the best group are recomputed with the reference's arithmetic,
//     acc = fmaf(c[j], v[j], acc) for j ascending, codebook rows from LDS.  The largest |p| (lowest index on a tie)
//     is the answer IF every other candidate is provably smaller:  upper(second-best key) + E' < |p| sigma.
//     Every codeword that was not rescored lies in a group whose key is <= that bound, so it cannot reach |p| even
//     after the approximation error: code and u are exactly the reference's first-max argmax and projection.
//  4. Otherwise (~1.5 % of N(0,1) subvectors -- about one per tile --; anything outside the window) the subvector is
//     QUEUED: its 16 floats and its destination go to the wave's ring in LDS (written and read by this wave only).  When
//     the wave has run out of tiles (or 32 are waiting) the ring goes through a SECOND PASS of the same prefilter with
//     everything the first one left out -- the codebook's lo part, the entry's lo part, a scale of the entry's own --, i.e.
//     three MFMAs per chain on a half tile whose columns are the queued subvectors; its bound is ~150 x tighter.  What
//     is still open after that (~1 % of the queued; non-finite values) is scanned exactly by the whole wave, one entry
//     at a time, each lane 4 codewords (scan1; round 4: a quarter wave per entry, four at a time).  Round 3 stopped the
//     whole wave for ONE unsettled subvector at a time (1,500 scans cost 1.5 us of a 42 us launch); here 24,000 second-pass entries and 200 scans cost ~4 us of a launch that is
//     16 % shorter (profiles/r04_encode_ab.txt).  Correctness never depends on a bound being tight -- only on it being
//     an upper bound.
//  5. Per-workgroup (min, max) of u go to the workspace; the level kernel folds them into (lb, ub).
#include <hip/hip_ext.h>

#include <type_traits>

#include "hsq_pf_common.hpp"

// DIAGNOSTIC build only (-DGQ_PF_STAMPS: build.py's libgq_hsq_clock.so, loaded by `bench.py --clock-child` and tools/stamp_read.py;
// never the product library): s_memtime stamps around the phases of a tile and s_memrealtime (100 MHz) at the ends of the
// loop, written behind the diagnostics log of the workspace, which no other code of the kernel reads
// (MI355X_MICROARCH.md, 'DVFS give-back' item 6: the in-kernel clock = shader cycles / real time).  Without the macro
// every GQ_STAMP is empty and the kernel contains no stamp.
#ifdef GQ_PF_STAMPS
#define GQ_STAMP(ID)                                                                                  \
    {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long ts_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        stamp_acc[ID] += (ts_ - ts_prev);                                                             \
        ts_prev = ts_;                                                                                \
    }
#define GQ_STAMPS_ONLY(...) __VA_ARGS__
#define GQ_LOG_SCAN(...)
#else
#define GQ_STAMP(ID)
#define GQ_STAMPS_ONLY(...)
#define GQ_LOG_SCAN(...) __VA_ARGS__
#endif

namespace gq {

// The d = 16 kernels run ONE workgroup of 8 waves per CU (two waves per SIMD, as before, but in one workgroup):
// the waves of a workgroup share their tiles through an LDS counter (see the kernel).
constexpr int PF_WAVES = 8;
constexpr int PF_THREADS = PF_WAVES * 64;
#ifndef GQ_PF_TAIL
#define GQ_PF_TAIL 4
#endif
constexpr int PF_TAIL = GQ_PF_TAIL;   // swept 2..12 in round 1 (52.3 us at 4..8, 54 at 2 and 12); again at the end of round 3: 3-4 40.35, 6 40.48, 8 40.6, 10 40.8 us
// End of a workgroup's run.  Every wave takes its ring through the second pass when its tiles are gone: a fixed ~1.2 us
// for ~11 entries, two waves per SIMD at once, on the launch's critical path (profiles/r05_encode_ab.txt: second pass +
// leftover scans = 3.9 of 36.5 us).  Round 5 tried the obvious cure -- a wave flushes its ring through the second pass
// once, when the tile it is about to start lies within PF_FLUSH_AHEAD tiles of the end of its workgroup's run, so that the
// other waves draw the remaining tiles meanwhile, and scans the 0-3 entries of its last tiles exactly (scan1) -- and
// measured it SLOWER at every distance (8 ... 32 tiles: 36.7-38.5 us against 35.7; same file, block B): a pass in the loop
// takes its issue slots from the SIMD's other wave, whose tiles are the critical path by then.  The switch stays (0 = off)
// for the record; scan1 serves the leftovers of a pass.
#ifndef GQ_PF_FLUSH_AHEAD
#define GQ_PF_FLUSH_AHEAD 0
#endif
#ifndef GQ_PF_PAIR
#define GQ_PF_PAIR -1   // -1: per sub-dimension (PfShape::PAIR); 0 / 1: off / on for every D (A/B builds)
#endif
#ifndef GQ_PF_FLUSH_MIN
#define GQ_PF_FLUSH_MIN 3
#endif
#ifndef GQ_PF_SCAN1_MAX
#define GQ_PF_SCAN1_MAX 4
#endif
constexpr int PF_FLUSH_AHEAD = GQ_PF_FLUSH_AHEAD;   // 0: no early flush (round 4's behaviour)
constexpr int PF_FLUSH_MIN = GQ_PF_FLUSH_MIN;       // entries that make an early flush worth a pass
constexpr int PF_SCAN1_MAX = GQ_PF_SCAN1_MAX;       // leftovers up to this many are scanned one by one by the whole wave
constexpr int PF_LDS_SEGS = 384;            // batched form: tensors whose segment records are kept in LDS (24 KiB)
// What depends on the sub-dimension.
template <int D>
struct PfShape {
    static_assert(D == 8 || D == 16 || D == 32, "the prefilter kernel is built for D = 8, 16 and 32");
    static constexpr int KS = D > 16 ? D / 16 : 1;    // MFMA k-steps (of 16 dimensions) per chain
    static constexpr bool HALF = D < 16;              // D = 8: the upper half of the one k-step is zero (lanes 32-63 hold no data)
    static constexpr int QS = 4 * D + 4;              // LDS floats per GROUP of 4 codewords (4 D used; an odd number of 16-byte units: random groups spread over the banks)
    static constexpr int QCAP = D > 16 ? 32 : 64;     // deferred subvectors a wave can hold (a ring in LDS: D floats each)
    static constexpr bool A_REGS = D <= 16;           // the codebook's f16 A fragments stay in registers (32 VGPRs); D = 32: read from LDS a row block ahead
    static constexpr int NF = 4 * KS;                 // f32x4 registers of a tile per lane: [(block * KS + k-step) * 2 + q]
    // the two waves of a SIMD end their runs as a pair (the end of the kernel).  Measured (profiles/r05_encode_ab.txt, block C):
    // D = 32 30.52 against 31.24 us, D = 16 33.83 against 32.96, D = 8 55.4 against 54.6 -- on for D = 32 only
    static constexpr bool PAIR = GQ_PF_PAIR < 0 ? D > 16 : GQ_PF_PAIR != 0;
    // Error bound of the f16 prefilter (header, 1): E' = sqrt(n2) * (ERR_REL * c2 + 1.001 dc) + ERR_ABS * c2 inside the window of n2.
    // The factor over 2^-11 covers the accumulation roundings of the MFMA and of the reference's chain, 2 D 2^-24 of
    // sum |c_j v_j| <= c2 ||v'||_2 <= 1.001 c2 sqrt(n2) together (0.4 % of the first term for D = 16, 0.8 % for D = 32), the
    // rounding of n2's own sum and square root (< 0.001 %) and ||ch_k||_2 <= c2 + dc (0.03 %).
    static constexpr float ERR_REL = (D > 16 ? 1.05f : 1.03f) * 4.8828125e-04f;     // x 2^-11
    // second pass (ch.vh + ch.vl + cl.vh): cl.vl and the splits' remainders (3 x 2^-22 x 2.02) + 4 D accumulated products
    // (x 2^-24 x 2.02): 1.19 x 2^-17 for D = 16, 2.21 x 2^-17 for D = 32; the codebook split's subnormal grid on top (ERR2_SUB)
    static constexpr float ERR2_REL = (D > 16 ? 2.3f : 1.2f) * 7.62939453125e-06f;  // x 2^-17
    static constexpr float ERR2_SUB = D > 16 ? 3.5e-7f : 2.4e-7f;                   // 2.02 sqrt(D) 2^-25
};
constexpr float ERR_ABS = 4.76837158203125e-07f;      // 2^-21 (subnormal f16 results: sqrt(D) 2^-25 per subvector)
constexpr unsigned N2_LO_BITS = 0x39800000u;          // 2^-12
constexpr unsigned N2_HI_BITS = 0x4B800000u;          // 2^24  (n2 = ||vh||_2^2: every |vh_j| <= 2^12)
constexpr int SIGMA_TARGET_EXP2 = 10;                 // the largest sampled n2 of a tile is steered to ~2^10 (norm 2^5 .. 2^6)

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
    int K;                    // codewords (<= 32 NRB; a multiple of 4): rows K .. 255 of the image and of the fragments are zeros
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
// DR (round 6): the tensors' REAL sub-dimension when the reference repaired it to one this kernel has no shape for
// (nearest_neighbor_compressor.py:23-29: 16 -> 24, 8 -> 12): rows of DR floats in memory (gradient, error buffer, codebook), D = the
// next of 16 / 32 everywhere else -- the missing elements are zeros in the fragments, in the LDS image and in the lane's subvector,
// and fmaf(0, 0, acc) == acc bit for bit (acc starts at +0 and can never become -0), so scores, codes and u are the exact kernels'.
// NRB (round 6): row blocks of 32 codewords that are scored -- 8 for K = 256 (and any K above 64), 2 for K <= 64, 1 for K <= 32
// (--k-bit 5 / 6, and K == dim: nearest_neighbor_compressor.py:40-47): 2 NRB chains per tile instead of 16.  Rows K .. 32 NRB - 1
// are zeros (image and fragments): their scores are +0 for every finite subvector, so they lose every comparison against a real
// row or tie with one of a lower index; the exact scan (where the non-finite subvectors end up) looks at rows below K only.
template <typename CodeT, int D, bool BATCHED, bool EF = false, bool SEGLDS = true, int DR = D, int NRB = 8>
__global__ __launch_bounds__(PF_THREADS, 1) void hsq_encode_pf_kernel(const PfArgs a) {
    typedef PfShape<D> SH;
    static_assert(DR == D || (D == 16 && DR == 12) || (D == 32 && DR == 24), "padded shapes: 12 -> 16, 24 -> 32");
    static_assert(NRB == 1 || NRB == 2 || NRB == 4 || NRB == 8, "row blocks: 1, 2, 4 or 8");
    const int K = a.K;
    constexpr int KS = SH::KS, QS = SH::QS, QCAP = SH::QCAP, NF = SH::NF;
    constexpr bool HALF = SH::HALF, PF_PAIR = SH::PAIR;
    GQ_STAMPS_ONLY(const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime(); unsigned long long nscanned = 0, npassed = 0;)
    const float *__restrict__ cb = a.cb;
    float *__restrict__ ws = a.ws;
    float *__restrict__ u = a.u;
    const int64_t M = a.M;
    // f32 codebook for the exact rescoring, codeword pairs interleaved element by element:
    // s_cb[(k>>2)*QS + 4*j + (k&3)] = c[k][j]
    __shared__ __attribute__((aligned(16))) float s_cb[64 * QS];
    __shared__ int s_next;                     // tile counter of this workgroup's run (see below)
    __shared__ int s_pairdone[PF_WAVES / 2], s_ringpub[PF_WAVES];   // PF_PAIR: waves of the pair that have ended their runs / a wave's ring (head << 8 | count) at that point
    // Batched form: the segment table (64 B per tensor) goes to LDS once.  Looking a tile's tensor up in
    // global memory costs two dependent round trips at the top of every tile (tile -> segment -> record),
    // which made this form 3x slower than the single-tensor one; from LDS the record is ~100 cycles away,
    // and the tile -> segment word is fetched one tile ahead.
    __shared__ int64_t s_seg[(BATCHED && SEGLDS) ? PF_LDS_SEGS * 8 : 1];
    // f16 hi / lo A fragments of the 8 row blocks as the waves of the workgroup produce them (one row block each)
    __shared__ __attribute__((aligned(16))) u32x4 s_frag[8 * KS * 64], s_fragl[8 * KS * 64];   // [(row block * KS + k-step) * 64 + lane]: hi parts (to registers; D = 32: read per chain) / lo parts (read by the second pass)
    __shared__ float s_c1[PF_WAVES], s_dc[PF_WAVES];
    // deferred exact scans (header, 4): every wave's own ring of PF_QCAP entries = the subvector's 16 floats +
    // {code address lo, hi, index into u, segment}.  Written and read by the same wave only: no flags, no atomics.
    __shared__ __attribute__((aligned(16))) float s_qv[PF_WAVES * QCAP * D];
    __shared__ __attribute__((aligned(16))) unsigned s_qm[PF_WAVES * QCAP * 4];

    // Batched form: the workgroup's (min, max) per tensor, for the PF_MM_SEGS tensors from its first tile's on (a run is
    // contiguous: typically 1-3 tensors, a few dozen where the list has many small ones).  Waves fold into it with LDS
    // atomics; ONE pair of global atomics per (workgroup, tensor) when the workgroup ends.  Before, every wave looked at the
    // tensor's global words (an agent-scope load: a trip to the memory side, ~2 us) and then hit them, at every change of
    // tensor and at its end -- 10 us of a 48 us launch on the ResNet-50 list (tools/graph_pieces.py, profiles/r04_minmax_lds.txt).
    constexpr int PF_MM_SEGS = 64;
    __shared__ unsigned s_mm[BATCHED ? 2 * PF_MM_SEGS : 2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform to the compiler too: tile indices and everything derived from them stay in scalar registers)
    const int j = lane & 31, h = lane >> 5;

    const int64_t ntiles = BATCHED ? a.ntiles : ((M + 63) >> 6);
    // Tiles: every workgroup owns one contiguous run [lo, hi) of them and its 8 waves draw from it through an
    // LDS counter (ds_add_rtn: ~100 cycles, so a wave is committed to one tile beyond the one it works on).
    // The two waves of a SIMD do
    // not run at one speed (the older one wins VALU arbitration, ~1.6x; profiles/r01_e_pf_kernel_stamps.txt)
    // and a static split between them, however tuned, left the slowest wave ~15 % behind the mean; drawn from
    // a shared counter the tiles go to whoever is free.  (Ticket counters in global memory cost an atomic round
    // trip per tile, ~0.7 us, and commit a wave two tiles ahead: measured slower than the static split.)
    // Contiguous runs also keep a wave's running (min,max) with one tensor for many tiles in the batched form.
    const int b = (int)blockIdx.x;
    const bool skewed = b < a.skew_blocks;
    // (tile indices are 32-bit: the launchers refuse more than 2^31 subvectors = 2^25 tiles.  As 64-bit integers every comparison
    // of the loop top was a VECTOR compare -- the scalar unit has no ordered 64-bit compare -- whose result the scalar branch
    // then waited for: profiles/r06_encode_ab.txt, block L)
    const int lo_tile = b * a.tiles_q + (b < a.tiles_r ? b : a.tiles_r) + ((skewed && (b & 1)) ? a.tiles_skew : 0);
    const int tile_end = lo_tile + a.tiles_q + (b < a.tiles_r ? 1 : 0) + (skewed ? ((b & 1) ? -a.tiles_skew : a.tiles_skew) : 0);
    // The second wave of a SIMD (waves 4-7: the slower of the pair) leaves the last PF_TAIL tiles of the run to
    // the first one: a tile it started that late would finish ~1.5 us after everybody else.
    const int tail_from = (tile_end - lo_tile) - (wave >= PF_WAVES / 2 ? PF_TAIL : 0);
    auto draw = [&]() {   // the next tile of this workgroup's run (may lie beyond tile_end)
        int k = 0x3FFFFFFF;
        if (lane == 0) {
            if (__hip_atomic_load(&s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tail_from)
                k = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return lo_tile + __builtin_amdgcn_readfirstlane(k);
    };
    int t = lo_tile + wave;
    int *const worklist = ws_worklist(ws);

    float lmin = INFINITY, lmax = -INFINITY;
    bool sawnan = false;   // a projection of this wave is NaN (wave-uniform)
    int cur_seg = -1;  // batched: segment the running (lmin, lmax) belongs to
    int seg_first = 0; // batched: the tensor of this workgroup's first tile (s_mm's entry 0)
    f32x4 cur[NF], nxt[NF];   // [(block * KS + k-step) * 2 + q]: floats [16 s + 8 h + 4 q, + 4) of subvector 32 block + j
    f32x4 nxte[NF];  // EF: the error tile that goes with nxt (dead otherwise)
    if (HALF) {      // D = 8: lanes 32-63 never load (their half of the k-step is zero)
#pragma unroll
        for (int i = 0; i < NF; ++i) cur[i] = nxt[i] = nxte[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

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
        int tile0;       // batched: the tensor's first tile
        int tiles_end;   // batched: the tile behind the tensor's last one
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
    auto tile_info = [&](int tile, int seg, auto from_global) {   // seg: the tile's tensor (batched; fetched ahead by the caller)
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
            ti.tile0 = (int)uniform64(r2);
            ti.tiles_end = ti.tile0 + (int)((ti.m + 63) >> 6);
            ti.sv0 = (int64_t)(tile - ti.tile0) * 64;
            ti.codes = (gcode_ptr)((uintptr_t)a.wire + (uintptr_t)uniform64(r3));
            ti.err = EF ? (gcf_ptr)(uintptr_t)uniform64(r7) : (gcf_ptr)0;
            const int rem = (int)ti.m - 1 - (int)ti.sv0;      // (both below 2^31: 32-bit, a scalar compare)
            ti.rem = rem > 63 ? 63 : rem;
        } else {
            ti.seg = 0;
            ti.tile0 = 0;
            ti.tiles_end = 0;
            ti.base = (gcf_ptr)a.grad;
            ti.m = M;
            ti.sv0 = (int64_t)tile * 64;
            ti.codes = (gcode_ptr)static_cast<CodeT *>(a.codes);
            ti.err = (gcf_ptr)0;
            const int rem = (int)M - 1 - (int)ti.sv0;      // (M < 2^31: the launcher refuses more)
            ti.rem = rem > 63 ? 63 : rem;
        }
        return ti;
    };
    // Addresses inside a tile: a wave-uniform 64-bit base (the tile's first subvector: scalar arithmetic) plus a 32-bit
    // lane offset, so that the loads and stores take an SGPR base and the per-lane part is two VALU operations per
    // block (clamp, scale) instead of ~10 of 64-bit compare / select / shift / add.  lane_sv: this lane's subvector of
    // each block, clamped to the tile's last valid one (tail: re-read it, the result is masked).
    auto lane_off = [&](const Tile &ti, int blk) {   // float offset of v[8h..] of subvector blk*32+j from the tile's first float
        const unsigned li = min((unsigned)(blk * 32 + j), (unsigned)ti.rem);
        return li * (unsigned)DR + (HALF ? 0u : 8u * (unsigned)h);
    };
    // floats [16 s + 8 h + 4 q, + 4) of a row exist (DR < D: the upper lanes' last fragments do not -- they stay zero and are
    // neither read nor written: the row behind the tensor's last one is not ours)
    auto frag_ok = [&](int s_, int q_) { return DR == D || 16 * s_ + 4 * q_ + 4 + 8 * h <= DR; };
    const bool loads_here = !HALF || h == 0;   // D = 8: the lower lanes hold the subvectors, the upper ones zeros
    auto load_rows = [&](gcf_ptr tb, const Tile &ti, f32x4(&dst)[NF]) {
        if (loads_here) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const gcv_ptr p = (gcv_ptr)(tb + lane_off(ti, blk));
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                    dst[(blk * KS + s) * 2] = frag_ok(s, 0) ? p[4 * s] : z;
                    dst[(blk * KS + s) * 2 + 1] = frag_ok(s, 1) ? p[4 * s + 1] : z;
                }
            }
        }
    };
    auto load_tile = [&](const Tile &ti, f32x4(&dst)[NF]) { load_rows(ti.base + ti.sv0 * DR, ti, dst); };
    auto load_err = [&](const Tile &ti, f32x4(&dst)[NF]) {
        if (EF && ti.err) load_rows(ti.err + ti.sv0 * DR, ti, dst);
    };
    // v = grad + scale*error, written back over grad (valid subvectors only)
    auto fold_err = [&](const Tile &ti, f32x4(&g)[NF], const f32x4(&e)[NF]) {
        if (EF && ti.err) {
            const gf_ptr tb = (gf_ptr)(ti.base + ti.sv0 * DR);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int q = 0; q < 2 * KS; ++q) {
                    f32x4 &x = g[2 * KS * blk + q];
                    const f32x4 &y = e[2 * KS * blk + q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float prod = a.ef_scale * y[c];
                        x[c] = x[c] + prod;
                    }
                }
                if (blk * 32 + j <= ti.rem && loads_here) {
                    const gv_ptr p = (gv_ptr)(tb + (unsigned)((blk * 32 + j) * DR + (HALF ? 0 : 8 * h)));
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        if (frag_ok(s, 0)) p[4 * s] = g[(blk * KS + s) * 2];
                        if (frag_ok(s, 1)) p[4 * s + 1] = g[(blk * KS + s) * 2 + 1];
                    }
                }
            }
        }
    };
    // (min, max) candidates of tensor `seg` (order-mapped): into the workgroup's LDS table, or -- a tensor beyond its window --
    // straight into the tensor's global words.  There: look before the atomic.  A tensor's words are hit by every wave that
    // touched it (~90 atomics/us per address), but only the first few still improve them; the words only ever move towards
    // the extremes, so a value that is already as good as ours -- however stale -- makes ours redundant.
    auto mm_fold = [&](int seg, unsigned mlo, unsigned mhi) {
        const unsigned idx = (unsigned)(seg - seg_first);
        if (idx < (unsigned)PF_MM_SEGS) {
            atomicMin(&s_mm[2 * idx], mlo);
            atomicMax(&s_mm[2 * idx + 1], mhi);
        } else {
            unsigned *mm = a.seg_minmax + 2 * seg;
            if (mlo < __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(mm, mlo);
            if (mhi > __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mm + 1, mhi);
        }
    };
    auto flush_minmax = [&]() {  // batched: fold this wave's running (min,max) into its segment
        const float lo = wave_min(lmin), hi = wave_max(lmax);
        if (lane == 0 && cur_seg >= 0 && lo <= hi) mm_fold(cur_seg, order_map(lo), order_map(hi));
        lmin = INFINITY;
        lmax = -INFINITY;
    };

    // B fragments: lane (col j, half h) holds f16(sigma * v[8h .. 8h+7]) of subvector j of each block;
    // n2p: sum of 4^exponent over those eight values (half of the subvector's n2: the header's error bound)
    half8 vh[2 * KS];   // [block * KS + k-step]
    float n2p[2] = {0.0f, 0.0f};
    Tile ti = {};

    // ---- prologue.  Everything the workgroup needs from memory is requested first, in the order it is used
    // (vector-memory results return in order): the codebook words for the LDS image, this wave's row block of
    // the codebook (A fragments: every wave splits ONE of the 8 row blocks into bf16 hi / lo and shares it
    // through LDS -- each wave splitting all 8 cost 0.8 us of VALU time per SIMD), and the wave's first tile,
    // whose HBM latency then hides behind the staging.  One barrier.  (Before: staging, fragments, barrier,
    // ||c||_1, barrier and only then the first tile's loads: 5.2 us; profiles/r02_b_pf_prologue_stamps.txt.)
    float cbv[256 * DR / PF_THREADS];
#pragma unroll
    for (int n = 0; n < 256 * DR / PF_THREADS; ++n) cbv[n] = (int)threadIdx.x + n * PF_THREADS < K * DR ? cb[threadIdx.x + n * PF_THREADS] : 0.0f;
    f32x4 aq[2 * KS];   // row wave*32 + j, k-step s: floats [16 s + 8 h, + 8)
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) aq[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (loads_here && wave * 32 + j < K) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (frag_ok(s, 0)) aq[2 * s] = *reinterpret_cast<const f32x4 *>(cb + (wave * 32 + j) * DR + 16 * s + (HALF ? 0 : 8 * h));
            if (frag_ok(s, 1)) aq[2 * s + 1] = *reinterpret_cast<const f32x4 *>(cb + (wave * 32 + j) * DR + 16 * s + (HALF ? 0 : 8 * h) + 4);
        }
    }
    const int seg_first_v = (BATCHED && lo_tile < ntiles) ? a.tile_seg[lo_tile] : 0;   // (requested with the first tile's word)
    if (t < tile_end) {
        ti = tile_info(t, BATCHED ? __builtin_amdgcn_readfirstlane(a.tile_seg[t]) : 0, std::true_type{});
        load_tile(ti, cur);
        load_err(ti, nxte);
    }
    if (threadIdx.x == 0) {
        s_next = PF_WAVES;   // the first PF_WAVES tiles of the run go to the waves by index
    }
    if (threadIdx.x < PF_WAVES / 2) s_pairdone[threadIdx.x] = 0;
    seg_first = BATCHED ? __builtin_amdgcn_readfirstlane(seg_first_v) : 0;
    if (BATCHED && threadIdx.x < PF_MM_SEGS) {
        s_mm[2 * threadIdx.x] = 0xFFFFFFFFu;
        s_mm[2 * threadIdx.x + 1] = 0u;
    }
#pragma unroll
    for (int n = 0; n < 256 * DR / PF_THREADS; ++n) {
        const int i = threadIdx.x + n * PF_THREADS, k = i / DR, jj = i % DR;
        s_cb[(k >> 2) * QS + 4 * jj + (k & 3)] = cbv[n];
    }
    if constexpr (DR != D) {      // the image's padding columns: zeros (the rescoring multiplies them with the subvector's zeros)
#pragma unroll
        for (int n = 0; n < 256 * (D - DR) / PF_THREADS; ++n) {
            const int i = threadIdx.x + n * PF_THREADS, k = i / (D - DR), jj = DR + i % (D - DR);
            s_cb[(k >> 2) * QS + 4 * jj + (k & 3)] = 0.0f;
        }
    }
    if (BATCHED && SEGLDS) {
        const int n = (a.nseg < PF_LDS_SEGS ? a.nseg : PF_LDS_SEGS) * 8;
        for (int i = threadIdx.x; i < n; i += PF_THREADS) s_seg[i] = a.seg_table[i];
    }
    {
        // This wave's row block of the codebook as an f16 A fragment (ONE rounding per entry, no lo part), and what the
        // error bound needs of the codebook, measured instead of trusting the caller's normalisation: the largest
        // ||c_k||_2 and the largest ||c_k - f16(c_k)||_2 (the rounding residuals as they are: x - f16(x) is exact in f32).
        // Row wave*32 + j: this lane's 8 elements + its partner's.  A row that is not finite or beyond the f16 range makes
        // a maximum NaN or >= 2^30 -- tested on the bits below.
        float l2 = 0.0f, d2 = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            half8 fh, fl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = e < 4 ? aq[2 * s][e] : aq[2 * s + 1][e - 4];
                fh[e] = (_Float16)x;
                const float rr = x - (float)fh[e];
                fl[e] = (_Float16)rr;   // c = hi + lo + r, |r| <= 2^-22 |c| + 2^-25 (the second term: f16's subnormal grid)
                l2 = __fmaf_rn(x, x, l2);
                d2 = __fmaf_rn(rr, rr, d2);
            }
            s_frag[(wave * KS + s) * 64 + lane] = __builtin_bit_cast(u32x4, fh);
            s_fragl[(wave * KS + s) * 64 + lane] = __builtin_bit_cast(u32x4, fl);
        }
        l2 += __shfl_xor(l2, 32, 64);
        d2 += __shfl_xor(d2, 32, 64);
        l2 = wave_max_nan(l2);
        d2 = wave_max_nan(d2);
        if (lane == 0) {
            s_c1[wave] = l2;
            s_dc[wave] = d2;
        }
    }
    __syncthreads();
    // A fragments of v_mfma_f32_32x32x16_f16: lane (row j, half h) holds f16(c[rb*32+j][16s+8h .. +7]); D <= 16: 32 VGPRs,
    // resident for the kernel's lifetime; D = 32 (64 VGPRs: they do not fit beside two tiles of 32 floats per lane) reads them
    // from s_frag a row block ahead (frag_hi).
    half8 ch[SH::A_REGS ? 8 : 1];
    if constexpr (SH::A_REGS) {
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) ch[rb] = __builtin_bit_cast(half8, s_frag[rb * 64 + lane]);
    }
    auto frag_hi = [&](int rb, int s) -> half8 {   // (D <= 16: s == 0, a register)
        if constexpr (SH::A_REGS) return ch[rb];
        else return __builtin_bit_cast(half8, s_frag[(rb * KS + s) * 64 + lane]);
    };
    unsigned c2b = __float_as_uint(s_c1[0]), dcb = __float_as_uint(s_dc[0]);   // squares: non-negative, so the bits order them, a NaN above everything
#pragma unroll
    for (int w = 1; w < PF_WAVES; ++w) {
        c2b = max(c2b, __float_as_uint(s_c1[w]));
        dcb = max(dcb, __float_as_uint(s_dc[w]));
    }
    // E' = sqrt(n2) * err_rel + err_abs (header, 1): the tile's rounding seen through the largest row, plus the codebook's
    // rounding seen through ||vh||_2 < 2 sqrt(n2).  A codebook outside the f16 range: +infinity, nothing is ever "safe".
    const float c2 = __builtin_sqrtf(__uint_as_float(c2b)) * 1.0000002f, dc = __builtin_sqrtf(__uint_as_float(dcb)) * 1.0000002f;
    const bool cb_ok = c2b < 0x4E800000u && dcb < 0x4E800000u;
    const float err_rel = cb_ok ? c2 * SH::ERR_REL + 1.001f * dc : INFINITY;   // n2 = ||vh||_2^2: ||v'||_2 <= (1 + 2^-11) sqrt(n2) (+ the subnormal grid: err_abs)
    const float err_abs = cb_ok ? c2 * ERR_ABS : INFINITY;
    const float err2_rel = cb_ok ? c2 * SH::ERR2_REL + SH::ERR2_SUB : INFINITY;   // the second pass (three MFMAs per chain and k-step)

    int tn = draw();                    // the tile after this wave's first one
    // sigma: the power of two the tile in flight was multiplied by before its conversion to f16 (wave-uniform, an SGPR).
    // First tile: the largest |element| of the tile goes to [2^4, 2^5); afterwards every tile's scale follows the norms
    // the previous tile showed (four sampled lanes, below).  A subvector that ends up outside the window of the error
    // bound is queued for the exact scan, so a bad guess costs time, never bits.
    // (the scale's exponent field lives in a scalar register: the compiler takes everything behind `t < tile_end` for
    // lane-dependent -- t starts from the wave's index -- so the value is read back through readfirstlane where it changes)
    float sigma_t = 1.0f;
    if (t < tile_end) {
        fold_err(ti, cur, nxte);
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < NF; ++i) m = fmaxf(fmaxf(fabsf(cur[i][0]), fabsf(cur[i][1])), fmaxf(fmaxf(fabsf(cur[i][2]), fabsf(cur[i][3])), m));
        const unsigned em = (unsigned)__builtin_amdgcn_readfirstlane((int)(__float_as_uint(wave_max(m)) >> 23));
        if (em >= 1u && em <= 254u) {
            int f = 258 - (int)em;
            f = f < 27 ? 27 : (f > 227 ? 227 : f);
            sigma_t = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(f << 23));
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int s = 0; s < KS; ++s) scale8_f16(cur[(blk * KS + s) * 2], cur[(blk * KS + s) * 2 + 1], sigma_t, vh[blk * KS + s], n2p[blk]);
    }
    // (min, max) of one exactly scanned projection into its tensor's words (batched form)
    auto fold_seg = [&](int seg, float v) {
        const unsigned mv = order_map(v);
        mm_fold(seg, mv, mv);
    };
    auto poison_seg = [&](int seg) {   // (lb, ub) of this tensor become NaN (torch.min / torch.max propagate it)
        atomicMin(a.seg_minmax + 2 * seg, MAPPED_NAN_LO);
        atomicMax(a.seg_minmax + 2 * seg + 1, MAPPED_NAN_HI);
    };
    int qhead = 0, qcnt = 0;   // this wave's ring: first entry, entries (wave-uniform)
    bool flushed = false;      // the early flush has run (wave-uniform)
    const int run_len = tile_end - lo_tile;
    // Exact scan of ONE ring entry by the whole wave: lane l scores codewords 4l .. 4l+3 (its own quad of the LDS image: 64
    // different quads, conflict-free; the entry's floats are a broadcast read), first maximum inside the lane in index order,
    // then across the wave by DPP moves (wave_first_max_nan_dpp: lower lanes hold lower indices).  ~450 cycles; for the one
    // or two entries a second pass leaves open (round 4's quarter-wave scan spent a batch of ~1,200 cycles on up to four).
    auto scan1 = [&](int slot) {   // slot: entry of s_qv / s_qm (any wave's ring)
        GQ_STAMPS_ONLY(nscanned += 1;)
        float w[D];
        {
            const f32x4 *qv = reinterpret_cast<const f32x4 *>(s_qv + D * slot);
#pragma unroll
            for (int q = 0; q < D / 4; ++q) {
                const f32x4 x = qv[q];
                w[4 * q] = x[0];
                w[4 * q + 1] = x[1];
                w[4 * q + 2] = x[2];
                w[4 * q + 3] = x[3];
            }
        }
        f32x4 p = exact_score_quad<D>(s_cb + lane * QS, w);
        if (4 * lane >= K) p = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // rows behind the codebook (K is a multiple of 4): +0, whatever 0 x inf gave -- never the first maximum
        float bv = p[0];
        int bi = 4 * lane;
        take_if_greater_nan(bv, bi, p[1], 4 * lane + 1);
        take_if_greater_nan(bv, bi, p[2], 4 * lane + 2);
        take_if_greater_nan(bv, bi, p[3], 4 * lane + 3);
        wave_first_max_nan_dpp(bv, bi);
        const bool isnan = nan_bits(bv);     // (every lane holds the result)
        if (isnan) sawnan = true;
        if (lane == 0) {
            const u32x4 m = *reinterpret_cast<const u32x4 *>(s_qm + 4 * slot);
            *(gcode_ptr)(uintptr_t)((uint64_t)m[0] | ((uint64_t)m[1] << 32)) = (CodeT)bi;
            ((gf_ptr)u)[m[2]] = bv;
            if (BATCHED) {
                if (isnan) poison_seg((int)m[3]);
                else fold_seg((int)m[3], bv);
            } else {
                GQ_LOG_SCAN(worklist[m[2]] = (int)m[2];)   // diagnostics only: which subvectors took an exact scan
                lmin = fminf(lmin, bv);
                lmax = fmaxf(lmax, bv);
            }
        }
    };
    // SECOND PASS over up to 32 ring entries from `first` on (header, 4): the same prefilter with everything it left out --
    // the codebook's lo part (cl from LDS), the entry's lo part (vl = f16(v' - vh), one v_fma_mix with an f16 addend) and a
    // scale of the ENTRY's own (its largest |element| goes to [2^4, 2^5): entries that fell out of the wave's window are at
    // home here) -- three MFMAs per chain, error ~2^-17 instead of ~2^-10.  Entry i is column i of the one block; lanes
    // (i, 0) and (i, 1) hold the two halves of its rows and, after the exchange, both score the same group (only (i, 0)
    // stores).  What even this bound does not settle (~1 in 150 entries; non-finite values) is scanned exactly (scan1).
    // Columns 0 .. na-1 are entries first .. of THIS wave's ring, columns na .. na+nb-1 entries first_b .. of wave_b's ring
    // (the SIMD partner's, handed over at the end of the run: see PF_PAIR below); n = na + nb <= 32.
    auto second_pass = [&](int first, int na, int wave_b, int first_b, int nb) {
        const int n = na + nb;
        GQ_STAMPS_ONLY(npassed += n;)
        auto slot_of = [&](int c) {   // column -> entry of s_qv / s_qm
            return c < na ? wave * QCAP + ((first + c) & (QCAP - 1)) : wave_b * QCAP + ((first_b + c - na) & (QCAP - 1));
        };
        const int col = j < n ? j : 0;   // (idle columns repeat entry 0 and store nothing)
        const int slot = slot_of(col);
        const f32x4 *qv = reinterpret_cast<const f32x4 *>(s_qv + D * slot);
        f32x4 xs[2 * KS];   // the entry's floats [16 s + 8 h, + 8) for every k-step (D = 8: zeros in the upper lanes)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (HALF) {
                const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                xs[0] = h ? z : qv[0];
                xs[1] = h ? z : qv[1];
            } else {
                xs[2 * s] = qv[4 * s + 2 * h];
                xs[2 * s + 1] = qv[4 * s + 2 * h + 1];
            }
        }
        auto both_halves = [&](float own, bool add) {   // own (+ or max) the partner lane's value: two copies through v_permlane32_swap
            float a = own, b = own;
            swap32(a, b);            // a = the lower lanes' values in both halves, b = the upper lanes'
            return add ? a + b : fmaxf(a, b);
        };
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i)
            m = fmaxf(fmaxf(fmaxf(fabsf(xs[i][0]), fabsf(xs[i][1])), fmaxf(fabsf(xs[i][2]), fabsf(xs[i][3]))), m);
        m = both_halves(m, false);
        const unsigned em = __float_as_uint(m) >> 23;
        int f = 258 - (int)em;
        f = f < 27 ? 27 : (f > 227 ? 227 : f);
        const float sig = (em >= 1u && em <= 254u) ? __uint_as_float((unsigned)f << 23) : 1.0f;   // per lane: the entry's scale
        // vh = f16(x sig), vl = f16(x sig - vh): the second conversion's fma is exact in f32 (a rounding residual)
        half8 bh[KS], bl[KS];
        float n2q = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 H, L;
            const f32x4 x0 = xs[2 * s], x1 = xs[2 * s + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
                unsigned r, l;
                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(sig));
                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r) : "v"(b), "v"(sig));
                asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(sig), "v"(r));
                asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(sig), "v"(r));
                const unsigned e = r & 0x7C007C00u;
                asm("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(n2q) : "v"(e));
                H[i] = r;
                L[i] = l;
            }
            bh[s] = __builtin_bit_cast(half8, H);
            bl[s] = __builtin_bit_cast(half8, L);
        }
        const float n2 = both_halves(n2q, true);
        // ---- 8 chains: the two trackers of the one block
        unsigned best2[2] = {0, 0}, second2[2] = {0, 0};
        unsigned vmask = KEY_MASK;
        asm volatile("" : "+v"(vmask));
        // (as in the tile loop: the MFMAs of chain rb + 1 are issued in front of the key operations of chain rb; the lo
        // fragments -- D = 32: the hi ones too -- come from LDS one chain ahead)
        half8 al[KS], ah[KS];
        auto frags_of = [&](int rb) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                al[s] = __builtin_bit_cast(half8, s_fragl[(rb * KS + s) * 64 + lane]);
                ah[s] = frag_hi(rb, s);
            }
        };
        auto chain3 = [&](f32x16 c) {   // the three products of a chain, every k-step: cl.vh + ch.vl + ch.vh
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], c, 0, 0, 0);
            }
            return c;
        };
        frags_of(0);
        f32x16 acc = {0};
        acc = chain3(acc);
        if (NRB > 1) frags_of(1);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            f32x16 nacc = {0};
            if (rb + 1 < NRB) {
                __builtin_amdgcn_sched_barrier(0);
                nacc = chain3(nacc);
                if (rb + 2 < NRB) frags_of(rb + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            unsigned k[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                k[q] = and_or(__float_as_uint(absmax4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3])), vmask,
                              (unsigned)((rb & 3) * 4 + q));
            const int trk = rb >> 2;
            const unsigned t0 = max3u(best2[trk], k[0], k[1]), a0 = med3u(best2[trk], k[0], k[1]);
            const unsigned b0 = med3u(t0, k[2], k[3]);
            best2[trk] = max3u(t0, k[2], k[3]);
            second2[trk] = max3u(second2[trk], a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            acc = nacc;
        }
        const bool useB = (best2[1] & KEY_MASK) > (best2[0] & KEY_MASK);
        const unsigned bw = useB ? best2[1] : best2[0], blo = useB ? best2[0] : best2[1];
        const int gid = (int)(bw & 31u);
        const int kmine = ((gid >> 2) + (useB ? 4 : 0)) * 32 + 8 * (gid & 3) + 4 * h;
        const unsigned smine = max3u(second2[0], second2[1], blo) | 31u;
        // both lanes of a column get both halves' (best key, its group, bound on the rest)
        int k0 = kmine, k1_ = kmine;
        swap32(k0, k1_);
        int s0 = (int)smine, s1 = (int)smine;
        swap32(s0, s1);
        int w0 = (int)bw, w1 = (int)bw;
        swap32(w0, w1);
        const bool pick1 = ((unsigned)w1 & KEY_MASK) > ((unsigned)w0 & KEY_MASK);
        const int kc = pick1 ? k1_ : k0;
        const unsigned rest = max3u((unsigned)s0, (unsigned)s1, (unsigned)(pick1 ? w0 : w1) | 31u);
        float vf[D];
#pragma unroll
        for (int q = 0; q < D / 4; ++q) {
            const f32x4 x = qv[q];
            vf[4 * q] = x[0];
            vf[4 * q + 1] = x[1];
            vf[4 * q + 2] = x[2];
            vf[4 * q + 3] = x[3];
        }
        const f32x4 p4 = exact_score_quad<D>(s_cb + (kc >> 2) * QS, vf);
        float val = p4[0];
        int idx = kc;
        take_if_greater(val, idx, p4[1], kc + 1);
        take_if_greater(val, idx, p4[2], kc + 2);
        take_if_greater(val, idx, p4[3], kc + 3);
        const unsigned n2b = __float_as_uint(n2);
        const float E = __builtin_amdgcn_sqrtf(n2) * err2_rel + err_abs;
        bool safe = ((n2b - N2_LO_BITS) <= (N2_HI_BITS - N2_LO_BITS)) && (__uint_as_float(rest) + E < fabsf(val * sig));
        if (nan_bits(val)) safe = false;
        const bool owner = h == 0 && j < n;
        if (owner && safe) {
            const u32x4 mt = *reinterpret_cast<const u32x4 *>(s_qm + 4 * slot);
            *(gcode_ptr)(uintptr_t)((uint64_t)mt[0] | ((uint64_t)mt[1] << 32)) = (CodeT)idx;
            ((gf_ptr)u)[mt[2]] = val;
            if (BATCHED) {
                fold_seg((int)mt[3], val);
            } else {
                GQ_LOG_SCAN(worklist[mt[2]] = (int)mt[2];)   // diagnostics only: which subvectors took the second pass
                lmin = fminf(lmin, val);
                lmax = fmaxf(lmax, val);
            }
        }
        uint64_t left = __ballot(owner && !safe);
        while (left) {   // (rare) one exact scan per entry that is still open
            const int i = __builtin_ctzll(left);
            left &= left - 1;
            scan1(slot_of(i));   // the whole wave on the one entry: ~450 cycles
        }
    };
    GQ_STAMPS_ONLY(const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(); unsigned long long ntl = 0, stamp_acc[6] = {0, 0, 0, 0, 0, 0}, ts_prev;
                   asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_prev)::"memory");)
    while (t < tile_end) {
        // tn was drawn at the end of the previous tile.  (The multi-tensor form used to draw a tile earlier still, to have the
        // tile -> tensor word of the tile after next in flight for a whole tile: a wave was committed to TWO tiles beyond its
        // own, and the ends of the workgroups' runs were that much more ragged -- the same elements as one flat tensor ran
        // 35.4 us, as two tensors 40.4.  A tensor's tiles are one contiguous range: the word is only looked up when tn leaves it.)
        // Chain order.  D <= 16 (A fragments in registers): block 0's eight row blocks, then block 1's.  D = 32 (A fragments
        // from LDS): (rb, block 0), (rb, block 1), so that both blocks share a row block's fragments, fetched a row block ahead.
        auto chain_rb = [](int c) { return SH::A_REGS ? (c % NRB) : (c >> 1); };
        auto chain_blk = [](int c) { return SH::A_REGS ? (c / NRB) : (c & 1); };
        half8 af[2][KS];   // D = 32: the row block in use / the next one
        if constexpr (!SH::A_REGS) {
#pragma unroll
            for (int s = 0; s < KS; ++s) af[0][s] = frag_hi(0, s);
        }
        auto chain = [&](int c, f32x16 x) {
            const int rb = chain_rb(c), blk = chain_blk(c);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                x = __builtin_amdgcn_mfma_f32_32x32x16_f16(SH::A_REGS ? ch[SH::A_REGS ? rb : 0] : af[rb & 1][s], vh[blk * KS + s], x, 0, 0, 0);
            return x;
        };
        // The FIRST chain's MFMA goes out before the tile's bookkeeping (round 6): its 16 passes run under the scalar work and the
        // issue of the next tile's loads instead of under a row of s_nop in front of the first key operation.
        f32x16 acc = {0};
        acc = chain(0, acc);
        __builtin_amdgcn_sched_barrier(0);
        Tile tin = ti;
        if (tn < tile_end) {
            // a wave's consecutive tiles mostly stay with one tensor (contiguous runs): its record is already in scalar
            // registers -- the LDS reads and the eight v_readfirstlane of a look-up only when the tensor changes
            if (BATCHED && tn >= ti.tile0 && tn < ti.tiles_end) {
                tin.sv0 = (int64_t)(tn - ti.tile0) * 64;
                const int rem = (int)ti.m - 1 - (int)tin.sv0;
                tin.rem = rem > 63 ? 63 : rem;
            } else {
                tin = tile_info(tn, BATCHED ? __builtin_amdgcn_readfirstlane(a.tile_seg[tn]) : 0, std::false_type{});
            }
            load_tile(tin, nxt);  // prefetch the next tile
            load_err(tin, nxte);
        }
        if (BATCHED && ti.seg != cur_seg) {
            flush_minmax();
            cur_seg = ti.seg;
        }
        GQ_STAMP(0)
        // ---- prefilter: 16 (block, row block) chains; top-2 GROUP keys per (block, row-block half) ----
        // The MFMA of chain c+1 is issued in front of the key operations of chain c (sched_barrier pins the order):
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
#pragma unroll
        for (int c = 0; c < 2 * NRB; ++c) {
            const int rb = chain_rb(c), trk = chain_blk(c) * 2 + (rb >> 2);
            if constexpr (!SH::A_REGS) {
                if (chain_blk(c) == 0 && rb + 1 < NRB) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) af[(rb + 1) & 1][s] = frag_hi(rb + 1, s);
                }
            }
            if (c + 1 < 2 * NRB) {
                f32x16 nacc = {0};
                __builtin_amdgcn_sched_barrier(0);
                nacc = chain(c + 1, nacc);
                __builtin_amdgcn_sched_barrier(0);
                track_lo(trk, group_key(acc, rb, 0), group_key(acc, rb, 1));
                track_hi(trk, group_key(acc, rb, 2), group_key(acc, rb, 3));
                __builtin_amdgcn_sched_barrier(0);
                acc = nacc;
            } else {
                track_lo(trk, group_key(acc, rb, 0), group_key(acc, rb, 1));
                track_hi(trk, group_key(acc, rb, 2), group_key(acc, rb, 3));
            }
        }

        GQ_STAMP(1)
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
        float vf[D];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = cur[2 * s + (e >> 2)][e & 3];            // block 0: floats 16s+8h+e of subvector j
                float y = cur[2 * KS + 2 * s + (e >> 2)][e & 3];   // block 1
                swap32(x, y);                                      // x = floats 16s+e (0..7), y = floats 16s+8+e of subvector `lane`
                vf[16 * s + e] = x;
                if (!HALF) vf[(16 * s + 8 + e) % D] = y;           // (D = 8: the upper lanes' registers hold zeros, not data)
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
        // n2 of subvector `lane`: its two halves' sums (the same exchange)
        float n2 = n2p[0];
        {
            float n2o = n2p[1];
            swap32(n2, n2o);
            n2 += n2o;
        }
        const unsigned n2b = __float_as_uint(n2);

        GQ_STAMP(2)
        // ---- exact rescoring of the better of the two halves' best groups (4 codewords; the
        // reference's fmaf chain).  The other half's best group joins the bound on everything that
        // was not rescored.
        const bool pick1 = (bk[1] & KEY_MASK) > (bk[0] & KEY_MASK);
        const int kc = pick1 ? k1[1] : k1[0];
        const unsigned rest = max3u(s2[0], s2[1], (pick1 ? bk[0] : bk[1]) | 31u);
        const f32x4 p4 = exact_score_quad<D>(s_cb + (kc >> 2) * QS, vf);   // kc is a multiple of 4: one quad
        float val = p4[0];
        int idx = kc;
        take_if_greater(val, idx, p4[1], kc + 1);
        take_if_greater(val, idx, p4[2], kc + 2);
        take_if_greater(val, idx, p4[3], kc + 3);

        // The bound (header, 1), in the tile's scaled units.  The window test is on the BITS of n2 (a sum of
        // squares: never negative; NaN, infinity and zero fall outside): this file is compiled for NaN-free float compares.
        const float E = __builtin_amdgcn_sqrtf(n2) * err_rel + err_abs;
        const float others = __uint_as_float(rest);  // >= every s~ outside the rescored group
        const bool in_window = (n2b - N2_LO_BITS) <= (N2_HI_BITS - N2_LO_BITS);
        bool safe = in_window && (others + E < fabsf(val * sigma_t));
        // a NaN score: the float comparison above is compiled for NaN-free operands (-fno-honor-nans: the
        // complement `others + E >= |val|` is what is evaluated, false for NaN), so the bits decide
        if (nan_bits(val)) safe = false;
        // All-zero subvector (every score is +0 -> first index, u = +0): common in real gradients (dead units), so it is
        // settled here, not by the exact scan.  n2 == 0 (all sixteen f16 values zero or subnormal) is the cheap hint; the
        // sixteen f32 values decide.
        if (__ballot(n2b == 0u)) {
            unsigned any = 0;
#pragma unroll
            for (int e = 0; e < D; e += 2) any |= (__float_as_uint(vf[e]) | __float_as_uint(vf[e + 1])) & 0x7FFFFFFFu;
            if (any == 0u && !nan_bits(val)) {
                safe = true;
                val = 0.0f;
                idx = 0;
            }
        }

        const bool valid = lane <= ti.rem;         // this lane's subvector exists (the tensor's last tile may be short)

        // The next tile's scale: the larger n2 of two sampled subvectors of THIS tile goes to about
        // 2^SIGMA_TARGET_EXP2 (two samples, one of each half tile; scalar arithmetic on exponents; a sample that overflowed
        // pulls the scale down hard).
        float sigma_n = sigma_t;
        {
            const unsigned s0 = (unsigned)__builtin_amdgcn_readlane((int)n2b, 5), s1 = (unsigned)__builtin_amdgcn_readlane((int)n2b, 58);
            const unsigned smp = s0 > s1 ? s0 : s1;
            if (smp != 0u) {
                const int e2 = (int)(smp >> 23);
                const int k = e2 >= 255 ? -16 : ((127 + SIGMA_TARGET_EXP2 - e2) >> 1);
                int f = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sigma_t)) >> 23) + k;
                f = f < 27 ? 27 : (f > 227 ? 227 : f);
                sigma_n = __uint_as_float((unsigned)f << 23);
            }
        }

        GQ_STAMP(3)
        // The draw of the tile after next, in two steps with work between them (round 6, block M): the look at the counter here, the
        // add behind the conversion, the result read at the end of the tile.  As one call at the end (draw()) the wave waited out
        // two dependent LDS round trips per tile there: -0.6 % (profiles/r06_encode_ab.txt).  The look may be ~1,000 cycles old
        // when the add is decided: the PF_TAIL rule it serves is a heuristic.
        int dk_seen = 0x3FFFFFFF;
        if (lane == 0) dk_seen = __hip_atomic_load(&s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // Consume the prefetched tile (convert it to the next B fragments) BEFORE this tile's
        // stores are issued: the wait for the prefetch then sees only long-finished memory ops.
        // Done the other way round, the compiler's vmcnt wait at the first use of `nxt` sits right
        // behind the just-issued stores and every tile eats a store round trip.
        half8 nvh[2 * KS];
        float nn2p[2] = {0.0f, 0.0f};
        if (EF && tn < tile_end) fold_err(tin, nxt, nxte);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int s = 0; s < KS; ++s) scale8_f16(nxt[(blk * KS + s) * 2], nxt[(blk * KS + s) * 2 + 1], sigma_n, nvh[blk * KS + s], nn2p[blk]);
#pragma unroll
        for (int i = 0; i < NF; ++i) cur[i] = nxt[i];
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) vh[i] = nvh[i];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) n2p[blk] = nn2p[blk];
        __builtin_amdgcn_sched_barrier(0);

        GQ_STAMP(4)
        int dk_next = 0x3FFFFFFF;
        if (lane == 0 && dk_seen < tail_from) dk_next = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- the few subvectors the bound could not settle (header, 4): into this wave's ring, which goes through the
        // second pass when 32 are waiting and when the wave has run out of tiles.  A flagged lane writes its D floats and
        // where the answer goes.
        const bool flagged = valid && !safe;
        uint64_t todo = __ballot(flagged);
        // the one early flush of this wave's ring (see PF_FLUSH_AHEAD): the next tile lies near the end of the workgroup's run
        bool want_flush = PF_FLUSH_AHEAD > 0 && !flushed && (tn - lo_tile) >= run_len - PF_FLUSH_AHEAD;
        while (todo || want_flush) {   // (one trip; a second one only when a tile flags more lanes than the ring has room for)
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)todo, 0u));
            const bool mine = ((todo >> lane) & 1) && rank < QCAP - qcnt;
            if (mine) {
                const int slot = wave * QCAP + ((qhead + qcnt + rank) & (QCAP - 1));
                f32x4 *qv = reinterpret_cast<f32x4 *>(s_qv + D * slot);
#pragma unroll
                for (int q = 0; q < D / 4; ++q) qv[q] = f32x4{vf[4 * q], vf[4 * q + 1], vf[4 * q + 2], vf[4 * q + 3]};
                const uint64_t ca = (uint64_t)(uintptr_t)(ti.codes + ti.sv0) + (uint64_t)sizeof(CodeT) * (unsigned)lane;
                *reinterpret_cast<u32x4 *>(s_qm + 4 * slot) =
                    u32x4{(unsigned)ca, (unsigned)(ca >> 32), (unsigned)((BATCHED ? (int64_t)t * 64 : ti.sv0) + lane), (unsigned)ti.seg};
            }
            const uint64_t took = __ballot(mine);
            qcnt += (int)__builtin_popcountll(took);
            todo &= ~took;
            if (qcnt >= 32 || (want_flush && !todo)) {   // (a full ring is the exception: a wave meets ~1 unsettled subvector per tile)
                if (qcnt >= 32 || qcnt >= PF_FLUSH_MIN) {
                    const int n = qcnt < 32 ? qcnt : 32;
                    second_pass(qhead, n, wave, 0, 0);
                    qhead = (qhead + n) & (QCAP - 1);
                    qcnt -= n;
                    if (want_flush && !todo) flushed = true;
                }
                if (!todo) want_flush = false;   // (too few entries for a pass: asked again at the next tile)
            }
        }

        if (valid && safe) {   // uniform bases (the tile's first code / projection) + the lane index
            (ti.codes + ti.sv0)[(unsigned)lane] = (CodeT)idx;
            ((gf_ptr)u + (BATCHED ? (int64_t)t * 64 : ti.sv0))[(unsigned)lane] = val;
            lmin = fminf(lmin, val);
            lmax = fmaxf(lmax, val);
        }
        ti = tin;
        t = tn;
        tn = lo_tile + __builtin_amdgcn_readfirstlane(dk_next);
        sigma_t = sigma_n;
        GQ_STAMP(5)
        GQ_STAMPS_ONLY(++ntl;)
    }
    GQ_STAMPS_ONLY(const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();)
    if (BATCHED) flush_minmax();

    // out of tiles: what the early flush did not see -- a few entries: exact scans by the whole wave; more (no flush happened:
    // short runs, PF_FLUSH_AHEAD = 0): the second pass
    // PF_PAIR (D = 32 only: PfShape::PAIR): the two waves of a SIMD (w and w ^ 4) end their runs as a pair.  The first one to
    // get here publishes its ring and LEAVES -- its partner's last tiles then have the SIMD to themselves --, the second one
    // takes both rings through ONE second pass instead of two, the second of them contended.  For D = 16 / 8 this measured
    // SLOWER (a pair's ~23 entries exceed the 32 columns of a pass often enough that the launch, which ends with its last
    // wave, waits for a second pass somewhere).  The hand-over is an LDS counter: a wave's ring writes, its published (head, count) word and its add are DS operations of one wave
    // (in order); the wave whose add returns 1 reads the other's word and entries behind that return.
    int other_first = 0, other_n = 0;
    const int partner = wave ^ (PF_WAVES / 2);
    if (PF_PAIR) {
        int old = 0;
        if (lane == 0) {
            __hip_atomic_store(&s_ringpub[wave], (qhead << 8) | qcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __hip_atomic_fetch_add(&s_pairdone[wave & (PF_WAVES / 2 - 1)], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 0) {
            qcnt = 0;     // the partner takes this ring
        } else {
            int pub = 0;
            if (lane == 0) pub = __hip_atomic_load(&s_ringpub[partner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pub = __builtin_amdgcn_readfirstlane(pub);
            other_first = pub >> 8;
            other_n = pub & 255;
        }
    }
    while (qcnt + other_n) {
        if (qcnt + other_n <= PF_SCAN1_MAX) {   // (a few entries: exact scans by the whole wave are cheaper than a pass)
            if (qcnt) {
                scan1(wave * QCAP + qhead);
                qhead = (qhead + 1) & (QCAP - 1);
                qcnt -= 1;
            } else {
                scan1(partner * QCAP + other_first);
                other_first = (other_first + 1) & (QCAP - 1);
                other_n -= 1;
            }
        } else {
            const int na = qcnt < 32 ? qcnt : 32, nb = other_n < 32 - na ? other_n : 32 - na;
            second_pass(qhead, na, partner, other_first, nb);
            qhead = (qhead + na) & (QCAP - 1);
            qcnt -= na;
            other_first = (other_first + nb) & (QCAP - 1);
            other_n -= nb;
        }
    }
#ifdef GQ_PF_STAMPS
    if (lane == 0 && blockIdx.x < 256) {   // 12 words per wave behind the log (which this build does not write); batched form: M = ntiles * 64 (tools/stamp_batched.py)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(ws_worklist(ws) + (M - 65536)) + (blockIdx.x * 8 + wave) * 12;
        for (int i = 0; i < 6; ++i) o[i] = stamp_acc[i];
        o[6] = rt_entry, o[7] = rt0, o[8] = rt1, o[9] = ntl, o[10] = __builtin_amdgcn_s_memrealtime(), o[11] = 1000 * nscanned + npassed;
    }
#endif
    if (BATCHED) {   // the workgroup's table into the tensors' global words: one pair of atomics per tensor it met
        __syncthreads();
        if (threadIdx.x < PF_MM_SEGS) {
            const unsigned lo = s_mm[2 * threadIdx.x], hi = s_mm[2 * threadIdx.x + 1];
            if (lo != 0xFFFFFFFFu || hi != 0u) {
                // look before the atomic (as mm_fold does): every workgroup that met the tensor comes here at the end of the
                // launch, all within a microsecond or two, and the words only ever move towards the extremes -- a value
                // already as good as ours, however stale, makes ours redundant.  (Two 12 M-element tensors: 256 workgroups
                // on four words, 42.0 us against 39.3 for the same elements as 76 tensors.)
                unsigned *mm = a.seg_minmax + 2 * (seg_first + (int)threadIdx.x);
                if (lo < __hip_atomic_load(mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(mm, lo);
                if (hi > __hip_atomic_load(mm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mm + 1, hi);
            }
        }
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

template <typename CodeT, int D, int DR, int NRB>
static int launch_pf_rows(const float *grad, const float *codebook, int64_t M, int K, CodeT *codes, float *u, float *ws,
                          hipStream_t st, int profile_slot) {
    if (M > 0x7FFFFFFFLL) return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_encode: prefilter path needs M < 2^31");
    static const int bpc = resident_blocks_per_cu(hsq_encode_pf_kernel<CodeT, D, false, false, true, DR, NRB>, PF_THREADS, 0);
    PfArgs a = {};
    a.grad = grad;
    a.M = M;
    a.codes = codes;
    a.u = u;
    a.cb = codebook;
    a.ws = ws;
    a.K = K;
    const int64_t blocks = pf16_grid((M + 63) / 64, bpc);
    pf_split(a, (M + 63) / 64, blocks);
    hipEvent_t ev_start, ev_stop;
    if (profile_events(profile_slot, &ev_start, &ev_stop)) {   // events attached to this dispatch (gq_profile_read)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<CodeT, D, false, false, true, DR, NRB>), dim3((unsigned)blocks),
                              dim3(PF_THREADS), 0, st, ev_start, ev_stop, 0, a);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<CodeT, D, false, false, true, DR, NRB>), dim3((unsigned)blocks),
                           dim3(PF_THREADS), 0, st, a);
    }
    GQ_CHECK_LAUNCH("gq_hsq_encode (prefilter)");
    return GQ_OK;
}

// K <= 32 / <= 64 (byte codes): the kernels that score one / two row blocks; everything else up to 256 rows: all eight
template <typename CodeT, int D, int DR = D>
static int launch_pf(const float *grad, const float *codebook, int64_t M, int K, CodeT *codes, float *u, float *ws,
                     hipStream_t st, int profile_slot) {
    if constexpr (sizeof(CodeT) == 1) {
        if (K <= 32) return launch_pf_rows<CodeT, D, DR, 1>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
        if (K <= 64) return launch_pf_rows<CodeT, D, DR, 2>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
    }
    return launch_pf_rows<CodeT, D, DR, 8>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
}

// d = 8, 16 or 32, K <= 256 (a multiple of 4); d = 12 / 24 (the reference's repaired dimensions) as the D = 16 / 32 kernels over rows of 12 / 24 floats
template <typename CodeT>
int launch_encode_pf(const float *grad, const float *codebook, int64_t M, int d, int K, CodeT *codes, float *u, float *ws,
                     hipStream_t st, int profile_slot) {
    if (K < 4 || K > 256 || (K & 3)) return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: the prefilter kernel was asked for K = %d", K);
    if (d == 16) return launch_pf<CodeT, 16>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
    if (d == 32) return launch_pf<CodeT, 32>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
    if (d == 8) return launch_pf<CodeT, 8>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
    if (d == 12) return launch_pf<CodeT, 16, 12>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
    if (d == 24) return launch_pf<CodeT, 32, 24>(grad, codebook, M, K, codes, u, ws, st, profile_slot);
    return fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode: the prefilter kernel was asked for d = %d", d);
}

template int launch_encode_pf<uint8_t>(const float *, const float *, int64_t, int, int, uint8_t *, float *, float *, hipStream_t, int);
template int launch_encode_pf<int32_t>(const float *, const float *, int64_t, int, int, int32_t *, float *, float *, hipStream_t, int);

}  // namespace gq

namespace gq {
template <int D, bool EF, int DR, int NRB>
static int encode_batched_rows(const char *what, const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                               const float *codebook, int K, uint8_t *wire, float *u_flat, uint32_t *seg_minmax, float *workspace,
                               float ef_scale, int profile_slot, void *stream) {
    if (nseg < 1 || ntiles < 1 || ntiles * 64 > 0x7FFFFFFFLL)
        return fail(GQ_ERR_INVALID_ARG, "%s: bad sizes nseg=%d ntiles=%lld", what, nseg, (long long)ntiles);
    if (!seg_table || !tile_seg || !codebook || !wire || !u_flat || !seg_minmax || !workspace)
        return fail(GQ_ERR_INVALID_ARG, "%s: null pointer", what);
    static const int bpc = resident_blocks_per_cu(hsq_encode_pf_kernel<uint8_t, D, true, EF, true, DR, NRB>, PF_THREADS, 0);
    PfArgs a = {};
    a.K = K;
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
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<uint8_t, D, true, EF, true, DR, NRB>), dim3((unsigned)blocks),
                              dim3(PF_THREADS), 0, st, ev_start, ev_stop, 0, a);
    } else if (nseg <= PF_LDS_SEGS) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<uint8_t, D, true, EF, true, DR, NRB>), dim3((unsigned)blocks),
                           dim3(PF_THREADS), 0, st, a);
    } else {   // (longer lists: the records from global memory; built for all eight row blocks only)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_encode_pf_kernel<uint8_t, D, true, EF, false, DR, 8>), dim3((unsigned)blocks),
                           dim3(PF_THREADS), 0, st, a);
    }
    GQ_CHECK_LAUNCH(what);
    return GQ_OK;
}

template <int D, bool EF, int DR = D>
static int encode_batched(const char *what, const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                          const float *codebook, int K, uint8_t *wire, float *u_flat, uint32_t *seg_minmax, float *workspace,
                          float ef_scale, int profile_slot, void *stream) {
    if (K <= 32 && nseg <= PF_LDS_SEGS)
        return encode_batched_rows<D, EF, DR, 1>(what, seg_table, tile_seg, nseg, ntiles, codebook, K, wire, u_flat, seg_minmax, workspace, ef_scale, profile_slot, stream);
    if (K <= 64 && nseg <= PF_LDS_SEGS)
        return encode_batched_rows<D, EF, DR, 2>(what, seg_table, tile_seg, nseg, ntiles, codebook, K, wire, u_flat, seg_minmax, workspace, ef_scale, profile_slot, stream);
    return encode_batched_rows<D, EF, DR, 8>(what, seg_table, tile_seg, nseg, ntiles, codebook, K, wire, u_flat, seg_minmax, workspace, ef_scale, profile_slot, stream);
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

// K <= 256 (a multiple of 4), d = 8 / 12 / 16 / 24 / 32, byte codes: the multi-tensor prefilter launch (gq_hsq_encode_batched)
GQ_INTERNAL int gqi_hsq_encode_batched_pf(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                          const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire, float *u_flat,
                                          uint32_t *seg_minmax, float *workspace, int profile_slot, void *stream) {
    const char *what = "gq_hsq_encode_batched";
    if (K < 4 || K > 256 || (K & 3)) return gq::fail(GQ_ERR_UNSUPPORTED, "%s: the prefilter launch takes K <= 256, a multiple of 4 (K = %d)", what, K);
#define GQ_PF_BATCHED(DD, DP)                                                                                             \
    if (d == DD)                                                                                                          \
        return ef ? gq::encode_batched<DP, true, DD>(what, seg_table, tile_seg, nseg, ntiles, codebook, K, wire, u_flat, seg_minmax, \
                                                     workspace, ef_scale, profile_slot, stream)                           \
                  : gq::encode_batched<DP, false, DD>(what, seg_table, tile_seg, nseg, ntiles, codebook, K, wire, u_flat, seg_minmax, \
                                                      workspace, 0.0f, profile_slot, stream);
    GQ_PF_BATCHED(16, 16)
    GQ_PF_BATCHED(32, 32)
    GQ_PF_BATCHED(8, 8)
    GQ_PF_BATCHED(12, 16)
    GQ_PF_BATCHED(24, 32)
#undef GQ_PF_BATCHED
    return gq::fail(GQ_ERR_UNSUPPORTED, "%s: d must be 8, 12, 16, 24 or 32 (K <= 256)", what);
}
