// HSQ decode + parameter-server aggregate for gfx950.
//
// Replaces, for each of R (codes, levels, lb, ub) payloads,
// probabilistic_scalar_compressor.py:29-33 (level -> norm) and
// nearest_neighbor_compressor.py:80-90 (codebook gather * norm), then
// ps_quantizer.py:48 (stack().mean(0)): payloads are summed in ascending rank order
// and divided by R, so the fp32 result equals the reference's for the same payloads.
// Write-bound: 4 B per output element + 2R/d B of payload reads; the codebook is
// staged in LDS once per workgroup.  -ffp-contract=off keeps mul / div / add unfused.
#include "gq_common.hpp"
#include "hsq_encode_common.hpp"
#include "hsq_levels_common.hpp"
#include <type_traits>

namespace gq {

constexpr int DEC_THREADS = 256;

// LDS row stride (floats) of a staged codebook.  Rows of d floats laid end to end start on very few
// bank positions (d = 16: 64 B rows, the 128 B bank window has TWO), and a gather of 16 random rows
// per ds_read_b128 then serialises ~8 ways (PMC: 16 conflict cycles per LDS instruction, LDS stalled 61 %
// of an R = 8 decode).  An odd number of 16-byte units per row spreads the row starts over all positions.
__host__ __device__ inline int cb_row_stride(int d) { return ((d >> 2) & 1) ? d : d + 4; }

// s = 2^n_bit, so the reference's division by s is an exact scaling: multiplying by inv_s = 2^-n_bit
// gives the same bits for every input and saves the IEEE division sequence per payload.
template <typename LevelT>
__device__ __forceinline__ float level_to_norm(LevelT l, float lb, float range, float inv_s) {
    float t = (float)l * range;
    t = t * inv_s;
    return t + lb;
}
template <>
__device__ __forceinline__ float level_to_norm<float>(float l, float, float, float) {
    return l;  // n_bit == 32: the payload already carries the f32 projection
}

// d % 4 == 0: one thread produces 4 consecutive floats (one dwordx4 store).
template <typename CodeT, typename LevelT, bool LDS_CB>
__global__ __launch_bounds__(DEC_THREADS) void hsq_decode_sum_v4_kernel(
    const CodeT *__restrict__ codes, const LevelT *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int R, int64_t M,
    int d, int K, int n_bit, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_cb[];
    if (LDS_CB) {
        const int rs = cb_row_stride(d);
        for (int i = threadIdx.x; i < K * d; i += DEC_THREADS) s_cb[(i / d) * rs + (i % d)] = cb[i];
        __syncthreads();
    }
    const int q_per = d >> 2;
    const int64_t total = M * q_per;
    const float s = 1.0f / (float)(1 << (n_bit & 31));   // inv_s, see level_to_norm
    const MeanDiv md = mean_div_of(R);
    const int64_t stride = (int64_t)gridDim.x * DEC_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * DEC_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / q_per;
        const int q = (int)(i - m * q_per);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int r = 0; r < R; ++r) {
            const int code = (int)codes[(int64_t)r * code_stride + m];
            const float lb = lb_ub ? lb_ub[r * lbub_stride] : 0.0f;
            const float range = lb_ub ? (lb_ub[r * lbub_stride + 1] - lb) : 0.0f;
            const float n = level_to_norm<LevelT>(levels[(int64_t)r * level_stride + m], lb, range, s);
            const float *row = LDS_CB ? s_cb + code * cb_row_stride(d) + 4 * q : cb + (int64_t)code * d + 4 * q;
            const f32x4 c = *reinterpret_cast<const f32x4 *>(row);
            f32x4 dec;
            dec[0] = c[0] * n;
            dec[1] = c[1] * n;
            dec[2] = c[2] * n;
            dec[3] = c[3] * n;
            if (r == 0)
                acc = dec;
            else {
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
        *reinterpret_cast<f32x4 *>(out + 4 * i) = acc;
    }
}

// d = 16 with byte codes and byte levels (the BASELINE wire).  With R payloads the decode is a gather
// of R codebook rows per subvector, and from a plain LDS image that gather runs at a sixth of the LDS
// rate (PMC: 16 conflict cycles per ds_read_b128; R = 8 took 69-80 us for 100 MB).  Here
//  * one thread produces the same quarter (4 floats) of FOUR consecutive subvectors: two dword loads
//    per payload bring its 4 codes and 4 levels;
//  * the codebook is staged FOUR times, row r copy c at byte r*256 + c*64.  A ds_read_b128 is served in
//    four fixed groups of 16 lanes (MI355X_MICROARCH.md, LDS): the four 4-lane teams of a group
//    (one subvector each, 64 contiguous bytes) read copies 0..3, so every group covers the 64 banks
//    exactly once whatever the codes are: conflict-free, 256 B/clk;
//  * packed f32 multiplies / adds (separate roundings, as the reference).
// Same arithmetic and summation order as the kernels above.
constexpr int DEC16_THREADS = 1024;
constexpr int DEC16_LBUB = 64;
#ifndef GQ_DEC16_CHUNK
#define GQ_DEC16_CHUNK 3
#endif
#ifndef GQ_DEC16_WAVES
#define GQ_DEC16_WAVES 8
#endif
constexpr int DEC16_CHUNK = GQ_DEC16_CHUNK;   // payloads whose words are requested together

// LDS byte address of a codebook row for this lane: code * 256 + (copy * 64 + quarter * 16), built by ONE v_perm_b32
// from byte k of the packed codes and the lane's constant (< 256): [0, 0, code_k, lane_const].
template <int K4>
__device__ __forceinline__ unsigned row_addr(unsigned c4, unsigned lane_const) {
    return __builtin_amdgcn_perm(c4, lane_const, 0x0c0c0000u | ((4u + K4) << 8));
}

// One payload's contribution to the four subvectors of a team (see the kernel): norms by lane q, shared through
// quad-permute DPP moves; probabilistic_scalar_compressor.py:31-32 unfused, nearest_neighbor_compressor.py:88.
// ABS0: the codebook image starts at LDS address 0 (a kernel whose only LDS is its dynamic array) and the v_perm_b32
// result IS the address; through a pointer the compiler adds the array's link-time base (0) to every row address.
typedef const f32x4 __attribute__((address_space(3))) lds_f32x4;
// FMA (opt-in, GQ_AGGREGATE_FMA in n_bit; payloads after the first): acc = fma(c, n, acc) instead of the reference's
// separately rounded product and sum -- half the operations per payload, within 1e-6 relative L2 of the exact mean
// (north_star grants 1e-5 on the decoded aggregate); never used for a plain decompress, R = 1 or error-feedback round trips.
template <bool FIRST, bool PACKED6, bool ABS0 = false, bool FMA = false>
__device__ __forceinline__ void dec16_payload(f32x4 (&acc)[4], unsigned c4, unsigned l4, float lb, float range, float inv_s,
                                              int q, const char *cb_bytes, unsigned lane_const) {
    const float n_own = level_to_norm<unsigned>(PACKED6 ? ((l4 >> (6 * q)) & 63u) : ((l4 >> (8 * q)) & 255u), lb, range, inv_s);
    const int n_bits = __builtin_bit_cast(int, n_own);
    const float n_team[4] = {   // quad_perm [k,k,k,k]: lane k of the team broadcasts
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0x00, 0xF, 0xF, true)),
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0x55, 0xF, 0xF, true)),
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0xAA, 0xF, 0xF, true)),
        __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(n_bits, 0xFF, 0xF, 0xF, true))};
    const unsigned a[4] = {row_addr<0>(c4, lane_const), row_addr<1>(c4, lane_const), row_addr<2>(c4, lane_const),
                           row_addr<3>(c4, lane_const)};
    if constexpr (FMA && !FIRST) {      // one v_fmac_f32_dpp per element: the team's norms are read across the quad by the multiply-add itself
        const float n_rdy = quad_norm_ready(n_own);
        const f32x4 c0 = ABS0 ? *reinterpret_cast<lds_f32x4 *>((uintptr_t)a[0]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[0]);
        const f32x4 c1 = ABS0 ? *reinterpret_cast<lds_f32x4 *>((uintptr_t)a[1]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[1]);
        const f32x4 c2 = ABS0 ? *reinterpret_cast<lds_f32x4 *>((uintptr_t)a[2]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[2]);
        const f32x4 c3 = ABS0 ? *reinterpret_cast<lds_f32x4 *>((uintptr_t)a[3]) : *reinterpret_cast<const f32x4 *>(cb_bytes + a[3]);
        acc[0] = fmac_quad4<0>(acc[0], n_rdy, c0);
        acc[1] = fmac_quad4<1>(acc[1], n_rdy, c1);
        acc[2] = fmac_quad4<2>(acc[2], n_rdy, c2);
        acc[3] = fmac_quad4<3>(acc[3], n_rdy, c3);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float n = n_team[k];
        const f32x4 c = ABS0 ? *reinterpret_cast<lds_f32x4 *>((uintptr_t)a[k])
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

// Payloads are taken DEC16_CHUNK at a time: the 2 x DEC16_CHUNK dwords of a chunk are requested back to back and
// only then consumed.  (Round 2 fetched a payload's two words inside the payload loop: every payload waited out its
// own round trip to HBM, R of them in a row per item, and the kernel's time was write time PLUS R x 2.1 us --
// latency, not arithmetic.)
template <bool PACKED6>   // levels: a byte each, or four 6-bit values per three bytes (GQ_LEVELS_PACKED6; level_stride in bytes either way)
__global__ __launch_bounds__(DEC16_THREADS) __attribute__((amdgpu_waves_per_eu(GQ_DEC16_WAVES, GQ_DEC16_WAVES)))
void hsq_decode_sum_d16u8_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int R, int64_t M,
    int K, int n_bit, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_cb[];   // [K][4 copies][16]
    for (int i = threadIdx.x; i < K * 16; i += DEC16_THREADS) {   // (row, copy, quarter)
        const int row = i >> 4, c = (i >> 2) & 3, q = i & 3;
        *reinterpret_cast<f32x4 *>(s_cb + row * 64 + c * 16 + 4 * q) = *reinterpret_cast<const f32x4 *>(cb + row * 16 + 4 * q);
    }
    // (lb, ub - lb) of the first DEC16_LBUB payloads once per workgroup: read per payload as a uniform
    // scalar load, every s_waitcnt lgkmcnt for it also drained the LDS gathers in flight
    __shared__ float2 s_lbub[DEC16_LBUB];
    for (int r = threadIdx.x; r < R && r < DEC16_LBUB; r += DEC16_THREADS) {
        const float lb = lb_ub[r * lbub_stride];
        s_lbub[r] = make_float2(lb, lb_ub[r * lbub_stride + 1] - lb);
    }
    __syncthreads();
    const float inv_s = 1.0f / (float)(1 << (n_bit & 31));
    const MeanDiv md = mean_div_of(R);
    const int q = threadIdx.x & 3;
    const unsigned lane_const = (unsigned)(((threadIdx.x >> 3) & 3) * 64 + 16 * q);   // this lane's copy and quarter, bytes
    const char *const cb_bytes = reinterpret_cast<const char *>(s_cb);
    auto lbub_of = [&](int r, float &lb, float &range) {
        if (r < DEC16_LBUB) {
            const float2 lr = s_lbub[r];
            lb = lr.x;
            range = lr.y;
        } else {
            lb = lb_ub[r * lbub_stride];
            range = lb_ub[r * lbub_stride + 1] - lb;
        }
    };
    const int64_t total = ((M + 3) >> 2) * 4;   // (group of 4 subvectors, quarter) items
    const int64_t stride = (int64_t)gridDim.x * DEC16_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * DEC16_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t m0 = (i >> 2) * 4;
        const int nv = (M - m0) < 4 ? (int)(M - m0) : 4;
        f32x4 acc[4];
        if (nv == 4) {
            const unsigned off = (unsigned)m0;      // M < 2^31 subvectors (checked by the launcher): a 32-bit lane offset
            const unsigned loff = PACKED6 ? 3u * (unsigned)(m0 >> 2) : off;
            const uint8_t *cp = codes, *lp = levels;   // uniform: advance on the scalar unit
            for (int r0 = 0; r0 < R; r0 += DEC16_CHUNK) {
                unsigned c4[DEC16_CHUNK], l4[DEC16_CHUNK];
#pragma unroll
                for (int jj = 0; jj < DEC16_CHUNK; ++jj) {
                    if (r0 + jj < R) {
                        c4[jj] = PACKED6 ? load_packed6(cp + (int64_t)jj * code_stride + off)      // (an alignment-free dword load)
                                         : *reinterpret_cast<const unsigned *>(cp + (int64_t)jj * code_stride + off);
                        l4[jj] = PACKED6 ? load_packed6(lp + (int64_t)jj * level_stride + loff)
                                         : *reinterpret_cast<const unsigned *>(lp + (int64_t)jj * level_stride + off);
                    }
                }
                cp += (int64_t)DEC16_CHUNK * code_stride;
                lp += (int64_t)DEC16_CHUNK * level_stride;
#pragma unroll
                for (int jj = 0; jj < DEC16_CHUNK; ++jj) {
                    if (r0 + jj < R) {
                        float lb, range;
                        lbub_of(r0 + jj, lb, range);
                        if (jj == 0 && r0 == 0)
                            dec16_payload<true, PACKED6>(acc, c4[jj], l4[jj], lb, range, inv_s, q, cb_bytes, lane_const);
                        else
                            dec16_payload<false, PACKED6>(acc, c4[jj], l4[jj], lb, range, inv_s, q, cb_bytes, lane_const);
                    }
                }
            }
        } else {   // the last, partial group of a tensor whose M is not a multiple of 4: byte loads
            for (int r = 0; r < R; ++r) {
                const uint8_t *cp = codes + (int64_t)r * code_stride + m0;
                const uint8_t *lp = levels + (int64_t)r * level_stride + (PACKED6 ? 3 * (m0 >> 2) : m0);
                unsigned c4 = 0, l4 = 0;
                for (int k = 0; k < nv; ++k) c4 |= (unsigned)cp[k] << (8 * k);
                if (PACKED6) {   // a group is always stored whole (slots past M hold 0)
                    l4 = (unsigned)lp[0] | ((unsigned)lp[1] << 8) | ((unsigned)lp[2] << 16);
                } else {
                    for (int k = 0; k < nv; ++k) l4 |= (unsigned)lp[k] << (8 * k);
                }
                float lb, range;
                lbub_of(r, lb, range);
                if (r == 0)
                    dec16_payload<true, PACKED6>(acc, c4, l4, lb, range, inv_s, q, cb_bytes, lane_const);
                else
                    dec16_payload<false, PACKED6>(acc, c4, l4, lb, range, inv_s, q, cb_bytes, lane_const);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < nv) {
                f32x4 a = acc[k];
                if (md.apply) {
                    a[0] = mean_div(a[0], md);
                    a[1] = mean_div(a[1], md);
                    a[2] = mean_div(a[2], md);
                    a[3] = mean_div(a[3], md);
                }
                *reinterpret_cast<f32x4 *>(out + (m0 + k) * 16 + 4 * q) = a;
            }
        }
    }
}

// The same decode for a compile-time payload count R <= DEC16_RMAX, software-pipelined ACROSS items.  In the kernel above
// every wave runs load -> R payloads of arithmetic -> 4 stores, and all waves of the chip do so in step (each has ~3 items):
// the payload words of an item are requested behind the stores of the previous one in the CU's memory pipeline and come
// back once that burst has drained, HBM writes pause while everybody computes -- T(R) = write time + R x compute time.
// Here the R (code, level) word pairs of an item live in registers and each pair is re-requested for the wave's NEXT item
// as soon as it has been consumed, i.e. before the current item's stores are queued; the loop body is straight-line for
// the compiler's vmcnt bookkeeping (a payload's words wait with R + 3 younger operations outstanding, the stores among
// them), so the stores drain under the next item's arithmetic.  (lb, ub - lb) sit in scalar registers.
constexpr int DEC16_RMAX = 16;
#ifndef GQ_DEC16R_WAVES
#define GQ_DEC16R_WAVES 8
#endif

template <int R, bool PACKED6, bool FMA = false>
__global__ __launch_bounds__(DEC16_THREADS) __attribute__((amdgpu_waves_per_eu(GQ_DEC16R_WAVES, GQ_DEC16R_WAVES)))
void hsq_decode_sum_d16u8_r_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int64_t M, int K,
    int n_bit, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_cb[];   // [K][4 copies][16] at LDS address 0: a row address is the v_perm_b32 result itself
    // Everything the workgroup needs before its first sum is requested up front, the first item's payload words first:
    // their trip to HBM then runs under the staging of the codebook image.
    const unsigned full = (unsigned)(M >> 2) * 4u;   // items of whole groups; M < 2^31 - 2^21 (launcher): 32-bit item numbers
    const unsigned stride = gridDim.x * DEC16_THREADS;
    unsigned i = blockIdx.x * DEC16_THREADS + threadIdx.x;
    // The four lanes of a team need the same word pair of every payload.  Each lane fetches the pair of ONE payload in four
    // -- lane q of a team: payloads q, 4 + q, 8 + q, ... -- and a payload's pair reaches the team by a quad-permute DPP move
    // when its turn comes: ceil(R / 4) x 2 wave-wide loads per item instead of 2 R (a wave-wide dword load keeps the CU's
    // address path busy for ~16 cycles whatever it fetches: at R = 16 that path, not the arithmetic, set the time) and as
    // many word registers, so that R = 16 fits the 64 registers of 8 waves per SIMD.
    constexpr int NW = (R + 3) / 4;           // word pairs per lane
    constexpr int LASTN = R - 4 * (NW - 1);   // payloads in the last group (1..4)
    const int q = threadIdx.x & 3;
    unsigned cw[NW], lw[NW];
    // group bases pinned in scalar registers (+ 4 j payload strides), the lane's own payload inside the group goes into the
    // 32-bit offset: q strides (launcher: 3 strides + the section fit 32 bits); lanes past R in the last group repeat payload R - 1
    typedef const uint8_t __attribute__((address_space(1))) gbyte;   // global address space kept through the integer round trip
    gbyte *cbase[NW], *lbase[NW];
    auto pin = [](const uint8_t *p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<gbyte *>(((uint64_t)hi << 32) | lo);
    };
    typedef unsigned __attribute__((aligned(1))) word_any;   // the packed form reads words at any byte (see load_packed6)
    auto ldw = [](gbyte *p) {
        return PACKED6 ? (unsigned)*reinterpret_cast<const word_any __attribute__((address_space(1))) *>(p)
                       : *reinterpret_cast<const unsigned __attribute__((address_space(1))) *>(p);
    };
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        cbase[j] = pin(codes + (int64_t)(4 * j) * code_stride);
        lbase[j] = pin(levels + (int64_t)(4 * j) * level_stride);
    }
    const unsigned qc = (unsigned)q * (unsigned)code_stride, ql = (unsigned)q * (unsigned)level_stride;
    const int ql_last = q < LASTN ? q : LASTN - 1;
    const unsigned qc_last = (unsigned)ql_last * (unsigned)code_stride, qll_last = (unsigned)ql_last * (unsigned)level_stride;
    auto request_group = [&](unsigned item, int j) {
        const unsigned off = item & ~3u;                           // first subvector of the group of four
        const unsigned loff = PACKED6 ? 3u * (item >> 2) : off;
        const bool last = LASTN != 4 && j == NW - 1;
        cw[j] = ldw(cbase[j] + (off + (last ? qc_last : qc)));
        lw[j] = ldw(lbase[j] + (loff + (last ? qll_last : ql)));
    };
    if (i < full) {
#pragma unroll
        for (int j = 0; j < NW; ++j) request_group(i, j);
    }
    constexpr int STAGE = 256 * 16 / DEC16_THREADS;   // K <= 256 rows of four 16-byte quarters
    f32x4 stage[STAGE];
#pragma unroll
    for (int n = 0; n < STAGE; ++n) {
        const int e = threadIdx.x + n * DEC16_THREADS;   // (row, copy, quarter)
        if (e < K * 16) stage[n] = *reinterpret_cast<const f32x4 *>(cb + (e >> 4) * 16 + 4 * (e & 3));
    }
    float lb[R], range[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float l = lb_ub[r * lbub_stride], u = lb_ub[r * lbub_stride + 1];
        lb[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, l)));
        range[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u - l)));
    }
#pragma unroll
    for (int n = 0; n < STAGE; ++n) {
        const int e = threadIdx.x + n * DEC16_THREADS;
        const int row = e >> 4, c = (e >> 2) & 3, q = e & 3;
        if (e < K * 16) *reinterpret_cast<f32x4 *>(s_cb + row * 64 + c * 16 + 4 * q) = stage[n];
    }
    __syncthreads();
    const float inv_s = 1.0f / (float)(1 << (n_bit & 31));
    const MeanDiv md = mean_div_of(R);
    const unsigned lane_const = (unsigned)(((threadIdx.x >> 3) & 3) * 64 + 16 * q);
    const char *const cb_bytes = reinterpret_cast<const char *>(s_cb);
    // the last, partial group of a tensor whose M is not a multiple of 4 (byte loads), by the first team of workgroup 0 --
    // BEFORE the pipelined loop, so that nothing of it stays in registers across the loop
    const int nv = (int)(M & 3);
    if (nv != 0 && blockIdx.x == 0 && threadIdx.x < 4) {
        const int64_t m0 = (int64_t)full;
        f32x4 acc[4];
#pragma unroll 1
        for (int r = 0; r < R; ++r) {   // a plain loop: (lb, ub) from memory, nothing of the main loop's register set kept
            const uint8_t *cp = codes + (int64_t)r * code_stride + m0;
            const uint8_t *lp = levels + (int64_t)r * level_stride + (PACKED6 ? 3 * (m0 >> 2) : m0);
            unsigned c = 0, l = 0;
            for (int k = 0; k < nv; ++k) c |= (unsigned)cp[k] << (8 * k);
            if (PACKED6) {   // a group is always stored whole (slots past M hold 0)
                l = (unsigned)lp[0] | ((unsigned)lp[1] << 8) | ((unsigned)lp[2] << 16);
            } else {
                for (int k = 0; k < nv; ++k) l |= (unsigned)lp[k] << (8 * k);
            }
            const float tl = lb_ub[r * lbub_stride], tr = lb_ub[r * lbub_stride + 1] - tl;
            if (r == 0)
                dec16_payload<true, PACKED6, true>(acc, c, l, tl, tr, inv_s, q, cb_bytes, lane_const);
            else
                dec16_payload<false, PACKED6, true, FMA>(acc, c, l, tl, tr, inv_s, q, cb_bytes, lane_const);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < nv) {
                f32x4 a = acc[k];
                if (md.apply) {
                    a[0] = mean_div(a[0], md);
                    a[1] = mean_div(a[1], md);
                    a[2] = mean_div(a[2], md);
                    a[3] = mean_div(a[3], md);
                }
                *reinterpret_cast<f32x4 *>(out + (m0 + k) * 16 + 4 * q) = a;
            }
        }
    }
#pragma clang loop unroll(disable)
    while (i < full) {
        const unsigned nxt = i + stride;
        const unsigned pre = nxt < full ? nxt : i;   // the last item re-requests itself: no branch around the loads
        f32x4 acc[4];
        float hold[4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = r >> 2;
            unsigned c4 = team_word(cw[j], r & 3), l4 = team_word(lw[j], r & 3);
            // R = 3: left alone, the compiler forms all 48 products before the first sum, which with the odd divisor's
            // quotients no longer fits the 64 registers of 8 waves: the third payload's words wait for the first's products
            if (R == 3 && r == 2) asm("" : "+v"(c4), "+v"(l4) : "v"(hold[0]), "v"(hold[1]), "v"(hold[2]), "v"(hold[3]));
            if (r == 0)
                dec16_payload<true, PACKED6, true>(acc, c4, l4, lb[r], range[r], inv_s, q, cb_bytes, lane_const);
            else
                dec16_payload<false, PACKED6, true, FMA>(acc, c4, l4, lb[r], range[r], inv_s, q, cb_bytes, lane_const);
            if (R == 3 && r == 0) {
                hold[0] = acc[0][0];
                hold[1] = acc[1][1];
                hold[2] = acc[2][2];
                hold[3] = acc[3][3];
            }
            if ((r & 3) == 3 || r == R - 1) request_group(pre, j);   // the group's four payloads are summed: its registers take the next item's
        }
        float *o = out + (int64_t)(i & ~3u) * 16 + 4 * q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 a = acc[k];
            if (md.apply) {
                a[0] = mean_div(a[0], md);
                a[1] = mean_div(a[1], md);
                a[2] = mean_div(a[2], md);
                a[3] = mean_div(a[3], md);
            }
            *reinterpret_cast<f32x4 *>(o + 16 * k) = a;
            // an odd R's quotients (four dependent operations each) one subvector at a time: interleaved over all sixteen
            // elements they need 3 x 16 registers at once, which R = 3 does not have at 8 waves
        }
        i = nxt;
    }
}

// Any R above DEC16_RMAX with the same pipelining, in chunks of DEC16_RMAX payloads: the words of (item, chunk) sit in
// registers, each pair is re-requested for the NEXT chunk -- of the same item, or the first chunk of the wave's next item --
// as soon as it has been consumed.  Every trip of the chunk loop issues its 2 x DEC16_RMAX loads unconditionally (a chunk
// with fewer payloads re-reads its last one), so the compiler's vmcnt bookkeeping stays exact and no wait lands on the
// previous item's stores.  The sums start from +0: only reached with R > 1, where the mean adds +0 anyway (mean_div).
template <bool PACKED6>
__global__ __launch_bounds__(DEC16_THREADS) __attribute__((amdgpu_waves_per_eu(GQ_DEC16R_WAVES, GQ_DEC16R_WAVES)))
void hsq_decode_sum_d16u8_rc_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int R, int64_t M, int K,
    int n_bit, float *__restrict__ out) {
    constexpr int C = 8;   // payloads per chunk
    extern __shared__ __attribute__((aligned(16))) float s_cb[];   // [K][4 copies][16] at LDS address 0, then (lb, ub - lb) of the R payloads
    for (int i = threadIdx.x; i < K * 16; i += DEC16_THREADS) {   // (row, copy, quarter)
        const int row = i >> 4, c = (i >> 2) & 3, q = i & 3;
        *reinterpret_cast<f32x4 *>(s_cb + row * 64 + c * 16 + 4 * q) = *reinterpret_cast<const f32x4 *>(cb + row * 16 + 4 * q);
    }
    float2 *const s_lbub = reinterpret_cast<float2 *>(s_cb + K * 64);
    for (int r = threadIdx.x; r < R; r += DEC16_THREADS) {
        const float lb = lb_ub[r * lbub_stride];
        s_lbub[r] = make_float2(lb, lb_ub[r * lbub_stride + 1] - lb);
    }
    __syncthreads();
    const float inv_s = 1.0f / (float)(1 << (n_bit & 31));
    const MeanDiv md = mean_div_of(R);
    const int q = threadIdx.x & 3;
    const unsigned lane_const = (unsigned)(((threadIdx.x >> 3) & 3) * 64 + 16 * q);
    const char *const cb_bytes = reinterpret_cast<const char *>(s_cb);
    const unsigned full = (unsigned)(M >> 2) * 4u;
    const unsigned stride = gridDim.x * DEC16_THREADS;
    const int nchunks = (R + C - 1) / C;
    typedef const uint8_t __attribute__((address_space(1))) gbyte;
    typedef unsigned __attribute__((aligned(1))) word_any;
    auto ldw = [](gbyte *p) {
        return PACKED6 ? (unsigned)*reinterpret_cast<const word_any __attribute__((address_space(1))) *>(p)
                       : *reinterpret_cast<const unsigned __attribute__((address_space(1))) *>(p);
    };
    const uint64_t codes0 = reinterpret_cast<uint64_t>(codes), levels0 = reinterpret_cast<uint64_t>(levels);
    unsigned c4[C], l4[C];
    // requests for slot jj of chunk `chunk` of item `item` (payload min(chunk * C + jj, R - 1))
    auto request = [&](unsigned item, int chunk, int jj) {
        const unsigned off = item & ~3u, loff = PACKED6 ? 3u * (item >> 2) : off;
        int r = chunk * C + jj;
        r = r < R ? r : R - 1;
        c4[jj] = ldw(reinterpret_cast<gbyte *>(codes0 + (uint64_t)r * (uint64_t)code_stride) + off);
        l4[jj] = ldw(reinterpret_cast<gbyte *>(levels0 + (uint64_t)r * (uint64_t)level_stride) + loff);
    };
    unsigned i = blockIdx.x * DEC16_THREADS + threadIdx.x;
    if (i < full) {
#pragma unroll
        for (int jj = 0; jj < C; ++jj) request(i, 0, jj);
    }
    while (i < full) {
        const unsigned nxt = i + stride;
        const unsigned pre = nxt < full ? nxt : i;
        f32x4 acc[4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool last = chunk + 1 == nchunks;
            const unsigned ni = last ? pre : i;
            const int nc = last ? 0 : chunk + 1;
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                const int r = chunk * C + jj;
                if (r < R) {
                    const float2 lr = s_lbub[r];
                    dec16_payload<false, PACKED6, true>(acc, c4[jj], l4[jj], lr.x, lr.y, inv_s, q, cb_bytes, lane_const);
                }
                request(ni, nc, jj);
            }
        }
        float *o = out + (int64_t)(i & ~3u) * 16 + 4 * q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 a = acc[k];
            a[0] = mean_div(a[0], md);
            a[1] = mean_div(a[1], md);
            a[2] = mean_div(a[2], md);
            a[3] = mean_div(a[3], md);
            *reinterpret_cast<f32x4 *>(o + 16 * k) = a;
        }
        i = nxt;
    }
    const int nv = (int)(M & 3);   // the last, partial group (byte loads), by the first team of workgroup 0
    if (nv != 0 && blockIdx.x == 0 && threadIdx.x < 4) {
        const int64_t m0 = (int64_t)full;
        f32x4 acc[4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            const uint8_t *cp = codes + (int64_t)r * code_stride + m0;
            const uint8_t *lp = levels + (int64_t)r * level_stride + (PACKED6 ? 3 * (m0 >> 2) : m0);
            unsigned c = 0, l = 0;
            for (int k = 0; k < nv; ++k) c |= (unsigned)cp[k] << (8 * k);
            if (PACKED6) {
                l = (unsigned)lp[0] | ((unsigned)lp[1] << 8) | ((unsigned)lp[2] << 16);
            } else {
                for (int k = 0; k < nv; ++k) l |= (unsigned)lp[k] << (8 * k);
            }
            const float2 lr = s_lbub[r];
            dec16_payload<false, PACKED6, true>(acc, c, l, lr.x, lr.y, inv_s, q, cb_bytes, lane_const);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < nv) {
                f32x4 a = acc[k];
                a[0] = mean_div(a[0], md);
                a[1] = mean_div(a[1], md);
                a[2] = mean_div(a[2], md);
                a[3] = mean_div(a[3], md);
                *reinterpret_cast<f32x4 *>(out + (m0 + k) * 16 + 4 * q) = a;
            }
        }
    }
}

constexpr int DEC16_RC_MAX = 1024;   // payloads whose (lb, ub - lb) fit behind the codebook image (8 KB)

template <bool P6>
static void launch_dec16_rc(int R, const uint8_t *codes, const uint8_t *levels, const float *lb_ub, int64_t cs, int64_t ls,
                            int64_t bs, const float *cb, int64_t M, int K, int n_bit, float *out, hipStream_t st) {
    static const int bpc = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_decode_sum_d16u8_rc_kernel<P6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 8 * DEC16_RC_MAX);
        (void)hipGetLastError();
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hsq_decode_sum_d16u8_rc_kernel<P6>, DEC16_THREADS,
                                                         (size_t)256 * 64 * sizeof(float) + 8 * 64) != hipSuccess || n < 1)
            n = 1;
        return n;
    }();
    const int64_t total = ((M + 3) >> 2) * 4;
    int64_t blocks = (total + DEC16_THREADS - 1) / DEC16_THREADS;
    if (blocks > (int64_t)cu_count() * bpc) blocks = (int64_t)cu_count() * bpc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_d16u8_rc_kernel<P6>), dim3((unsigned)blocks), dim3(DEC16_THREADS),
                       (size_t)K * 64 * sizeof(float) + 8 * (size_t)R, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, K, n_bit, out);
}

template <int R, bool P6, bool FMA = false>
static void launch_dec16_r(const uint8_t *codes, const uint8_t *levels, const float *lb_ub, int64_t cs, int64_t ls, int64_t bs,
                           const float *cb, int64_t M, int K, int n_bit, float *out, hipStream_t st) {
    static const int bpc = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_decode_sum_d16u8_r_kernel<R, P6, FMA>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipGetLastError();
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hsq_decode_sum_d16u8_r_kernel<R, P6, FMA>, DEC16_THREADS,
                                                         (size_t)256 * 64 * sizeof(float)) != hipSuccess || n < 1)
            n = 1;
        return n;
    }();
    const int64_t total = ((M + 3) >> 2) * 4;
    int64_t blocks = (total + DEC16_THREADS - 1) / DEC16_THREADS;
    if (blocks > (int64_t)cu_count() * bpc) blocks = (int64_t)cu_count() * bpc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_d16u8_r_kernel<R, P6, FMA>), dim3((unsigned)blocks), dim3(DEC16_THREADS),
                       (size_t)K * 64 * sizeof(float), st, codes, levels, lb_ub, cs, ls, bs, cb, M, K, n_bit, out);
}

template <bool P6>
static bool launch_dec16_fixed_r(int R, const uint8_t *codes, const uint8_t *levels, const float *lb_ub, int64_t cs, int64_t ls,
                                 int64_t bs, const float *cb, int64_t M, int K, int n_bit, float *out, hipStream_t st, bool fma = false) {
    if (M >= ((int64_t)1 << 31) - ((int64_t)1 << 21)) return false;
    // the lane's payload inside a group of four travels in the 32-bit offset of its loads: up to 3 strides + the section
    if (cs < 0 || ls < 0 || 3 * (cs > ls ? cs : ls) + M + 64 >= ((int64_t)1 << 32)) return false;
    if (fma) {   // GQ_AGGREGATE_FMA: built for the power-of-two payload counts (ranks); every other R keeps the exact kernels
        switch (R) {
#define GQ_DEC16_FMA(N) case N: launch_dec16_r<N, P6, true>(codes, levels, lb_ub, cs, ls, bs, cb, M, K, n_bit, out, st); return true;
            GQ_DEC16_FMA(2) GQ_DEC16_FMA(4) GQ_DEC16_FMA(8) GQ_DEC16_FMA(16)
#undef GQ_DEC16_FMA
            default: break;
        }
    }
    switch (R) {
#define GQ_DEC16_CASE(N) case N: launch_dec16_r<N, P6>(codes, levels, lb_ub, cs, ls, bs, cb, M, K, n_bit, out, st); return true;
        GQ_DEC16_CASE(1) GQ_DEC16_CASE(2) GQ_DEC16_CASE(3) GQ_DEC16_CASE(4)
        GQ_DEC16_CASE(5) GQ_DEC16_CASE(6) GQ_DEC16_CASE(7) GQ_DEC16_CASE(8)
        GQ_DEC16_CASE(9) GQ_DEC16_CASE(10) GQ_DEC16_CASE(11) GQ_DEC16_CASE(12)
        GQ_DEC16_CASE(13) GQ_DEC16_CASE(14) GQ_DEC16_CASE(15) GQ_DEC16_CASE(16)
#undef GQ_DEC16_CASE
        default:
            if (R > DEC16_RC_MAX) return false;
            launch_dec16_rc<P6>(R, codes, levels, lb_ub, cs, ls, bs, cb, M, K, n_bit, out, st);
            return true;
    }
}

// Levels + decode of ONE payload in a single launch: decompress(compress(g)) given the encode's codes and projections
// (nearest_neighbor_compressor.py:74-90 after the argmax; what PSQuantizer.record does for a user, ps_quantizer.py:37, and
// what a single-rank step runs: the level kernel's 5 us are launch and round-trip latency, not work).  The kernel is
// hsq_decode_sum_d16u8_r_kernel<1> whose level word is not loaded but worked out from the team's four projections with
// the level kernel's own arithmetic (LevelQuant), written to the wire like the level kernel writes it, and decoded.
// (lb, ub): every workgroup folds the encode's (min,max) partials in its prologue, under the staging of the codebook image.
template <bool PACKED6>
__global__ __launch_bounds__(DEC16_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void hsq_levels_decode_d16u8_kernel(
    const float *__restrict__ u, const uint8_t *__restrict__ codes, const float *__restrict__ partials,
    float *__restrict__ lb_ub, uint8_t *__restrict__ levels, int n_bit, int random_mode, const float *__restrict__ r,
    uint64_t seed, const float *__restrict__ cb, int64_t M, int K, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_cb[];   // [K][4 copies][16] at LDS address 0, then the fold's 2 x 16 floats
    static_assert(GQ_MAX_PARTIALS == DEC16_THREADS, "one (min,max) pair per thread");
    const unsigned full = (unsigned)(M >> 2) * 4u;
    const unsigned stride = gridDim.x * DEC16_THREADS;
    unsigned i = blockIdx.x * DEC16_THREADS + threadIdx.x;
    const int q = threadIdx.x & 3;
    // first item's code word and projection, the partials and the codebook: one round trip for all of it
    unsigned c4 = 0;
    float uq = 0.0f;
    if (i < full) {
        c4 = *reinterpret_cast<const unsigned *>(codes + (i & ~3u));
        uq = u[i];                 // item i = (group i >> 2, quarter q): the group's subvector q is number (i & ~3) + q = i
    }
    const int final_flag = ws_counter(partials)[2];
    const float2 part = reinterpret_cast<const float2 *>(partials)[threadIdx.x];
    constexpr int STAGE = 256 * 16 / DEC16_THREADS;
    f32x4 stage[STAGE];
#pragma unroll
    for (int n = 0; n < STAGE; ++n) {
        const int e = threadIdx.x + n * DEC16_THREADS;
        if (e < K * 16) stage[n] = *reinterpret_cast<const f32x4 *>(cb + (e >> 4) * 16 + 4 * (e & 3));
    }
    float *const s_red = s_cb + K * 64;
    {
        float lo = wave_min(part.x), hi = wave_max(part.y);
        if (__ballot(part.x != part.x) != 0) lo = hi = __uint_as_float(0x7FC00000u);   // a (NaN, NaN) pair: lb = ub = NaN
        if ((threadIdx.x & 63) == 0) {
            s_red[threadIdx.x >> 6] = lo;
            s_red[16 + (threadIdx.x >> 6)] = hi;
        }
    }
#pragma unroll
    for (int n = 0; n < STAGE; ++n) {
        const int e = threadIdx.x + n * DEC16_THREADS;
        const int row = e >> 4, c = (e >> 2) & 3, qq = e & 3;
        if (e < K * 16) *reinterpret_cast<f32x4 *>(s_cb + row * 64 + c * 16 + 4 * qq) = stage[n];
    }
    __syncthreads();
    float lb, ub;
    if (final_flag != 0) {   // the caller's final pair (gq_hsq.h)
        lb = partials[0];
        ub = partials[1];
    } else {
        lb = s_red[0];
        ub = s_red[16];
        bool nan = lb != lb;
#pragma unroll
        for (int w = 1; w < DEC16_THREADS / 64; ++w) {
            nan = nan || (s_red[w] != s_red[w]);
            lb = fminf(lb, s_red[w]);
            ub = fmaxf(ub, s_red[16 + w]);
        }
        if (nan) lb = ub = __uint_as_float(0x7FC00000u);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        lb_ub[0] = lb;
        lb_ub[1] = ub;
    }
    const LevelQuant lq(lb, ub, n_bit, random_mode, r, seed);
    const float range = ub - lb;
    const float inv_s = 1.0f / (float)(1 << (n_bit & 31));
    const unsigned lane_const = (unsigned)(((threadIdx.x >> 3) & 3) * 64 + 16 * q);
    const char *const cb_bytes = reinterpret_cast<const char *>(s_cb);
    // the four levels of a team as the word the decode reads: a byte (or 6 bits) each, OR-ed over the team by two quad-permute steps
    auto team_levels = [&](int l) {
        int w = PACKED6 ? ((l & 63) << (6 * q)) : ((l & 255) << (8 * q));
        w |= __builtin_amdgcn_mov_dpp(w, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
        w |= __builtin_amdgcn_mov_dpp(w, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
        return (unsigned)w;
    };
    // the last, partial group of a tensor whose M is not a multiple of 4, by the first team of workgroup 0, before the loop
    const int nv = (int)(M & 3);
    if (nv != 0 && blockIdx.x == 0 && threadIdx.x < 4) {
        const int64_t m0 = (int64_t)full;
        unsigned c = 0;
        for (int k = 0; k < nv; ++k) c |= (unsigned)codes[m0 + k] << (8 * k);
        const int l = q < nv ? lq.level(u[m0 + q], m0 + q) : 0;
        const unsigned l4 = team_levels(l);
        if (q == 0) {
            if (PACKED6) {   // a group is always stored whole (slots past M hold 0)
                uint8_t *dst = levels + 3 * (m0 >> 2);
                dst[0] = (uint8_t)l4;
                dst[1] = (uint8_t)(l4 >> 8);
                dst[2] = (uint8_t)(l4 >> 16);
            } else {
                for (int k = 0; k < nv; ++k) levels[m0 + k] = (uint8_t)(l4 >> (8 * k));
            }
        }
        f32x4 acc[4];
        dec16_payload<true, PACKED6, true>(acc, c, l4, lb, range, inv_s, q, cb_bytes, lane_const);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) *reinterpret_cast<f32x4 *>(out + (m0 + k) * 16 + 4 * q) = acc[k];
    }
#pragma clang loop unroll(disable)
    while (i < full) {
        const unsigned nxt = i + stride;
        const unsigned pre = nxt < full ? nxt : i;
        const unsigned m0 = i & ~3u;
        const unsigned l4 = team_levels(lq.level(uq, (int64_t)i));
        if (PACKED6) {   // the group's three bytes by lanes 0..2 of the team: one byte-store instruction
            if (q < 3) levels[3 * (m0 >> 2) + q] = (uint8_t)(l4 >> (8 * q));
        } else if (q == 0) {
            *reinterpret_cast<unsigned *>(levels + m0) = l4;
        }
        f32x4 acc[4];
        dec16_payload<true, PACKED6, true>(acc, c4, l4, lb, range, inv_s, q, cb_bytes, lane_const);
        c4 = *reinterpret_cast<const unsigned *>(codes + (pre & ~3u));
        uq = u[pre];
        float *o = out + (int64_t)m0 * 16 + 4 * q;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(o + 16 * k) = acc[k];
        i = nxt;
    }
}

// any d: one thread per output float.
template <typename CodeT, typename LevelT>
__global__ __launch_bounds__(DEC_THREADS) void hsq_decode_sum_scalar_kernel(
    const CodeT *__restrict__ codes, const LevelT *__restrict__ levels, const float *__restrict__ lb_ub,
    int64_t code_stride, int64_t level_stride, int64_t lbub_stride, const float *__restrict__ cb, int R, int64_t M,
    int d, int n_bit, float *__restrict__ out) {
    const int64_t total = M * d;
    const float s = 1.0f / (float)(1 << (n_bit & 31));   // inv_s, see level_to_norm
    const MeanDiv md = mean_div_of(R);
    const int64_t stride = (int64_t)gridDim.x * DEC_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * DEC_THREADS + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / d;
        const int jj = (int)(i - m * d);
        float acc = 0.0f;
        for (int r = 0; r < R; ++r) {
            const int code = (int)codes[(int64_t)r * code_stride + m];
            const float lb = lb_ub ? lb_ub[r * lbub_stride] : 0.0f;
            const float range = lb_ub ? (lb_ub[r * lbub_stride + 1] - lb) : 0.0f;
            const float n = level_to_norm<LevelT>(levels[(int64_t)r * level_stride + m], lb, range, s);
            const float dec = cb[(int64_t)code * d + jj] * n;
            acc = (r == 0) ? dec : acc + dec;
        }
        if (md.apply) acc = mean_div(acc, md);
        out[i] = acc;
    }
}

template <typename CodeT, typename LevelT>
static int launch_decode(const CodeT *codes, const LevelT *levels, const float *lb_ub, int64_t cs, int64_t ls,
                         int64_t bs, const float *cb, int R, int64_t M, int d, int K, int n_bit, float *out,
                         hipStream_t st) {
    const bool fma = (n_bit & GQ_AGGREGATE_FMA) != 0 && R >= 2;   // opt-in fused accumulation (the d16 / byte kernels, R = 2, 4, 8, 16)
    n_bit &= 0xFF;
    const int64_t cap = (int64_t)cu_count() * 8;
    if constexpr (sizeof(CodeT) == 1 && (std::is_same<LevelT, uint8_t>::value || std::is_same<LevelT, Packed6>::value)) {
        constexpr bool P6 = std::is_same<LevelT, Packed6>::value;
        // the packed form reads codes and levels with alignment-free dword loads (a group of levels starts at any byte)
        const uintptr_t align = P6 ? 0 : (reinterpret_cast<uintptr_t>(codes) | (uintptr_t)cs |
                                          reinterpret_cast<uintptr_t>(levels) | (uintptr_t)ls);
        if (d == 16 && K <= 256 && lb_ub && (align & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(cb) & 15) == 0 && (M >= (int64_t)K || P6)) {
            if (launch_dec16_fixed_r<P6>(R, reinterpret_cast<const uint8_t *>(codes), reinterpret_cast<const uint8_t *>(levels),
                                         lb_ub, cs, ls, bs, cb, M, K, n_bit, out, st, fma)) {
                GQ_CHECK_LAUNCH("gq_hsq_decode_sum");
                return GQ_OK;
            }
            const int64_t total = ((M + 3) >> 2) * 4;
            const size_t lds = (size_t)K * 64 * sizeof(float);   // four copies of every row
            static const int bpc = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hsq_decode_sum_d16u8_kernel<P6>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                (void)hipGetLastError();
                int n = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hsq_decode_sum_d16u8_kernel<P6>, DEC16_THREADS,
                                                                 (size_t)256 * 64 * sizeof(float)) != hipSuccess || n < 1)
                    n = 1;
                return n;
            }();
            int64_t blocks = (total + DEC16_THREADS - 1) / DEC16_THREADS;
            if (blocks > (int64_t)cu_count() * bpc) blocks = (int64_t)cu_count() * bpc;
            hipLaunchKernelGGL(hsq_decode_sum_d16u8_kernel<P6>, dim3((unsigned)blocks), dim3(DEC16_THREADS), lds, st,
                               reinterpret_cast<const uint8_t *>(codes), reinterpret_cast<const uint8_t *>(levels),
                               lb_ub, cs, ls, bs, cb, R, M, K, n_bit, out);
            GQ_CHECK_LAUNCH("gq_hsq_decode_sum");
            return GQ_OK;
        }
    }
    if constexpr (std::is_same<LevelT, Packed6>::value) {
        return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_decode_sum: GQ_LEVELS_PACKED6 is served for d = 16, K <= 256, byte codes, "
                                        "16-byte aligned out / codebook");
    } else {
    if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(cb) & 15) == 0) {
        const int64_t total = M * (d >> 2);
        int64_t blocks = (total + DEC_THREADS - 1) / DEC_THREADS;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        const size_t lds = (size_t)K * cb_row_stride(d) * sizeof(float);
        // stage the codebook in LDS when it fits and the launch is big enough to amortise it
        if (lds <= 64 * 1024 && total >= (int64_t)K * d) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_v4_kernel<CodeT, LevelT, true>), dim3((unsigned)blocks),
                               dim3(DEC_THREADS), lds, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, d, K, n_bit, out);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_v4_kernel<CodeT, LevelT, false>), dim3((unsigned)blocks),
                               dim3(DEC_THREADS), 0, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, d, K, n_bit, out);
        }
    } else {
        const int64_t total = M * d;
        int64_t blocks = (total + DEC_THREADS - 1) / DEC_THREADS;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hsq_decode_sum_scalar_kernel<CodeT, LevelT>), dim3((unsigned)blocks),
                           dim3(DEC_THREADS), 0, st, codes, levels, lb_ub, cs, ls, bs, cb, R, M, d, n_bit, out);
    }
    GQ_CHECK_LAUNCH("gq_hsq_decode_sum");
    return GQ_OK;
    }
}

template <typename CodeT>
static int dispatch_levels(const CodeT *codes, int64_t cs, const void *levels, int level_bytes, int64_t ls,
                           const float *lb_ub, int64_t bs, const float *cb, int R, int64_t M, int d, int K, int n_bit,
                           float *out, hipStream_t st) {
    switch (level_bytes) {
        case 0:
            return launch_decode<CodeT, float>(codes, static_cast<const float *>(levels), nullptr, cs, ls, 0, cb, R, M,
                                               d, K, 0, out, st);
        case 1:
            return launch_decode<CodeT, uint8_t>(codes, static_cast<const uint8_t *>(levels), lb_ub, cs, ls, bs, cb, R,
                                                 M, d, K, n_bit, out, st);
        case 2:
            return launch_decode<CodeT, uint16_t>(codes, static_cast<const uint16_t *>(levels), lb_ub, cs, ls, bs, cb,
                                                  R, M, d, K, n_bit, out, st);
        case 4:
            return launch_decode<CodeT, int32_t>(codes, static_cast<const int32_t *>(levels), lb_ub, cs, ls, bs, cb, R,
                                                 M, d, K, n_bit, out, st);
        case GQ_LEVELS_PACKED6:
            if constexpr (sizeof(CodeT) == 1)
                return launch_decode<CodeT, Packed6>(codes, static_cast<const Packed6 *>(levels), lb_ub, cs, ls, bs, cb, R, M,
                                                     d, K, n_bit, out, st);
            return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_decode_sum: GQ_LEVELS_PACKED6 goes with byte codes");
        default:
            return fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: level_bytes must be 0, 1, 2, 4 or GQ_LEVELS_PACKED6");
    }
}

}  // namespace gq

GQ_API int gq_hsq_decode_sum_strided(const void *codes, int code_bytes, int64_t code_stride_bytes, const void *levels,
                                     int level_bytes, int64_t level_stride_bytes, const float *lb_ub,
                                     int64_t lbub_stride_bytes, const float *codebook, int R, int64_t M, int d, int K,
                                     int n_bit, float *out, void *stream) {
    if (M < 1 || d < 1 || K < 1 || R < 1)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: bad sizes R=%d M=%lld d=%d K=%d", R, (long long)M, d, K);
    if (!codes || !levels || !codebook || !out) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: null pointer");
    if ((n_bit & ~(0xFF | GQ_AGGREGATE_FMA)) != 0) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: n_bit %d", n_bit);
    const int nb = n_bit & 0xFF;
    if (level_bytes != 0 && (!lb_ub || nb < 1 || nb > 30))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: lb_ub / n_bit required with integer levels");
    const int lsz = level_bytes == 0 ? 4 : (level_bytes == GQ_LEVELS_PACKED6 ? 1 : level_bytes);
    if (level_bytes == GQ_LEVELS_PACKED6 && (nb > 6 || code_bytes != 1))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: GQ_LEVELS_PACKED6 holds levels up to 63 and goes with byte codes");
    if ((code_bytes != 1 && code_bytes != 4) || code_stride_bytes % code_bytes || level_stride_bytes % lsz ||
        lbub_stride_bytes % 4)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum: strides must be multiples of the element size");
    hipStream_t st = gq::as_stream(stream);
    const int64_t cs = code_stride_bytes / code_bytes, ls = level_stride_bytes / lsz, bs = lbub_stride_bytes / 4;
    if (code_bytes == 1)
        return gq::dispatch_levels<uint8_t>(static_cast<const uint8_t *>(codes), cs, levels, level_bytes, ls, lb_ub, bs,
                                            codebook, R, M, d, K, n_bit, out, st);
    return gq::dispatch_levels<int32_t>(static_cast<const int32_t *>(codes), cs, levels, level_bytes, ls, lb_ub, bs,
                                        codebook, R, M, d, K, n_bit, out, st);
}

GQ_API int gq_hsq_decode_sum(const void *codes, int code_bytes, const void *levels, int level_bytes,
                             const float *lb_ub, const float *codebook, int R, int64_t M, int d, int K, int n_bit,
                             float *out, void *stream) {
    const int lsz = level_bytes == 0 ? 4 : level_bytes;
    if (level_bytes == GQ_LEVELS_PACKED6)   // contiguous payloads: a section is 3 * ceil(M / 4) bytes
        return gq_hsq_decode_sum_strided(codes, code_bytes, M * (int64_t)code_bytes, levels, level_bytes, 3 * ((M + 3) / 4),
                                         lb_ub, 8, codebook, R, M, d, K, n_bit, out, stream);
    return gq_hsq_decode_sum_strided(codes, code_bytes, M * (int64_t)code_bytes, levels, level_bytes, M * (int64_t)lsz,
                                     lb_ub, 8, codebook, R, M, d, K, n_bit, out, stream);
}

GQ_API int gq_hsq_levels_decode(const float *u, int64_t M, int n_bit, int random_mode, const float *r, uint64_t seed,
                                const float *minmax_partials, float *lb_ub, void *levels, int level_bytes, const void *codes,
                                const float *codebook, int K, float *out, void *stream) {
    if (M < 1 || n_bit < 1 || n_bit > 8 || K < 1)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode: bad sizes M=%lld n_bit=%d K=%d", (long long)M, n_bit, K);
    if (!u || !minmax_partials || !lb_ub || !levels || !codes || !codebook || !out)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode: null pointer");
    if (random_mode < GQ_RANDOM_OFF || random_mode > GQ_RANDOM_DEVICE || (random_mode == GQ_RANDOM_GIVEN && !r))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode: random_mode %d", random_mode);
    const int64_t top = ((int64_t)1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0);
    if ((level_bytes == 1 && top > 255) || (level_bytes == GQ_LEVELS_PACKED6 && top > 63))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode: level_bytes=%d cannot hold level %lld", level_bytes, (long long)top);
    const uintptr_t a4 = reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(codes) |
                         (level_bytes == 1 ? reinterpret_cast<uintptr_t>(levels) : 0);
    const uintptr_t a16 = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(codebook);
    if ((level_bytes != 1 && level_bytes != GQ_LEVELS_PACKED6) || K > 256 || (a4 & 3) || (a16 & 15) ||
        M >= ((int64_t)1 << 31) - ((int64_t)1 << 21))
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_levels_decode: served for d = 16, K <= 256, byte codes, byte or packed 6-bit levels, "
                                            "4-byte aligned u / codes / levels, 16-byte aligned out / codebook (use gq_hsq_levels + gq_hsq_decode_sum)");
    hipStream_t st = gq::as_stream(stream);
    const size_t lds = (size_t)K * 64 * sizeof(float) + 32 * sizeof(float);
    auto launch = [&](auto p6) {
        constexpr bool P6 = decltype(p6)::value;
        static const int bpc = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gq::hsq_levels_decode_d16u8_kernel<P6>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 128);
            (void)hipGetLastError();
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gq::hsq_levels_decode_d16u8_kernel<P6>, gq::DEC16_THREADS,
                                                             (size_t)64 * 1024 + 128) != hipSuccess || n < 1)
                n = 1;
            return n;
        }();
        const int64_t total = ((M + 3) >> 2) * 4;
        int64_t blocks = (total + gq::DEC16_THREADS - 1) / gq::DEC16_THREADS;
        if (blocks > (int64_t)gq::cu_count() * bpc) blocks = (int64_t)gq::cu_count() * bpc;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gq::hsq_levels_decode_d16u8_kernel<P6>), dim3((unsigned)blocks), dim3(gq::DEC16_THREADS), lds,
                           st, u, static_cast<const uint8_t *>(codes), minmax_partials, lb_ub, static_cast<uint8_t *>(levels), n_bit,
                           random_mode, r, seed, codebook, M, K, out);
    };
    if (level_bytes == GQ_LEVELS_PACKED6)
        launch(std::true_type{});
    else
        launch(std::false_type{});
    GQ_CHECK_LAUNCH("gq_hsq_levels_decode");
    return GQ_OK;
}
