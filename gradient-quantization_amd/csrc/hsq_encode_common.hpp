// Device helpers shared by the HSQ encode kernels (hsq_encode.hip, hsq_encode_pf.hip).
#pragma once
#include "gq_common.hpp"

namespace gq {

constexpr int ENC_THREADS = 256;
constexpr int ENC_WAVES = ENC_THREADS / 64;

__device__ __forceinline__ void swap32(float &x, float &y) {
    // v_permlane32_swap: lanes 32..63 of x <-> lanes 0..31 of y.
    // After it: x = (x.lo, y.lo), y = (x.hi, y.hi).
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32(int &x, int &y) {
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
    x = (int)r[0];
    y = (int)r[1];
}

// Strict '>' keeps the FIRST maximum when candidates are visited in ascending index.
__device__ __forceinline__ void take_if_greater(float &bv, int &bi, float v, int idx) {
    const bool gt = fabsf(v) > fabsf(bv);
    bv = gt ? v : bv;
    bi = gt ? idx : bi;
}

// ---- non-finite gradients --------------------------------------------------------------------------------
// The reference's torch.argmax ranks NaN above every number and keeps the first one
// (nearest_neighbor_compressor.py:72), and torch.min / torch.max propagate NaN into (lb, ub)
// (probabilistic_scalar_compressor.py:13-14).  The kernels' hot loops compare with '>' and fminf / fmaxf, which
// ignore NaN; a subvector that holds a NaN or an infinity is therefore re-done by the wave-wide scan below (a
// rare path), and a NaN projection poisons (lb, ub) explicitly.  Bit tests: two files of the library are built
// with -fno-honor-nans, where `x != x` folds to false.
__device__ __forceinline__ bool nan_bits(float v) { return (__float_as_uint(v) & 0x7FFFFFFFu) > 0x7F800000u; }
__device__ __forceinline__ bool nonfinite_bits(float v) { return (__float_as_uint(v) & 0x7F800000u) == 0x7F800000u; }
// |v| as an ordered integer with every NaN equal and above +infinity
__device__ __forceinline__ unsigned nan_rank(float v) {
    const unsigned a = __float_as_uint(v) & 0x7FFFFFFFu;
    return a > 0x7F800000u ? 0x7F800001u : a;
}
__device__ __forceinline__ void take_if_greater_nan(float &bv, int &bi, float v, int idx) {
    const bool gt = nan_rank(v) > nan_rank(bv);
    bv = gt ? v : bv;
    bi = gt ? idx : bi;
}
// first maximum of (value, index) pairs across the wave in torch.argmax's order; every lane gets the result.
// A lane without a candidate passes idx = 0x7FFFFFFF.
__device__ __forceinline__ void wave_first_max_nan(float &bv, int &bi) {
    unsigned br = nan_rank(bv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const unsigned orr = nan_rank(ov);
        const bool take = oi != 0x7FFFFFFF && (bi == 0x7FFFFFFF || orr > br || (orr == br && oi < bi));
        bv = take ? ov : bv;
        bi = take ? oi : bi;
        br = take ? orr : br;
    }
}
// The same reduction without a trip through LDS per step (__shfl_xor is ds_bpermute_b32: ~100 cycles, and the six steps
// depend on each other): quad permutes and row mirrors as DPP moves, v_permlane16_swap / v_permlane32_swap for the last
// two steps.  Candidates are ordered by the 64-bit key (nan_rank << 32 | ~index): the largest rank, the lowest index
// among equals.  Every lane holds a candidate and gets the result.
__device__ __forceinline__ void wave_first_max_nan_dpp(float &bv, int &bi) {
    unsigned br = nan_rank(bv);
    auto better = [&](unsigned orr, int oi, float ov) {
        const uint64_t mine = ((uint64_t)br << 32) | (unsigned)~bi, theirs = ((uint64_t)orr << 32) | (unsigned)~oi;
        const bool take = theirs > mine;
        br = take ? orr : br;
        bi = take ? oi : bi;
        bv = take ? ov : bv;
    };
#define GQ_DPP_STEP(CTRL)                                                                                          \
    better(__builtin_amdgcn_update_dpp(0u, br, CTRL, 0xF, 0xF, false),                                             \
           (int)__builtin_amdgcn_update_dpp(0u, (unsigned)bi, CTRL, 0xF, 0xF, false),                              \
           __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(bv), CTRL, 0xF, 0xF, false)))
    GQ_DPP_STEP(0xB1);    // quad_perm [1,0,3,2]
    GQ_DPP_STEP(0x4E);    // quad_perm [2,3,0,1]
    GQ_DPP_STEP(0x141);   // row_half_mirror: the other quad of each 8
    GQ_DPP_STEP(0x140);   // row_mirror: the other 8 of each row of 16
#undef GQ_DPP_STEP
    {   // rows 0 <-> 1, 2 <-> 3: of two copies, the first keeps the even rows' values in both rows, the second the odd rows'
        const auto r = __builtin_amdgcn_permlane16_swap(br, br, false, false);
        const auto i = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
        const auto v = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
        br = r[0], bi = (int)i[0], bv = __uint_as_float(v[0]);
        better(r[1], (int)i[1], __uint_as_float(v[1]));
    }
    {   // lanes 0-31 <-> 32-63
        const auto r = __builtin_amdgcn_permlane32_swap(br, br, false, false);
        const auto i = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
        const auto v = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
        br = r[0], bi = (int)i[0], bv = __uint_as_float(v[0]);
        better(r[1], (int)i[1], __uint_as_float(v[1]));
    }
}
// Wave-wide exact argmax for ONE subvector (rare path of the exact kernels): lane L scores codewords L, L + 64, ...
// with the reference's fmaf chain, c(k, e) and v(e) supplied by the caller (v wave-uniform).
template <class RowGet, class VGet>
__device__ __forceinline__ void nonfinite_argmax(int K, int d, RowGet c, VGet v, float &val, int &idx) {
    const int lane = threadIdx.x & 63;
    float bv = 0.0f;
    int bi = 0x7FFFFFFF;
    for (int k = lane; k < K; k += 64) {
        float acc = 0.0f;
        for (int e = 0; e < d; ++e) acc = __fmaf_rn(c(k, e), v(e), acc);
        if (bi == 0x7FFFFFFF || nan_rank(acc) > nan_rank(bv)) {
            bv = acc;
            bi = k;
        }
    }
    wave_first_max_nan(bv, bi);
    val = bv;
    idx = bi;
}
// order-mapped (hsq_pf_common.hpp: order_map) images of -NaN / +NaN: below every mapped number / above every one.
// Sent to a tensor's (min, max) words by atomicMin / atomicMax they make its (lb, ub) NaN.
constexpr unsigned MAPPED_NAN_LO = 0x003FFFFFu;   // order_map(0xFFC00000)
constexpr unsigned MAPPED_NAN_HI = 0xFFC00000u;   // order_map(0x7FC00000)

// Row of the 32x32 MFMA result held in accumulator register r of a lane in half h
// is  (r&3) + 8*(r>>2) + 4*h ; this is the h-independent part.
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// Workspace layout (gq_hsq_workspace_bytes):
//   [ (min,max) x GQ_MAX_PARTIALS | int32 x4: fix-up count, ticket, `final` flag, - | fix-up log int32[M] ]
// The int32 block must be zero before the first use (the prefilter kernel's last workgroup re-zeroes
// what it uses).  GQ_MAIN_PARTIALS caps the persistent grids.
constexpr int GQ_MAIN_PARTIALS = GQ_MAX_PARTIALS - GQ_FIXUP_PARTIALS;
__host__ __device__ inline int *ws_counter(float *ws) { return reinterpret_cast<int *>(ws + 2 * GQ_MAX_PARTIALS); }
__host__ __device__ inline const int *ws_counter(const float *ws) {
    return reinterpret_cast<const int *>(ws + 2 * GQ_MAX_PARTIALS);
}
__host__ __device__ inline int *ws_worklist(float *ws) { return reinterpret_cast<int *>(ws + 2 * GQ_MAX_PARTIALS) + 4; }

// Per-block (min,max) of u -> partials[2*blockIdx.x], and block 0 pads the unused slots.  The level kernel
// (gq_hsq_levels) folds the pairs into (lb, ub): every one of its workgroups reads the 8 KiB from L2, which costs it
// ~0.3 us once -- less than a last-workgroup fold inside the encode did (ticket round trip + fold: ~1.8 us on the
// encode's critical path; profiles/r02_b_pf_prologue_stamps.txt).
// sawnan (wave-uniform): a projection of this wave's tiles is NaN -> the workgroup's pair is (NaN, NaN) and the
// level kernel's fold makes (lb, ub) NaN, as torch.min / torch.max do.
template <int WAVES = ENC_WAVES>
__device__ __forceinline__ void write_minmax_partials(float lmin, float lmax, float *__restrict__ partials,
                                                      bool sawnan = false) {
    __shared__ float s_min[WAVES], s_max[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lmin = wave_min(lmin);
    lmax = wave_max(lmax);
    if (sawnan) lmin = lmax = __uint_as_float(0x7FC00000u);
    if (lane == 0) {
        s_min[wave] = lmin;
        s_max[wave] = lmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = s_min[0], b = s_max[0];
        bool anynan = nan_bits(a);
        const int launched = (int)(blockDim.x >> 6);   // a kernel may run with fewer waves than it is built for
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            if (w >= launched) break;
            anynan = anynan || nan_bits(s_min[w]);
            a = fminf(a, s_min[w]);
            b = fmaxf(b, s_max[w]);
        }
        if (anynan) a = b = __uint_as_float(0x7FC00000u);
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
        if (blockIdx.x == 0) ws_counter(partials)[2] = 0;  // pairs are partials, not yet the final (lb, ub)
    }
    if (blockIdx.x == 0) {
        for (int i = gridDim.x + threadIdx.x; i < GQ_MAX_PARTIALS; i += blockDim.x) {
            partials[2 * i] = INFINITY;
            partials[2 * i + 1] = -INFINITY;
        }
    }
}


// Resident workgroups per CU of a kernel (occupancy API): the persistent grids are sized to
// exactly one resident wave of workgroups so that no workgroup queues behind another.
template <typename KernelT>
static int resident_blocks_per_cu(KernelT kernel, int threads, size_t lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) n = 1;
    return n;
}

// hsq_encode_pf.hip: f16 MFMA prefilter + exact f32 rescoring + second pass + exact scans (d = 8 / 12 / 16 / 24 / 32, K <= 256 and a multiple of 4).
template <typename CodeT>
int launch_encode_pf(const float *grad, const float *codebook, int64_t M, int d, int K, CodeT *codes, float *u, float *workspace,
                     hipStream_t st, int profile_slot = -1);

// LDS plan of the operand-staging kernels (hsq_encode_lds_kernel, pvq_encode_lds_kernel): false if (d, K) does not fit
constexpr int LDS_ROW_PAD = 4;
bool lds_plan(int d, int K, int *dpad, int *chunk_rows, size_t *bytes, int *waves = nullptr);

int launch_encode_pfd_paged(const float *grad, const float *codebook, int64_t M, int d, int K, int32_t *codes, float *u,
                            float *workspace, hipStream_t st);

}  // namespace gq
