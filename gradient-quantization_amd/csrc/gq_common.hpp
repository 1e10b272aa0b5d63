// Shared host/device helpers for libgq_hsq.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gq_hsq.h"

#define GQ_API extern "C" __attribute__((visibility("default")))
// per-variant launchers behind the exported entry points of gq_api.hip (gq_internal.h): C names, not exported
#define GQ_INTERNAL extern "C" __attribute__((visibility("hidden")))

namespace gq {

// Thread-local text of the last failure; returned by gq_last_error().
char *last_error_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Number of compute units of the current device (cached).
int cu_count();
// the start / stop events of a gq_profile_read slot (created on first use); false for slot < 0 or a failure:
// the caller then launches plainly
bool profile_events(int slot, hipEvent_t *start, hipEvent_t *stop);

#define GQ_CHECK_LAUNCH(what)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) return gq::fail(GQ_ERR_HIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

// ---- device helpers -------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// sum / R of the parameter-server mean (ps_quantizer.py:48: torch.stack(...).mean(0) divides the sum by R).  For R a power
// of two -- 1, 2, 4, 8 ranks -- the division is an exact scaling: x * (1/R) is the same correctly rounded x * 2^-k, subnormal
// results included, for ONE VALU operation instead of the ~10 of the IEEE division sequence (16 divisions per lane and
// iteration were a quarter of the R = 8 decode's instructions).  For an ODD R (3, 5, 7 ... users) the quotient by the
// constant takes four: y = RN(1/R) once, q0 = RN(x y), r = x - q0 R (exact in one fma), RN(q0 + r y) is the correctly
// rounded x / R (Markstein's correction step), with one v_max_f32 on the residual for the one case it breaks (x = +-inf:
// inf - inf).  Subnormal quotients included: there q0 is on the 2^-149 grid, r is exact, and the only way the rounding of
// q0 + r y could differ from that of x / R is a quotient exactly half way between two grid points, which needs an even
// divisor -- with R = 6, 10, 12 exactly those ties come out wrong (2.8 M of the 2^32 inputs on the CPU), which is why an
// even R that is not a power of two keeps the IEEE sequence.  tools/div_check.hip compares the four operations with the
// device's own x / R for every one of the 2^32 inputs and every odd R up to GQ_ODD_DIV_MAX on the MI355X (0 differences,
// profiles/r03_experiments.txt 18); the kernel-vs-oracle tests cover R = 3 ... 19 incl. subnormal and infinite sums.
#define GQ_ODD_DIV_MAX 4097
struct MeanDiv {
    float fR, inv;
    bool pow2, odd;
    bool apply;   // false: one payload, plain decompress -- the value is stored as decoded (a -0 stays -0)
};
// mean = the call IS the parameter-server aggregate (every multi-tensor decode): torch's sum starts from +0, so an element
// whose payloads all decode to -0 (a QSGD level 0 with a negative sign, a codeword element times a zero norm) comes out
// as +0 even for one payload; the single-tensor entry points with R == 1 are the plain `decompress` and keep the -0.
__host__ __device__ inline MeanDiv mean_div_of(int R, bool mean = false) {
    MeanDiv m;
    m.fR = (float)R;
    m.inv = 1.0f / (float)R;
    m.pow2 = R > 0 && (R & (R - 1)) == 0;
    m.odd = (R & 1) && R >= 3 && R <= GQ_ODD_DIV_MAX;
    m.apply = mean || R > 1;
    return m;
}
__device__ __forceinline__ float odd_quotient(float x, float fR, float y) {
    const float q0 = __fmul_rn(x, y);
    // x = +-inf: the residual is inf - inf = NaN, and the maximum with a number is that number (v_max_f32 returns the
    // operand that is not a NaN), so that the last fma is (-FLT_MAX) y + (+-inf) = +-inf; a NaN x stays one through q0
    const float r = fmaxf(__fmaf_rn(-q0, fR, x), -3.402823466e+38f);
    return __fmaf_rn(r, y, q0);
}
template <class V>   // float or a vector of floats
__device__ __forceinline__ V mean_div(V x, const MeanDiv &m) {
    if (m.odd) {
        // no (+0) + sum here: the four operations turn a -0 into the +0 of the mean by themselves (q0 = -0, r = (+0) R + (-0)
        // = +0, (+0) y + (-0) = +0) and every other x is what x + 0 is
        if constexpr (__is_same(V, float)) {
            return odd_quotient(x, m.fR, m.inv);
        } else {
            V q;
#pragma unroll
            for (int i = 0; i < (int)(sizeof(V) / sizeof(float)); ++i) q[i] = odd_quotient(x[i], m.fR, m.inv);
            return q;
        }
    }
    x = x + 0.0f;   // (+0) + sum, as torch.stack(...).mean(0) accumulates: -0 becomes +0, everything else is unchanged
    return m.pow2 ? x * m.inv : x / m.fR;
}

// a / b for many a and ONE b, correctly rounded, in three operations: y = RN(1/b) is computed once (a true division);
// q0 = RN(a y) is within an ulp of the quotient, r = a - q0 b is exact in one fma, RN(q0 + r y) is the correctly
// rounded quotient (Markstein's correction step).  Valid while nothing on the way is subnormal: the caller checks
// 2^-80 <= b <= 2^20 and 2^-102 <= a <= b (then a/b >= 2^-122, and r, a multiple of 2^-47 ulp-units of a, is
// representable); a == 0 would be fine too but is not worth a test.  Checked against `a / b` on 5.9e9 (a, b) pairs
// on the CPU (every mantissa of b; b with the 16 highest mantissas against every mantissa of a) and on the GPU by
// the kernel-vs-oracle tests.  b == 0 with a == 0 gives 0 * inf = NaN like 0 / 0.
__device__ __forceinline__ float shared_quotient(float a, float b, float y) {
    const float q0 = __fmul_rn(a, y);
    const float r = __fmaf_rn(-q0, b, a);
    return __fmaf_rn(r, y, q0);
}
// max(mx, |a|, |b|) that PROPAGATES NaN (IEEE-754-2019 `maximum`; v_maximum3_f32 is new in gfx950), for the QSGD bucket
// norms: torch.max(|v|, dim=1) makes a bucket's norm NaN when one of its elements is (qsgd_compressor.py:49), which fmaxf
// would hide.  Same cost as v_max3_f32.
__device__ __forceinline__ float absmax3_nan(float mx, float a, float b) {
    float d;
    asm("v_maximum3_f32 %0, |%1|, |%2|, %3" : "=v"(d) : "v"(a), "v"(b), "v"(mx));
    return d;
}
__device__ __forceinline__ float max_nan(float a, float b) {
    float d;
    asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float wave_max_nan(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max_nan(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// GQ_LEVELS_PACKED6: four 6-bit levels per three bytes (include/gq_hsq.h)
struct Packed6 {};
__device__ __forceinline__ void store_packed6(uint8_t *dst, int l0, int l1, int l2, int l3) {   // dst = section + 3 * group
    const unsigned w = (unsigned)(l0 & 63) | ((unsigned)(l1 & 63) << 6) | ((unsigned)(l2 & 63) << 12) | ((unsigned)(l3 & 63) << 18);
    dst[0] = (uint8_t)w;
    dst[1] = (uint8_t)(w >> 8);
    dst[2] = (uint8_t)(w >> 16);
}
// the four levels of group g as one word: three bytes at section + 3g (an unaligned dword read; the fourth byte is the
// next group's, or padding / the next section of the wire: every section is followed by at least one more byte)
__device__ __forceinline__ unsigned load_packed6(const uint8_t *src) {
    unsigned w;
    __builtin_memcpy(&w, src, 4);
    return w;
}

// GQ_RANDOM_DEVICE_KEYED: the seed of a tensor's stream from the caller's seed and the bits of the tensor's (lb, ub)
__device__ __forceinline__ uint64_t keyed_seed(uint64_t seed, float lb, float ub) {
    const uint64_t k = ((uint64_t)__float_as_uint(lb) << 32) | (uint64_t)__float_as_uint(ub);
    return seed ^ (k * 0x9E3779B97F4A7C15ull) ^ (k >> 29);
}

// lane t of every aligned team of four lanes broadcasts its word to the team (quad_perm [t, t, t, t]); all four lanes active
__device__ __forceinline__ unsigned team_word(unsigned w, int t) {
    const int x = (int)w;
    switch (t) {
        case 0: return (unsigned)__builtin_amdgcn_mov_dpp(x, 0x00, 0xF, 0xF, true);
        case 1: return (unsigned)__builtin_amdgcn_mov_dpp(x, 0x55, 0xF, 0xF, true);
        case 2: return (unsigned)__builtin_amdgcn_mov_dpp(x, 0xAA, 0xF, 0xF, true);
        default: return (unsigned)__builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true);
    }
}

// Counter-based uniform [0,1) generator for GQ_RANDOM_DEVICE: a 32-bit avalanche hash (two
// multiply-xorshift rounds, then a third round that folds in the upper halves) of (seed, index); the top
// 24 bits -> k * 2^-24, the same grid of values torch.rand produces for float32.  ~12 VALU operations --
// the 64-bit splitmix64 it replaces cost ~40 and was most of the QSGD compress kernel's time.
// GQ_RANDOM_DEVICE_COUNTER: `seed` is the address of two device words { seed, step counter }; the stream of this launch is
// keyed by both.  Every multi-tensor decode launch adds one to the counter (bump_rng_counter), so launches whose arguments
// never change -- nodes of a HIP graph -- draw fresh numbers every step.  Called first thing in a kernel (a uniform load).
// A caller keeps one pair per (tensor group, user slot), all in one array that one gq_rng_step / gq_mean_rows_step steps.
__device__ __forceinline__ void resolve_seed(int &random_mode, uint64_t &seed) {
    if (random_mode == GQ_RANDOM_DEVICE_COUNTER) {
        const uint64_t *st = reinterpret_cast<const uint64_t *>(static_cast<uintptr_t>(seed));
        uint64_t z = st[1] + 0x9E3779B97F4A7C15ull;      // splitmix64 of the step counter
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        seed = st[0] ^ z ^ (z >> 31);
        random_mode = GQ_RANDOM_DEVICE;
    }
}
__device__ __forceinline__ void bump_rng_counter(uint64_t *state, int npairs) {   // nothing else reads the words meanwhile
    if (state && blockIdx.x == 0 && (int)threadIdx.x < npairs) state[2 * threadIdx.x + 1] += 1;
}

// The small per-step work of an aggregate (gq_step_tail of include/gq_hsq.h): mean of the uncompressed tensors' rows, one step
// of the draws' { seed, step } words, the accumulators' reset.  It depends on EARLIER launches only, so a decode-mean launch
// can take it along (a launch of its own -- gq_mean_rows -- costs ~4 us of kernel and a boundary in a ~70 us step): the first
// lines of the kernel, spread over the grid; by the time a workgroup reaches its tiles its lanes are back together.
struct StepTail {
    const uint8_t *rows;        // [R][row_stride_bytes]: the dense region of the gathered wire (n == 0: no mean)
    int64_t row_stride_bytes, n;
    float *out;
    uint64_t *rng_state;        // nullable
    uint64_t *reset_dst;        // reset_words == 0: none
    const uint64_t *reset_src;
    int R, rng_pairs, reset_words;
};
__device__ __forceinline__ void step_tail_run(const StepTail &t) {
    if (t.n == 0 && !t.rng_state && t.reset_words == 0) return;
    bump_rng_counter(t.rng_state, t.rng_pairs);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < t.reset_words; i += blockDim.x) t.reset_dst[i] = t.reset_src[i];
    if (t.n) {
        const MeanDiv md = mean_div_of(t.R, true);
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += stride) {
            float acc = reinterpret_cast<const float *>(t.rows)[i];
            for (int r = 1; r < t.R; ++r) acc = acc + reinterpret_cast<const float *>(t.rows + (int64_t)r * t.row_stride_bytes)[i];
            t.out[i] = mean_div(acc, md);   // (+0 + row 0 + row 1 + ...) / R, rows ascending, a true division: torch's CPU mean
        }
    }
}

// The rider of the one-rank step's fused level + decode launch (hsq_batched.hip: hsq_levels_ef_tile_kernel<.., OUT>)
struct FusedTail {
    float *dense_mean;          // nullable; mean[(byte offset in the wire - dense_off) / 4 + i]
    int64_t dense_off;
    uint64_t *rng_state;        // nullable
    uint64_t *reset_dst;        // reset_words == 0: none
    const uint64_t *reset_src;
    unsigned *ticket;           // zero between launches (needed when rng_state or reset_words)
    int rng_pairs, reset_words;
};

// The tensors that travel uncompressed (IdenticalCompressor, ps_quantizer.py:18-19: <= 1000 elements each) ride in the same
// launch as the level quantiser / the QSGD compress: workgroup b copies tensors b, b + grid, ... into their place in the
// wire.  dense_table int64[ndense][3] = { source (float *), byte offset in ONE user's wire, elements }.
__device__ __forceinline__ void copy_dense_segments(const int64_t *__restrict__ dense_table, int ndense, uint8_t *__restrict__ wire) {
    typedef const float __attribute__((address_space(1))) *gcf_ptr;
    for (int t = blockIdx.x; t < ndense; t += gridDim.x) {
        const gcf_ptr src = (gcf_ptr)(uintptr_t)dense_table[3 * t];
        float *dst = reinterpret_cast<float *>(wire + dense_table[3 * t + 1]);
        const int64_t n = dense_table[3 * t + 2];
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

__device__ __forceinline__ uint32_t uniform_bits(uint64_t seed, uint64_t idx) {
    uint32_t h = (uint32_t)idx + (uint32_t)seed * 0x9E3779B1u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    h += (uint32_t)(seed >> 32) ^ ((uint32_t)(idx >> 32) * 0x85EBCA77u);
    h *= 0xC2B2AE3Du;
    h ^= h >> 15;
    return h;
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    return (float)(uniform_bits(seed, idx) >> 8) * 5.9604644775390625e-08f;  // 2^-24
}

// acc += (the norm of lane K of this lane's quad) * c as ONE v_fmac_f32_dpp (GQ_AGGREGATE_FMA): the broadcast rides in the
// multiply-add the way it rides in v_mul_f32_dpp for the unfused form.  (The compiler does not fold a DPP move into a
// v_fmac -- the tied accumulator -- and with four explicit moves per payload the fused form was SLOWER than the unfused
// one: 31.3 against 25.9 us at R = 8, profiles/r04_decode_r.txt.)  n_own must have left the VALU two wait states ago
// (DPP read-after-VALU-write hazard, which the compiler does not see inside an asm): quad_norm_ready() right behind its
// definition holds that.
__device__ __forceinline__ float quad_norm_ready(float n_own) {
    asm volatile("s_nop 1" : "+v"(n_own));
    return n_own;
}
template <int K>
__device__ __forceinline__ float fmac_quad(float acc, float n_own, float c) {
    static_assert(K >= 0 && K < 4, "lane of the quad");
    if constexpr (K == 0) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(n_own), "v"(c));
    if constexpr (K == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(n_own), "v"(c));
    if constexpr (K == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(n_own), "v"(c));
    if constexpr (K == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(n_own), "v"(c));
    return acc;
}
// Four of them on one norm as ONE asm block behind its own `s_nop 1`: whatever the compiler puts in front of the block (a
// copy or a spill reload of n_own) has left the VALU two wait states before the first DPP read, and nothing can be
// scheduled between the four.  (The single-instruction form above relies on quad_norm_ready() alone; the kernels use this.)
template <int K>
__device__ __forceinline__ f32x4 fmac_quad4(const f32x4 &acc, float n_own, const f32x4 &c) {
    static_assert(K >= 0 && K < 4, "lane of the quad");
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
#define GQ_FMAC_QUAD4(QP)                                                                     \
    asm("s_nop 1\n\t"                                                                         \
        "v_fmac_f32_dpp %0, %4, %5 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_fmac_f32_dpp %1, %4, %6 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_fmac_f32_dpp %2, %4, %7 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_fmac_f32_dpp %3, %4, %8 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1"     \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                              \
        : "v"(n_own), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]))
    if constexpr (K == 0) GQ_FMAC_QUAD4("[0,0,0,0]");
    if constexpr (K == 1) GQ_FMAC_QUAD4("[1,1,1,1]");
    if constexpr (K == 2) GQ_FMAC_QUAD4("[2,2,2,2]");
    if constexpr (K == 3) GQ_FMAC_QUAD4("[3,3,3,3]");
#undef GQ_FMAC_QUAD4
    return f32x4{a0, a1, a2, a3};
}

}  // namespace gq
