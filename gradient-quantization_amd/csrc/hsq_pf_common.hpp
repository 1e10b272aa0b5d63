// Shared pieces of the bf16x3-prefilter encode kernels (hsq_encode_pf.hip: d = 16; hsq_encode_pfd.hip:
// d = 8 and d = 32): bf16 hi/lo splitting, key operations, the exact rescoring of a codeword group and the
// end-of-kernel (lb, ub) fold.  Files that include this are compiled with -fno-honor-nans (build.py).
#pragma once
#include "hsq_encode_common.hpp"

namespace gq {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr unsigned KEY_MASK = 0x7FFFFFE0u;  // sign + 5 low mantissa bits make room for the group id (2^-18 relative)
constexpr float ERR_SCALE = 1.0025f * 3.0517578125e-05f;  // 2^-15 (x ||c||_1 x max|v_j|), analytic bound ~0.44 of it

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two f32 -> packed bf16 pair (round to nearest even), one VALU op on gfx950
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// x = hi + lo + O(2^-18 x): hi = bf16(x), lo = bf16(x - hi)   (x - hi is exact in f32)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16);
    const float r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    lo = cvt_pk_bf16(r0, r1);
}
// 8 consecutive floats -> the hi and lo bf16x8 MFMA fragments
__device__ __forceinline__ void split8(const f32x4 &q0, const f32x4 &q1, bf16x8 &hi, bf16x8 &lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair(q0[0], q0[1], h0, l0);
    split_pair(q0[2], q0[3], h1, l1);
    split_pair(q1[0], q1[1], h2, l2);
    split_pair(q1[2], q1[3], h3, l3);
    const u32x4 H = {h0, h1, h2, h3}, L = {l0, l1, l2, l3};
    hi = __builtin_bit_cast(bf16x8, H);
    lo = __builtin_bit_cast(bf16x8, L);
}

// ---- f16 operands of the d = 16 kernel (hsq_encode_pf.hip): two MFMAs per chain -------------------------------
// Facts checked on the device by tools/f16_probe.hip (profiles/r04_f16_probe.txt): v_mfma_f32_32x32x16_f16 keeps
// subnormal f16 inputs; v_fma_mixlo_f16 / v_fma_mixhi_f16 of (x, 2^s, 0) and v_cvt_pk_f16_f32 round to nearest even;
// v_dot2_f32_f16(a, a, acc) is within 2^-23 of the sum of squares.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// 8 consecutive floats of a codebook row -> hi and lo f16 fragments: c = hi + lo + r, |r| <= 2^-22 |c| + 2^-25
// (c - hi is exact in f32; the second term is the subnormal grid of f16 and is what is left for |c| < 2^-3)
__device__ __forceinline__ void split8_f16(const f32x4 &q0, const f32x4 &q1, half8 &hi, half8 &lo) {
    const float x[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (_Float16)x[i];
        lo[i] = (_Float16)(x[i] - (float)hi[i]);
    }
}
// 8 consecutive floats of a subvector, times the wave's power of two `scale` (an SGPR), -> ONE f16 fragment: one
// v_fma_mix per element (the product is exact in f32, the conversion rounds once), no lo part.
// n2 += ||vh||_2^2 of the eight values (v_dot2_f32_f16 per pair): the rounding error of element j is at most half an ulp
// <= 2^-11 |vh_j| (subnormal or zero results: 2^-25, counted separately), so 2^-11 sqrt(n2) bounds the L2 norm of the tile's
// rounding error, and (1 + 2^-11) sqrt(n2) bounds ||v'||_2 for the codebook-rounding term.  (Rounds 4-5 summed 4^exponent
// instead -- one v_and_b32 per pair for a first term up to 2x tighter and a second term 2x looser: 23,621 unsettled subvectors
// per 25 M-element launch against 22,508 with the plain norm, profiles/r06_encode_ab.txt block I.)
__device__ __forceinline__ void scale8_f16(const f32x4 &q0, const f32x4 &q1, float scale, half8 &hi, float &n2) {
    u32x4 H = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = i < 2 ? q0[2 * i] : q1[2 * i - 4], b = i < 2 ? q0[2 * i + 1] : q1[2 * i - 3];
        unsigned r;   // the lo form writes bits 0-15 and leaves the rest, the hi form then fills bits 16-31
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(scale));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r) : "v"(b), "s"(scale));
        H[i] = r;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(n2) : "v"(H[i]));
    hi = __builtin_bit_cast(half8, H);
}

__device__ __forceinline__ unsigned and_or(unsigned x, unsigned mask, unsigned c) { return (x & mask) | c; }
// one VALU op each (hipcc does not reliably form these from min/max compositions)
__device__ __forceinline__ unsigned max3u(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned med3u(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// max(|a|,|b|,|c|,|d|) straight on MFMA results.  Plain C so that hipcc inserts the MFMA -> VALU
// wait states itself (it pads nothing around inline asm: reading an accumulator from an asm
// statement returned stale scores).  This file is compiled with -fno-honor-nans (build.py):
// without it every fmaxf on an MFMA output gets a NaN-canonicalising v_max_f32 x,x,x in front
// (6 VALU ops per group instead of 2).  NaN gradients are undefined input either way.
__device__ __forceinline__ float absmax4(float a, float b, float c, float d) {
    return fmaxf(fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)), fabsf(d));   // v_max3_f32 + v_max_f32
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Exact reference arithmetic for FOUR consecutive codewords at once: p = fmaf chain over j
// ascending, from +0.  `quad` points at the group-interleaved LDS image: (c0[j], c1[j], c2[j], c3[j]) for
// j = 0..D-1.  Four plain v_fma_f32 per element, pinned by inline assembly: written as vector code (or as
// scalar fmaf, which the compiler packs again) this became v_pk_fma_f32 whose destination pair overlapped
// its multiplicand pair under op_sel (v_pk_fma v[2:3], v[188:189], v[2:3], v[6:7] op_sel:[0,1,0]) in the
// register-tight error-feedback instantiation, and there ONE result in ~1e5 -- always the low half, always
// lanes 48-63 -- came out with a wrong term (found by tools/fuzz_batched.py; codes right, u off by 1e-3
// relative).  The plain form is also faster beside the MFMAs (encode 52-54 -> 49 us).
#ifndef GQ_RESCORE_BATCH
#define GQ_RESCORE_BATCH 8
#endif
template <int D>
__device__ __forceinline__ f32x4 exact_score_quad(const float *__restrict__ quad, const float (&v)[D]) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    // the rows are fetched in batches of 8 (the accumulators of the prefilter are dead by now, there is room): with
    // the reads issued one by one beside their FMAs the group cost 16 LDS round trips in a row
    constexpr int B = D < GQ_RESCORE_BATCH ? D : GQ_RESCORE_BATCH;
#pragma unroll
    for (int j0 = 0; j0 < D; j0 += B) {
        f32x4 c[B];
#pragma unroll
        for (int jj = 0; jj < B; ++jj) c[jj] = *reinterpret_cast<const f32x4 *>(quad + 4 * (j0 + jj));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < B; ++jj) {
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(c[jj][0]), "v"(v[j0 + jj]));
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(c[jj][1]), "v"(v[j0 + jj]));
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(c[jj][2]), "v"(v[j0 + jj]));
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(c[jj][3]), "v"(v[j0 + jj]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const f32x4 r = {a0, a1, a2, a3};
    return r;
}

// order-preserving float -> uint32 map for integer atomic min/max
__device__ __forceinline__ unsigned order_map(float f) {
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float order_unmap(unsigned m) {
    return __uint_as_float(m ^ ((m >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

}  // namespace gq
