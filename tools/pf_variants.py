"""Diagnostic / experimental builds of the d16 prefilter encode (hsq_encode_pf.hip) for tools/ab_time.py -- never shipped:
    python tools/pf_variants.py nofix zb4 zblo m32 m32ns        # -> tools/exp/libgq_<name>.so each
  nofix   the product with the exact fix-up switched off (the baseline of the diagnostic builds below: wrong for ~5e-4 of the subvectors)
  zb4     DVFS diagnostic: every prefilter MFMA's B operand is one of four zero vectors the compiler cannot see through (the tile's
          bf16 split is kept alive): same instruction stream, the matrix pipe multiplies zeros
  zblo    only the lo part of the tile (the ch x vl MFMA of each chain) is zero
  m32     two MFMAs per chain (ch x vl dropped; vl still computed): the issue cost of 16 MFMAs per tile
  m32ns   ... and the lo part of the tile not computed at all (what a two-term prefilter would issue)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "gradient-quantization_amd", "csrc", "hsq_encode_pf.hip")).read()


def rep(s, old, new, count=1):
    assert s.count(old) >= 1, old
    return s.replace(old, new, count)


def nofix(s):
    return rep(s, "uint64_t todo = __ballot(valid && !safe);", "uint64_t todo = 0; asm volatile(\"\" :: \"s\"(__ballot(valid && !safe)));")


def keep_alive(s):
    return rep(s, "        f32x16 acc = {0};\n",
               "        asm volatile(\"\" :: \"v\"(vh[0]), \"v\"(vl[0]), \"v\"(vh[1]), \"v\"(vl[1]));\n        f32x16 acc = {0};\n")


def zeros(s, which):
    decl = "        bf16x8 zvh[2], zvl[2];\n"
    for b in (0, 1):
        for n in ("zvh", "zvl"):
            decl += "        %s[%d] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; asm volatile(\"\" : \"+v\"(%s[%d]));\n" % (n, b, n, b)
    s = rep(s, "        f32x16 acc = {0};\n", decl + "        f32x16 acc = {0};\n")
    n = 0
    for v in which:
        s, k = re.subn(r"(__builtin_amdgcn_mfma_f32_32x32x16_bf16\(c[hl]\[\w+\], )%s\[(\w+)\]" % v, r"\1z%s[\2]" % v, s)
        n += k
    assert n == 3 * len(which) - (len(which) - 1) * 0 or n > 0
    return s


def m32(s):
    s, k = re.subn(r"\n\s*n?acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16\(ch\[\w+\], vl\[\w+\], n?acc, 0, 0, 0\);", "", s)
    assert k == 2, k
    return s


def nosplit_lo(s):
    # the lo halves become copies of the hi halves' registers' worth of nothing: never computed
    s = rep(s, "        for (int blk = 0; blk < 2; ++blk) split8(nxt[2 * blk], nxt[2 * blk + 1], nvh[blk], nvl[blk]);",
            "        for (int blk = 0; blk < 2; ++blk) { split8hi(nxt[2 * blk], nxt[2 * blk + 1], nvh[blk]); nvl[blk] = nvh[blk]; }")
    s = rep(s, "namespace gq {\n", "namespace gq {\n__device__ __forceinline__ void split8hi(const f32x4 &q0, const f32x4 &q1, bf16x8 &hi) {\n"
            "    const u32x4 H = {cvt_pk_bf16(q0[0], q0[1]), cvt_pk_bf16(q0[2], q0[3]), cvt_pk_bf16(q1[0], q1[1]), cvt_pk_bf16(q1[2], q1[3])};\n"
            "    hi = __builtin_bit_cast(bf16x8, H);\n}\n")
    return s


def hold(s):
    """One operand held between consecutive MFMAs: even chains (cl,vh)(ch,vh)(ch,vl), odd chains (ch,vl)(ch,vh)(cl,vh)."""
    s = rep(s, "        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[0], vl[0], acc, 0, 0, 0);\n        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[0], vh[0], acc, 0, 0, 0);\n",
            "        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[0], vh[0], acc, 0, 0, 0);\n        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[0], vl[0], acc, 0, 0, 0);\n")
    s = rep(s, "                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[nr], vh[nb], nacc, 0, 0, 0);\n",
            "                if ((c + 1) & 1) nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vl[nb], nacc, 0, 0, 0);\n"
            "                else nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[nr], vh[nb], nacc, 0, 0, 0);\n")
    s = rep(s, "                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vl[nb], nacc, 0, 0, 0);\n                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vh[nb], nacc, 0, 0, 0);\n",
            "                nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vh[nb], nacc, 0, 0, 0);\n"
            "                if ((c + 1) & 1) nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[nr], vh[nb], nacc, 0, 0, 0);\n"
            "                else nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[nr], vl[nb], nacc, 0, 0, 0);\n")
    return s


def h32(s):
    """m32ns with f16 operands (v_mfma_f32_32x32x16_f16, v_cvt_pk_f16_f32; no scaling: for N(0,1) test data only): the
    clock and the time of a two-MFMA f16 prefilter, answers unchecked."""
    s = nosplit_lo(m32(nofix(s)))
    s = rep(s, "namespace gq {\n", "namespace gq {\ntypedef _Float16 half8_ __attribute__((ext_vector_type(8)));\n"
            "__device__ __forceinline__ unsigned cvt_pk_f16_(float lo, float hi) { unsigned r; asm(\"v_cvt_pk_f16_f32 %0, %1, %2\" : \"=v\"(r) : \"v\"(lo), \"v\"(hi)); return r; }\n"
            "__device__ __forceinline__ void split8h_(const f32x4 &q0, const f32x4 &q1, bf16x8 &hi, bf16x8 &lo) {\n"
            "    half8_ h, l; float x[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};\n"
            "    for (int i = 0; i < 8; ++i) { h[i] = (_Float16)x[i]; l[i] = (_Float16)(x[i] - (float)h[i]); }\n"
            "    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);\n}\n")
    s = rep(s, "        split8(q0, q1, fh, fl);\n", "        split8h_(q0, q1, fh, fl);\n")
    s = s.replace("const u32x4 H = {cvt_pk_bf16(q0[0], q0[1]), cvt_pk_bf16(q0[2], q0[3]), cvt_pk_bf16(q1[0], q1[1]), cvt_pk_bf16(q1[2], q1[3])};",
                  "const u32x4 H = {cvt_pk_f16_(q0[0], q0[1]), cvt_pk_f16_(q0[2], q0[3]), cvt_pk_f16_(q1[0], q1[1]), cvt_pk_f16_(q1[2], q1[3])};")
    # the first tile's split (prologue of the loop) also goes through the hi-only f16 form
    s = rep(s, "        for (int blk = 0; blk < 2; ++blk) split8(cur[2 * blk], cur[2 * blk + 1], vh[blk], vl[blk]);",
            "        for (int blk = 0; blk < 2; ++blk) { split8hi(cur[2 * blk], cur[2 * blk + 1], vh[blk]); vl[blk] = vh[blk]; }")
    s, k = re.subn(r"__builtin_amdgcn_mfma_f32_32x32x16_bf16\((c[hl]\[\w+\]), (v[hl]\[\w+\]), ",
                   r"__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_, \1), __builtin_bit_cast(half8_, \2), ", s)
    assert k == 4, k
    return s


def noqueue(s):
    """Round 4's kernel with nothing ever flagged (no queueing, no exact scans): the floor of its tile loop."""
    return rep(s, "        const bool flagged = valid && !safe;\n",
               "        const bool flagged = false;\n        asm volatile(\"\" :: \"s\"(__ballot(valid && !safe)));\n")


def zb2(s):
    """DVFS diagnostic for round 4's kernel: nothing flagged, and the B operand of every prefilter MFMA one of two zero
    vectors the compiler cannot see through (the tile's conversion kept alive)."""
    s = noqueue(s)
    s = rep(s, "        f32x16 acc = {0};\n",
            "        half8 zvh[2];\n        zvh[0] = half8{0, 0, 0, 0, 0, 0, 0, 0}; asm volatile(\"\" : \"+v\"(zvh[0]));\n"
            "        zvh[1] = half8{0, 0, 0, 0, 0, 0, 0, 0}; asm volatile(\"\" : \"+v\"(zvh[1]));\n"
            "        asm volatile(\"\" :: \"v\"(vh[0]), \"v\"(vh[1]));\n        f32x16 acc = {0};\n")
    s, k = re.subn(r"(__builtin_amdgcn_mfma_f32_32x32x16_f16\(c[hl]\[\w+\], )vh\[(\w+)\]", r"\1zvh[\2]", s)
    assert k == 4, k
    return s


def fp8lo(s):
    """Timing probe: the lo MFMA of every chain (cl x vh) as v_mfma_f32_32x32x16_fp8_fp8 on two registers of each operand
    taken as they are (random bits), nothing flagged: what an 8-bit lo term would cost in time and clock.  Answers unchecked."""
    s = noqueue(s)
    s, k = re.subn(r"__builtin_amdgcn_mfma_f32_32x32x16_f16\(cl\[(\w+)\], vh\[(\w+)\], ",
                   r"__builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, __builtin_shufflevector(__builtin_bit_cast(u32x4, cl[\1]), __builtin_bit_cast(u32x4, cl[\1]), 0, 1)), "
                   r"__builtin_bit_cast(long, __builtin_shufflevector(__builtin_bit_cast(u32x4, vh[\2]), __builtin_bit_cast(u32x4, vh[\2]), 0, 1)), ", s)
    assert k == 2, k
    return s


def m1(s):
    """Timing probe: ONE MFMA per chain (ch x vh), nothing flagged."""
    s = noqueue(s)
    s, k = re.subn(r"\n\s*n?acc = __builtin_amdgcn_mfma_f32_32x32x16_f16\(cl\[\w+\], vh\[\w+\], n?acc, 0, 0, 0\);", "", s)
    assert k == 2, k
    return s


def fp8lo2(s):
    """Timing probe closer to the real thing: the lo MFMA (cl x vh) as v_mfma_f32_32x32x16_fp8_fp8 on cl * 2^16 converted to
    fp8 in the prologue and on the tile's f16 fragment converted by v_cvt_scalef32_pk_fp8_f16 (eight operations per tile and
    lane); nothing flagged.  Scale semantics unpinned (scale 1.0), answers unchecked."""
    s = noqueue(s)
    s = rep(s, "    unsigned c2b = __float_as_uint(s_c1[0]);",
            "    long cl8[8];\n    for (int rb = 0; rb < 8; ++rb) {\n        int w0 = 0, w1 = 0;\n"
            "        w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)cl[rb][0] * 65536.0f, (float)cl[rb][1] * 65536.0f, w0, false);\n"
            "        w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)cl[rb][2] * 65536.0f, (float)cl[rb][3] * 65536.0f, w0, true);\n"
            "        w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)cl[rb][4] * 65536.0f, (float)cl[rb][5] * 65536.0f, w1, false);\n"
            "        w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)cl[rb][6] * 65536.0f, (float)cl[rb][7] * 65536.0f, w1, true);\n"
            "        cl8[rb] = (long)(((unsigned long long)(unsigned)w1 << 32) | (unsigned)w0);\n    }\n"
            "    unsigned c2b = __float_as_uint(s_c1[0]);")
    s = rep(s, "        f32x16 acc = {0};\n",
            "        long vb8[2];\n        for (int blk = 0; blk < 2; ++blk) {\n            const u32x4 H = __builtin_bit_cast(u32x4, vh[blk]);\n            unsigned w0 = 0, w1 = 0;\n"
            "            asm(\"v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0\" : \"+v\"(w0) : \"v\"(H[0]));\n"
            "            asm(\"v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0 op_sel:[0,0,1]\" : \"+v\"(w0) : \"v\"(H[1]));\n"
            "            asm(\"v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0\" : \"+v\"(w1) : \"v\"(H[2]));\n"
            "            asm(\"v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0 op_sel:[0,0,1]\" : \"+v\"(w1) : \"v\"(H[3]));\n"
            "            vb8[blk] = (long)(((unsigned long long)w1 << 32) | w0);\n        }\n        f32x16 acc = {0};\n")
    s, k = re.subn(r"__builtin_amdgcn_mfma_f32_32x32x16_f16\(cl\[(\w+)\], vh\[(\w+)\], ", r"__builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(cl8[\1], vb8[\2], ", s)
    assert k == 2, k
    return s


# ---- round 5: where the ~5 us between `noqueue` and the product go (the safety arithmetic stays alive in all of these) ----
def nowr(s):
    """Diagnostic (wrong for ~1.5 % of the subvectors): `safe` is computed and kept alive, but nothing is queued and every
    valid lane stores its rescored candidate: the product minus queue writes, second pass and exact scans."""
    s = rep(s, "        uint64_t todo = __ballot(flagged);\n", "        uint64_t todo = 0; asm volatile(\"\" :: \"s\"(__ballot(flagged)));\n")
    return rep(s, "        if (valid && safe) {   // uniform bases", "        if (valid) {   // uniform bases")


def nopass(s):
    """Diagnostic (wrong for the queued subvectors): entries are queued as in the product, but the ring is dropped instead of
    going through the second pass: the product minus second pass and exact scans (queue writes stay)."""
    s = rep(s, "                second_pass(qhead, 32);\n", "                asm volatile(\"\" :: \"s\"(qhead));\n")
    return rep(s, "        second_pass(qhead, n);\n", "        asm volatile(\"\" :: \"s\"(qhead), \"s\"(n));\n")


def r16(s, hdr=None):
    """Rescoring rows fetched as ONE batch of 16 ds_read_b128 instead of two of 8 (one LDS round trip per tile less; +32 VGPRs)."""
    return s      # (the edit is in hsq_pf_common.hpp: built with -DGQ_RESCORE_BATCH=16)


# ---- round 5, second half: the verdict's (c) and (d), and the gather split over two memory pipes ----
def prep2x(s):
    """Timing probe for a per-codebook prepared image (verdict item 1c): the prologue's f16 split and norm measurement done
    TWICE (the second time on copies the compiler cannot see through; same answers) -- the time this adds is what a prologue
    that only loads prepared fragments (same bytes, same barrier) would save."""
    s = rep(s, "                fl[e] = (_Float16)rr;   // c = hi + lo + r",
            "                fl[e] = (_Float16)rr;\n"
            "                { float x2 = x; asm volatile(\"\" : \"+v\"(x2)); const _Float16 h2 = (_Float16)x2; const float r2 = x2 - (float)h2; const _Float16 l2h = (_Float16)r2;\n"
            "                  float t2 = __fmaf_rn(x2, x2, l2); float u2 = __fmaf_rn(r2, r2, d2); asm volatile(\"\" :: \"v\"(t2), \"v\"(u2));\n"
            "                  fh[e] = h2; fl[e] = l2h; }   // c = hi + lo + r")
    s = rep(s, "        l2 = wave_max_nan(l2);\n        d2 = wave_max_nan(d2);\n",
            "        { float a2 = l2, b2 = d2; asm volatile(\"\" : \"+v\"(a2), \"+v\"(b2)); a2 = wave_max_nan(a2); b2 = wave_max_nan(b2); asm volatile(\"\" :: \"v\"(a2), \"v\"(b2)); }\n"
            "        l2 = wave_max_nan(l2);\n        d2 = wave_max_nan(d2);\n")
    return s


def twophase(s):
    """Verdict item 1d: the f32 codebook image twice in LDS, the second copy 8 banks (two 16-byte bank groups) further on;
    lanes 8-15 of every 16 gather from the second.  Bit-identical answers."""
    s = rep(s, "    __shared__ __attribute__((aligned(16))) float s_cb[64 * QS];",
            "    __shared__ __attribute__((aligned(16))) float s_cb[64 * QS];\n    __shared__ __attribute__((aligned(16))) float s_cb2[64 * QS + 8];")
    s = rep(s, "        s_cb[(k >> 2) * QS + 4 * jj + (k & 3)] = cbv[n];",
            "        s_cb[(k >> 2) * QS + 4 * jj + (k & 3)] = cbv[n];\n        s_cb2[8 + (k >> 2) * QS + 4 * jj + (k & 3)] = cbv[n];")
    s = rep(s, "        const f32x4 p4 = exact_score_quad<D>(s_cb + (kc >> 2) * QS, vf);   // kc is a multiple of 4: one quad",
            "        const f32x4 p4 = exact_score_quad<D>(((lane & 8) ? s_cb2 + 8 : s_cb) + (kc >> 2) * QS, vf);")
    return s


def l1half(s):
    """Timing probe: the second half of the rescoring gather (8 of the 16 rows of a group) comes through the vector L1 from
    global memory instead of LDS (same 16-byte-per-lane pattern on the caller's codebook: values wrong) -- would two memory
    pipes instead of one shorten the gather?"""
    s = rep(s, "namespace gq {\n", """namespace gq {
template <int D>
__device__ __forceinline__ f32x4 exact_score_quad_l1(const float *__restrict__ quad, const float *__restrict__ gq_, const float (&v)[D]) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    f32x4 c[D];
#pragma unroll
    for (int jj = D / 2; jj < D; ++jj) c[jj] = *reinterpret_cast<const f32x4 *>(gq_ + 4 * jj);
#pragma unroll
    for (int jj = 0; jj < D / 2; ++jj) c[jj] = *reinterpret_cast<const f32x4 *>(quad + 4 * jj);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < D; ++jj) {
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(c[jj][0]), "v"(v[jj]));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(c[jj][1]), "v"(v[jj]));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(c[jj][2]), "v"(v[jj]));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(c[jj][3]), "v"(v[jj]));
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 r = {a0, a1, a2, a3};
    return r;
}
""")
    s = rep(s, "        const f32x4 p4 = exact_score_quad<D>(s_cb + (kc >> 2) * QS, vf);   // kc is a multiple of 4: one quad",
            "        const f32x4 p4 = exact_score_quad_l1<D>(s_cb + (kc >> 2) * QS, cb + (kc >> 2) * 4 * D, vf);")
    return s


def w12(s, waves=12, aregs=False):
    """Three waves per SIMD for d = 16: workgroups of 12 waves (768 threads, <= 168 VGPRs), the codebook's f16 A fragments read
    from LDS per chain as d = 32 does (-32 VGPRs) -- does a third wave fill the loop's stalls (5,084 cycles per tile against
    4,256 of issue)?  Bit-identical answers."""
    s = rep(s, "constexpr int PF_WAVES = 8;", "constexpr int PF_WAVES = %d;" % waves)
    if not aregs:
        s = rep(s, "static constexpr bool A_REGS = D <= 16;", "static constexpr bool A_REGS = D < 16;")
    s = rep(s, "    float cbv[256 * D / PF_THREADS];\n#pragma unroll\n    for (int n = 0; n < 256 * D / PF_THREADS; ++n) cbv[n] = cb[threadIdx.x + n * PF_THREADS];",
            "    constexpr int NCB = (256 * D + PF_THREADS - 1) / PF_THREADS;\n    float cbv[NCB];\n#pragma unroll\n    for (int n = 0; n < NCB; ++n) cbv[n] = (threadIdx.x + n * PF_THREADS < 256 * D) ? cb[threadIdx.x + n * PF_THREADS] : 0.0f;")
    s = rep(s, "    for (int n = 0; n < 256 * D / PF_THREADS; ++n) {\n        const int i = threadIdx.x + n * PF_THREADS, k = i / D, jj = i % D;\n        s_cb[(k >> 2) * QS + 4 * jj + (k & 3)] = cbv[n];",
            "    for (int n = 0; n < NCB; ++n) {\n        const int i = threadIdx.x + n * PF_THREADS, k = i / D, jj = i % D;\n        if (i < 256 * D) s_cb[(k >> 2) * QS + 4 * jj + (k & 3)] = cbv[n];")
    s = rep(s, "    if (loads_here) {\n#pragma unroll\n        for (int s = 0; s < KS; ++s) {\n            aq[2 * s] =", "    if (loads_here && wave < 8) {\n#pragma unroll\n        for (int s = 0; s < KS; ++s) {\n            aq[2 * s] =")
    s = rep(s, "        float l2 = 0.0f, d2 = 0.0f;\n#pragma unroll\n        for (int s = 0; s < KS; ++s) {\n            half8 fh, fl;", "        float l2 = 0.0f, d2 = 0.0f;\n        if (wave < 8) {\n#pragma unroll\n        for (int s = 0; s < KS; ++s) {\n            half8 fh, fl;")
    s = rep(s, "        if (lane == 0) {\n            s_c1[wave] = l2;\n            s_dc[wave] = d2;\n        }\n", "        if (lane == 0) {\n            s_c1[wave] = l2;\n            s_dc[wave] = d2;\n        }\n        }\n")
    s = rep(s, "    for (int w = 1; w < PF_WAVES; ++w) {\n        c2b = max(c2b", "    for (int w = 1; w < 8; ++w) {\n        c2b = max(c2b")
    return s


def counts(s):
    """Diagnostic: every wave adds its second-pass entries, exact scans, passes and tiles to four words behind the workspace's
    (min, max) pairs (flat AND multi-tensor form) -- how often do the rare paths run?  Built with -DGQ_PF_STAMPS."""
    s = rep(s, "        GQ_STAMPS_ONLY(npassed += n;)", "        GQ_STAMPS_ONLY(npassed += n; npasses += 1;)")
    s = rep(s, "unsigned long long nscanned = 0, npassed = 0;)", "unsigned long long nscanned = 0, npassed = 0, npasses = 0;)")
    s = rep(s, "#ifdef GQ_PF_STAMPS\n    if (!BATCHED && lane == 0 && blockIdx.x < 256) {",
            "#ifdef GQ_PF_STAMPS\n    if (lane == 0) {\n        int *cnt_ = reinterpret_cast<int *>(ws) + 2 * GQ_MAX_PARTIALS + 4;   // (the first words of the fix-up log: unused by this kernel)\n"
            "        atomicAdd(cnt_, (int)npassed); atomicAdd(cnt_ + 1, (int)nscanned); atomicAdd(cnt_ + 2, (int)npasses); atomicAdd(cnt_ + 3, (int)ntl);\n    }\n"
            "    if (!BATCHED && lane == 0 && blockIdx.x < 256) {")
    return s


VARIANTS = {
    "counts": counts,
    "w12": w12,
    "w12a": lambda s: w12(s, 12, True),
    "w16": lambda s: w12(s, 16, False),
    "prep2x": prep2x,
    "twophase": twophase,
    "l1half": l1half,
    "nowr": nowr,
    "nopass": nopass,
    "fp8lo2": fp8lo2,
    "m1": m1,
    "fp8lo": fp8lo,
    "zb2": zb2,
    "noqueue": noqueue,
    "h32": h32,
    "hold": hold,
    "nofix": lambda s: nofix(s),
    "zb4": lambda s: zeros(keep_alive(nofix(s)), ("vh", "vl")),
    "zblo": lambda s: zeros(keep_alive(nofix(s)), ("vl",)),
    "m32": lambda s: keep_alive(m32(nofix(s))),
    "m32ns": lambda s: nosplit_lo(m32(nofix(s))),
}

# variants that are a compile-time switch of the shipped source: name -> extra hipcc flags (the source is taken as it is)
EXTRA_FLAGS = {"counts": ["-DGQ_PF_STAMPS"], "skew0": ["-DGQ_PF_SKEW_PERMILLE=0"], "skew5": ["-DGQ_PF_SKEW_PERMILLE=5"], "skew20": ["-DGQ_PF_SKEW_PERMILLE=20"], "skew30": ["-DGQ_PF_SKEW_PERMILLE=30"], "r16": ["-DGQ_RESCORE_BATCH=16"],
               # round 5's early ring flush (hsq_encode_pf.hip, PF_FLUSH_AHEAD): off / other distances from the end of the run
               "nopair": ["-DGQ_PF_PAIR=0"], "pair": ["-DGQ_PF_PAIR=1"], "tail0": ["-DGQ_PF_TAIL=0"], "tail1": ["-DGQ_PF_TAIL=1"], "tail2": ["-DGQ_PF_TAIL=2"], "tail3": ["-DGQ_PF_TAIL=3"], "tail6": ["-DGQ_PF_TAIL=6"], "fa16": ["-DGQ_PF_FLUSH_AHEAD=16"],
               "fa0": ["-DGQ_PF_FLUSH_AHEAD=0"], "fa8": ["-DGQ_PF_FLUSH_AHEAD=8"], "fa12": ["-DGQ_PF_FLUSH_AHEAD=12"],
               "fa24": ["-DGQ_PF_FLUSH_AHEAD=24"], "fa32": ["-DGQ_PF_FLUSH_AHEAD=32"],
               "fa16s2": ["-DGQ_PF_SCAN1_MAX=2"], "fa16s8": ["-DGQ_PF_SCAN1_MAX=8"], "fa16m1": ["-DGQ_PF_FLUSH_MIN=1"]}
for _n in EXTRA_FLAGS:
    VARIANTS.setdefault(_n, lambda s: s)

if __name__ == "__main__":
    os.makedirs("/tmp/gq_pfv", exist_ok=True)
    for name in sys.argv[1:]:
        src = "/tmp/gq_pfv/hsq_encode_pf_%s.hip" % name
        open(src, "w").write(VARIANTS[name](SRC))
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), os.path.join(ROOT, "tools", "exp", "libgq_%s.so" % name), src]
                              + EXTRA_FLAGS.get(name, []), stderr=subprocess.DEVNULL)
