/*
 * gq_oracle.c -- CPU restatement of the reference's HSQ / QSGD gradient
 * quantisation hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a plain-C restatement of what the reference
 * (xinyandai/gradient-quantization, Python/PyTorch) computes on this path.  It
 * is NOT part of the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (gradient-quantization_amd/)
 * never links, imports or calls anything in oracle/ and fails loudly when its
 * HIP library is missing.
 *
 * Parity pin: every function here is checked against the golden vectors under
 * tests/golden/, which were produced by importing the reference itself
 * (tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
 *
 * Arithmetic contract (why this is bit-exact with torch.mm on CPU):
 * the reference's inner products run in MKL sgemm with K = c_dim; that result
 * is bit-identical to a single-accumulator ascending chain
 *     acc = 0; for j = 0..d-1: acc = fmaf(c[j], v[j], acc)
 * (re-verified by the golden tests).  Compile with
 * -ffp-contract=off so that nothing else is fused.
 *
 * -mavx2 -mfma: fmaf() becomes one vfmadd instruction (the same correctly rounded fused operation as libm's fmaf,
 * without the PLT call per multiply-add that a baseline x86-64 build makes).
 *
 * Build:  gcc -O2 -mavx2 -mfma -fPIC -shared -fopenmp -ffp-contract=off -o libgq_oracle.so gq_oracle.c gq_cpu.c -lm
 */
#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GQ_EXPORT __attribute__((visibility("default")))

/* A thread that has run AVX / AVX-512 code without a closing vzeroupper keeps the upper halves of its vector registers
 * dirty, and every legacy-SSE instruction it executes afterwards (this file's scalar loops are built for baseline x86-64)
 * pays a merge with that state.  torch's CPU kernels do leave such threads behind, and they share the OpenMP pool with this
 * library: measured after one strided torch.mean, ONE worker of the team ran its share 60x slower (6.9 s against 0.12 s for
 * the same slice), the parallel region waited for it, and a pytest run of the whole CPU suite took 11 minutes instead of 3.
 * Every exported function and every thread of every parallel region therefore starts with vzeroupper. */
static inline void gq_clean_vector_state(void) {
#if defined(__x86_64__)
    /* the clobber list matters since this file holds 256-bit values of its own: without it the compiler may keep a
     * ymm constant live across the instruction and find its upper half zeroed */
    if (__builtin_cpu_supports("avx"))
        __asm__ volatile("vzeroupper" ::: "memory", "xmm0", "xmm1", "xmm2", "xmm3", "xmm4", "xmm5", "xmm6", "xmm7", "xmm8",
                         "xmm9", "xmm10", "xmm11", "xmm12", "xmm13", "xmm14", "xmm15");
#endif
}

GQ_EXPORT int gq_oracle_num_threads(void) {
    gq_clean_vector_state();
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

GQ_EXPORT void gq_oracle_set_num_threads(int n) {
    gq_clean_vector_state();
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/*
 * HSQ encode.  Follows compressors/nearest_neighbor_compressor.py:63-73:
 *   p     = mm(codewords, vec.view(-1,d).T).T         (:68)
 *   codes = argmax(|p|, dim=1)   -- first maximum     (:69,:72)
 *   u     = p.gather(1, codes)   -- SIGNED projection (:73)
 * grad: [M*d] f32, codebook: [K*d] f32 row-major (already row-normalised),
 * codes: [M] int32, u: [M] f32.
 *
 * hsq_encode_one_scalar is the literal restatement (one subvector, codewords in turn).  gq_oracle_hsq_encode runs the
 * SAME arithmetic eight codewords at a time: lane i of a 256-bit register carries codeword kb+i's own accumulator
 * through the same ascending chain acc = fma(c[j], v[j], acc), j = 0..d-1 (vfmadd231ps is the fused operation fmaf()
 * is; a lane never sees another lane's value), four such registers per pass so that one broadcast of v[j] feeds 32
 * chains.  The first-maximum rule is kept per lane by a strict `>` over ascending blocks and across lanes by taking
 * the lowest index among equal maxima.  A subvector with any NaN score (NaN or infinite input) is redone by the
 * scalar form, whose NaN ranking is torch.argmax's.  tests/test_oracle_golden.py holds both forms to the reference's
 * fixtures bit for bit; gq_oracle_hsq_encode_scalar stays exported for that comparison.
 */
static inline void hsq_encode_one_scalar(const float *v, const float *codebook, int d, int K, int32_t *code, float *u) {
    float best_abs = -1.0f, best_p = 0.0f;
    int32_t best_k = 0;
    for (int k = 0; k < K; ++k) {
        const float *c = codebook + (int64_t)k * d;
        float acc = 0.0f;
        for (int j = 0; j < d; ++j) acc = fmaf(c[j], v[j], acc);
        float a = fabsf(acc);
        /* torch.argmax: first index of the maximum; a NaN counts as the maximum */
        if (a > best_abs || (isnan(a) && !isnan(best_abs))) {
            best_abs = a;
            best_p = acc;
            best_k = k;
        }
    }
    *code = best_k;
    *u = best_p;
}

GQ_EXPORT void gq_oracle_hsq_encode_scalar(const float *grad, const float *codebook, int64_t M, int d, int K,
                                           int32_t *codes, float *u) {
    gq_clean_vector_state();
#pragma omp parallel
    {
        gq_clean_vector_state();   /* per thread: see the comment at its definition */
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) hsq_encode_one_scalar(grad + m * (int64_t)d, codebook, d, K, codes + m, u + m);
    }
}

GQ_EXPORT void gq_oracle_hsq_encode(const float *grad, const float *codebook, int64_t M, int d, int K,
                                    int32_t *codes, float *u) {
    gq_clean_vector_state();
    /* transposed image [d][Kp], Kp = K rounded up to 32, padded with zero codewords: a padded lane scores +0 (or NaN
     * for a non-finite input, which sends the subvector to the scalar form) and its index is above every real one,
     * so it never wins a tie */
    const int Kp = (K + 31) & ~31;
    float *cT = (float *)aligned_alloc(32, (size_t)Kp * (size_t)d * sizeof(float));
    if (!cT) { gq_oracle_hsq_encode_scalar(grad, codebook, M, d, K, codes, u); return; }
    for (int j = 0; j < d; ++j)
        for (int k = 0; k < Kp; ++k) cT[(size_t)j * Kp + k] = k < K ? codebook[(int64_t)k * d + j] : 0.0f;
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff));
    const __m256i lane_id = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
#pragma omp parallel
    {
        gq_clean_vector_state();   /* per thread: see the comment at its definition */
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float *v = grad + m * (int64_t)d;
        __m256 best_abs = _mm256_set1_ps(-1.0f), best_p = _mm256_setzero_ps(), unord = _mm256_setzero_ps();
        __m256i best_k = _mm256_setzero_si256();
        for (int kb = 0; kb < Kp; kb += 32) {
            __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
            const float *c = cT + kb;
            for (int j = 0; j < d; ++j, c += Kp) {
                const __m256 vj = _mm256_broadcast_ss(v + j);
                a0 = _mm256_fmadd_ps(_mm256_load_ps(c), vj, a0);
                a1 = _mm256_fmadd_ps(_mm256_load_ps(c + 8), vj, a1);
                a2 = _mm256_fmadd_ps(_mm256_load_ps(c + 16), vj, a2);
                a3 = _mm256_fmadd_ps(_mm256_load_ps(c + 24), vj, a3);
            }
            const __m256 acc[4] = {a0, a1, a2, a3};
            for (int b = 0; b < 4; ++b) {
                const __m256 ab = _mm256_and_ps(acc[b], absmask);
                const __m256 gt = _mm256_cmp_ps(ab, best_abs, _CMP_GT_OQ);
                best_abs = _mm256_blendv_ps(best_abs, ab, gt);
                best_p = _mm256_blendv_ps(best_p, acc[b], gt);
                best_k = _mm256_castps_si256(_mm256_blendv_ps(_mm256_castsi256_ps(best_k),
                             _mm256_castsi256_ps(_mm256_add_epi32(lane_id, _mm256_set1_epi32(kb + 8 * b))), gt));
                unord = _mm256_or_ps(unord, _mm256_cmp_ps(acc[b], acc[b], _CMP_UNORD_Q));
            }
        }
        if (_mm256_movemask_ps(unord)) {
            hsq_encode_one_scalar(v, codebook, d, K, codes + m, u + m);
            continue;
        }
        float fa[8], fp[8];
        int32_t fk[8];
        _mm256_storeu_ps(fa, best_abs);
        _mm256_storeu_ps(fp, best_p);
        _mm256_storeu_si256((__m256i *)fk, best_k);
        int w = 0;
        for (int i = 1; i < 8; ++i)
            if (fa[i] > fa[w] || (fa[i] == fa[w] && fk[i] < fk[w])) w = i;
        codes[m] = fk[w];
        u[m] = fp[w];
    }
    }
    free(cT);
}

/* torch.min / torch.max over the whole tensor
 * (compressors/probabilistic_scalar_compressor.py:13-14).  lb_ub[0]=min, [1]=max. */
GQ_EXPORT void gq_oracle_minmax(const float *u, int64_t M, float *lb_ub) {
    gq_clean_vector_state();
    float lo = INFINITY, hi = -INFINITY;
    int has_nan = 0;
    for (int64_t i = 0; i < M; ++i) {
        float x = u[i];
        if (isnan(x)) has_nan = 1;
        if (x < lo) lo = x;
        if (x > hi) hi = x;
    }
    if (has_nan) lo = hi = NAN;
    lb_ub[0] = lo;
    lb_ub[1] = hi;
}

/*
 * Scalar (norm) quantiser.  Follows
 * compressors/probabilistic_scalar_compressor.py:12-27:
 *   if lb - ub == 0: levels = 0                                      (:15-16)
 *   x = |(u - lb) / (ub - lb)| * s                                   (:17)
 *   l = trunc(clamp(x, 0, s-1))                                      (:18)
 *   if random: l += (x - float(l) > r)        r = torch.rand(M)      (:20-26)
 * r may be NULL when random == 0.
 */
GQ_EXPORT void gq_oracle_scalar_levels(const float *u, int64_t M, int n_bit, int random, const float *r,
                                       float lb, float ub, int32_t *levels) {
    gq_clean_vector_state();
    const float s = (float)(1 << n_bit);
    if (lb - ub == 0.0f) {
        memset(levels, 0, (size_t)M * sizeof(int32_t));
        return;
    }
    const float range = ub - lb;
#pragma omp parallel
    {
        gq_clean_vector_state();   /* per thread: see the comment at its definition */
#pragma omp for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        float q = (u[i] - lb) / range;
        float x = fabsf(q) * s;
        float c = x < 0.0f ? 0.0f : (x > s - 1.0f ? s - 1.0f : x);
        int32_t l = isnan(c) ? INT_MIN : (int32_t)c;
        if (random) {
            float prob = x - (float)l;
            l += (prob > r[i]) ? 1 : 0;
        }
        levels[i] = l;
    }
    }
}

/* Scalar de-quantiser, probabilistic_scalar_compressor.py:29-33:
 *   n = float(l) * (ub - lb) / s + lb      -- mul, then /s, then add; nothing fused */
GQ_EXPORT void gq_oracle_scalar_decode(const int32_t *levels, int64_t M, int n_bit, float lb, float ub,
                                       float *norms) {
    gq_clean_vector_state();
    const float s = (float)(1 << n_bit);
    const float range = ub - lb;
    for (int64_t i = 0; i < M; ++i) {
        float t = (float)levels[i] * range;
        t = t / s;
        norms[i] = t + lb;
    }
}

/* HSQ decode, nearest_neighbor_compressor.py:85-90:
 *   out[m,:] = codewords[codes[m],:] * norms[m] */
GQ_EXPORT void gq_oracle_hsq_decode(const int32_t *codes, const float *norms, const float *codebook, int64_t M,
                                    int d, float *out) {
    gq_clean_vector_state();
#pragma omp parallel
    {
        gq_clean_vector_state();   /* per thread: see the comment at its definition */
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float *c = codebook + (int64_t)codes[m] * d;
        const float n = norms[m];
        float *o = out + m * (int64_t)d;
        for (int j = 0; j < d; ++j) o[j] = c[j] * n;
    }
    }
}

/*
 * Full HSQ compress in one call (what bench.py's cpu_baseline times):
 * encode + global min/max + levels.  Returns lb/ub through lb_ub[2].
 */
GQ_EXPORT void gq_oracle_hsq_compress(const float *grad, const float *codebook, int64_t M, int d, int K, int n_bit,
                                      int random, const float *r, int32_t *codes, float *u, float *lb_ub,
                                      int32_t *levels) {
    gq_clean_vector_state();
    gq_oracle_hsq_encode(grad, codebook, M, d, K, codes, u);
    gq_oracle_minmax(u, M, lb_ub);
    gq_oracle_scalar_levels(u, M, n_bit, random, r, lb_ub[0], lb_ub[1], levels);
}

/*
 * Parameter-server aggregate, quantizers/ps_quantizer.py:48:
 *   g = torch.stack(decoded_u, 0).mean(0)
 * decoded: [U][n] contiguous.  Sum in user order, then divide by U.
 */
GQ_EXPORT void gq_oracle_mean_users(const float *decoded, int U, int64_t n, float *out) {
    gq_clean_vector_state();
    for (int64_t i = 0; i < n; ++i) {
        float acc = 0.0f + decoded[i];   /* torch's sum starts from +0: an all -0 column comes out as +0 */
        for (int k = 1; k < U; ++k) acc += decoded[(int64_t)k * n + i];
        out[i] = acc / (float)U;
    }
}

/*
 * QSGD compress, compressors/qsgd_compressor.py:42-64 (bucket = `d` consecutive
 * elements of the flattened tensor):
 *   norm  = max |v| over the bucket                        (:49)
 *   x     = |v / norm| * s                                 (:50,:52)
 *   l     = trunc(clamp(x, 0, s-1))                        (:53)
 *   l    += (x - float(l) > r)        if random            (:55-61)
 *   signs = sign(v) > 0                                    (:63)
 * A zero bucket gives 0/0 = NaN -> the int32 cast of NaN is INT_MIN on the
 * reference's x86 CPU path; reproduced explicitly here.
 */
GQ_EXPORT void gq_oracle_qsgd_compress(const float *grad, int64_t Mb, int d, int n_bit, int random, const float *r,
                                       float *norm, uint8_t *signs, int32_t *levels) {
    gq_clean_vector_state();
    const float s = (float)(1 << n_bit);
#pragma omp parallel
    {
        gq_clean_vector_state();   /* per thread: see the comment at its definition */
#pragma omp for schedule(static)
    for (int64_t b = 0; b < Mb; ++b) {
        const float *v = grad + b * (int64_t)d;
        float mx = 0.0f;
        int has_nan = 0;
        for (int j = 0; j < d; ++j) {
            float a = fabsf(v[j]);
            if (isnan(a)) has_nan = 1;
            if (a > mx) mx = a;
        }
        if (has_nan) mx = NAN;
        norm[b] = mx;
        for (int j = 0; j < d; ++j) {
            int64_t i = b * (int64_t)d + j;
            float q = v[j] / mx;
            float x = fabsf(q) * s;
            float c = x < 0.0f ? 0.0f : (x > s - 1.0f ? s - 1.0f : x);
            int32_t l = isnan(c) ? INT_MIN : (int32_t)c;
            if (random) {
                float prob = x - (float)l;
                l += (prob > r[i]) ? 1 : 0;
            }
            levels[i] = l;
            signs[i] = v[j] > 0.0f ? 1 : 0;
        }
    }
    }
}

/* QSGD decompress, qsgd_compressor.py:66-71:
 *   out = (float(l) * (2*signs - 1)) * norm / s */
GQ_EXPORT void gq_oracle_qsgd_decompress(const float *norm, const uint8_t *signs, const int32_t *levels, int64_t Mb,
                                         int d, int n_bit, float *out) {
    gq_clean_vector_state();
    const float s = (float)(1 << n_bit);
    for (int64_t b = 0; b < Mb; ++b) {
        for (int j = 0; j < d; ++j) {
            int64_t i = b * (int64_t)d + j;
            float sv = (float)levels[i] * (2.0f * (float)signs[i] - 1.0f);
            float t = sv * norm[b];
            out[i] = t / s;
        }
    }
}

/*
 * ProbabilisticVectorCompressor encode -- compressors/probabilistic_vector_compressor.py:42-63.
 * PINNED by tests/golden/pvq_*.npz and residual_*.npz, which the reference itself produced with ONE
 * operation defined by the generator: the class calls torch.argmin on a bool tensor (:58), which no torch
 * with bool tensors implements; make_golden.py defines it as (index of the first True) - 1, i.e. the line's
 * `+ 1` lands on the first index whose cumulative probability reaches the draw -- the inverse-CDF sample
 * the code is written for (K-1 when no entry is True: all-zero subvector, 0/0 probabilities).  Every other
 * line ran unedited, and their arithmetic is what this function restates:
 *   p = c_dagger . v            :47  torch.mm == the ascending fmaf chain (as in hsq_encode)
 *   l1 = sum_k |p_k|            :48  torch.norm(p, 1, dim=1) over the strided view: sequential f32 adds, k ascending
 *   prob_k = |p_k| / l1         :49  IEEE f32 division
 *   cum_k = f32( sum_{i<=k} f64(prob_i) )   :57  torch.cumsum on the CPU accumulates in DOUBLE and rounds every
 *                                    output to f32 (measured: 100 % bit-equal on 2.8e8 entries; a f32 running
 *                                    sum matches only 25 % of them at K = 256)
 *   code = first k with cum_k >= r - 1e-5f  :52-58
 *   u = sign(p_code) * l1       :60-61
 * `cum_out` (nullable): the K cumulative sums of the first `cum_rows` subvectors, for the sub-expression test.
 */
GQ_EXPORT void gq_oracle_pvq_encode_ex(const float *grad, const float *cdag, int64_t M, int d, int K, const float *r,
                                       int32_t *codes, float *u, float *l1_out, float *p_out, float *cum_out,
                                       int64_t cum_rows) {
    gq_clean_vector_state();
#pragma omp parallel
    {
        gq_clean_vector_state();   /* per thread: see the comment at its definition */
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float *v = grad + m * (int64_t)d;
        float l1 = 0.0f;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc = fmaf(cdag[(int64_t)k * d + j], v[j], acc);
            l1 = l1 + fabsf(acc);
            if (p_out && m < cum_rows) p_out[m * (int64_t)K + k] = acc;
        }
        if (l1_out) l1_out[m] = l1;
        const float thr = r[m] - 1e-5f;
        double cum = 0.0;
        float sel = 0.0f;
        int code = K - 1, found = 0;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc = fmaf(cdag[(int64_t)k * d + j], v[j], acc);
            const float prob = fabsf(acc) / l1;
            cum = cum + (double)prob;
            const float cf = (float)cum;
            if (cum_out && m < cum_rows) cum_out[m * (int64_t)K + k] = cf;
            int hit = !found && (cf >= thr);
            if (hit || (!found && k == K - 1)) {
                code = k;
                sel = acc;
            }
            found = found || hit;
        }
        codes[m] = code;
        u[m] = (sel > 0.0f ? 1.0f : (sel < 0.0f ? -1.0f : 0.0f)) * l1;
    }
    }
}

GQ_EXPORT void gq_oracle_pvq_encode(const float *grad, const float *cdag, int64_t M, int d, int K, const float *r,
                                    int32_t *codes, float *u) {
    gq_clean_vector_state();
    gq_oracle_pvq_encode_ex(grad, cdag, M, d, K, r, codes, u, 0, 0, 0, 0);
}
