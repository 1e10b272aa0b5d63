/*
 * CPU twins of the core C-ABI entry points ("CPU twins gq_cpu_* with identical
 * signatures -- the on-box oracle").
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: only tests/ may load this.  Each gq_cpu_X has the
 * parameter list of gq_X in include/gq_hsq.h (argument for argument; `stream` is ignored and every pointer is a
 * HOST pointer), returns the same GQ_OK / GQ_ERR_INVALID_ARG, and computes with the restated reference arithmetic
 * of gq_oracle.c (which cites the reference file:line of every step).  A test can therefore drive ONE piece of
 * ctypes code against libgq_hsq.so (device pointers) and against this library (host pointers) and compare the
 * outputs byte for byte (tests/test_cpu_twins.py).
 *
 * Workspace contract of the twins: the (min,max) of u at workspace[0], workspace[1] -- one pair where the GPU
 * library keeps one pair per workgroup; gq_cpu_hsq_workspace_bytes() is sized like the GPU's so that one
 * allocation serves both.  GQ_RANDOM_DEVICE (the counter-based generator inside the kernels) has no CPU twin:
 * INVALID_ARG.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GQ_EXPORT __attribute__((visibility("default")))
#define GQ_OK 0
#define GQ_ERR_INVALID_ARG (-1)
#define GQ_ERR_UNSUPPORTED (-2)
#define GQ_RANDOM_OFF 0
#define GQ_RANDOM_GIVEN 1
#define GQ_RANDOM_DEVICE 2

/* gq_oracle.c (same shared object) */
void gq_oracle_hsq_encode(const float *grad, const float *codebook, int64_t M, int d, int K, int32_t *codes, float *u);
void gq_oracle_minmax(const float *u, int64_t M, float *lb_ub);
void gq_oracle_scalar_levels(const float *u, int64_t M, int n_bit, int random, const float *r, float lb, float ub,
                             int32_t *levels);
void gq_oracle_scalar_decode(const int32_t *levels, int64_t M, int n_bit, float lb, float ub, float *norms);
void gq_oracle_hsq_decode(const int32_t *codes, const float *norms, const float *codebook, int64_t M, int d,
                          float *out);
void gq_oracle_mean_users(const float *decoded, int U, int64_t n, float *out);
void gq_oracle_qsgd_compress(const float *grad, int64_t Mb, int d, int n_bit, int random, const float *r, float *norm,
                             uint8_t *signs, int32_t *levels);
void gq_oracle_qsgd_decompress(const float *norm, const uint8_t *signs, const int32_t *levels, int64_t Mb, int d,
                               int n_bit, float *out);

/* gq_hsq_workspace_bytes: same figure as the GPU library's layout for the common case (4096 pairs + counters). */
GQ_EXPORT size_t gq_cpu_hsq_workspace_bytes(int64_t M) {
    (void)M;
    return (size_t)(2 * 4096 + 16) * sizeof(float);
}

static int store_codes(const int32_t *src, int64_t M, void *dst, int bytes) {
    if (bytes == 4) {
        memcpy(dst, src, (size_t)M * 4);
    } else if (bytes == 2) {
        for (int64_t i = 0; i < M; ++i) ((uint16_t *)dst)[i] = (uint16_t)src[i];
    } else {
        for (int64_t i = 0; i < M; ++i) ((uint8_t *)dst)[i] = (uint8_t)src[i];
    }
    return GQ_OK;
}

static int32_t *load_codes(const void *src, int64_t M, int bytes) {
    int32_t *dst = (int32_t *)malloc((size_t)(M > 0 ? M : 1) * 4);
    if (!dst) return NULL;
    for (int64_t i = 0; i < M; ++i)
        dst[i] = bytes == 4 ? ((const int32_t *)src)[i]
                            : (bytes == 2 ? (int32_t)((const uint16_t *)src)[i] : (int32_t)((const uint8_t *)src)[i]);
    return dst;
}

/* gq_hsq_encode (include/gq_hsq.h): nearest_neighbor_compressor.py:65-73 */
GQ_EXPORT int gq_cpu_hsq_encode(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes,
                                int code_bytes, float *u, float *workspace, void *stream) {
    (void)stream;
    if (M < 1 || d < 1 || K < 1 || !grad || !codebook || !codes || !u || !workspace) return GQ_ERR_INVALID_ARG;
    if (code_bytes != 1 && code_bytes != 4) return GQ_ERR_INVALID_ARG;
    if (code_bytes == 1 && K > 256) return GQ_ERR_INVALID_ARG;
    int32_t *c = (int32_t *)malloc((size_t)M * 4);
    if (!c) return GQ_ERR_UNSUPPORTED;
    gq_oracle_hsq_encode(grad, codebook, M, d, K, c, u);
    store_codes(c, M, codes, code_bytes);
    free(c);
    gq_oracle_minmax(u, M, workspace);
    return GQ_OK;
}

/* gq_minmax_partials: probabilistic_scalar_compressor.py:13-14 */
GQ_EXPORT int gq_cpu_minmax_partials(const float *v, int64_t n, float *workspace, void *stream) {
    (void)stream;
    if (n < 1 || !v || !workspace) return GQ_ERR_INVALID_ARG;
    gq_oracle_minmax(v, n, workspace);
    return GQ_OK;
}

/* GQ_LEVELS_PACKED6 (include/gq_hsq.h): four 6-bit levels per three bytes; a NaN quotient's INT_MIN travels as 0 like in the byte form */
#define GQ_LEVELS_PACKED6 (-6)
static void store_packed6(const int32_t *l, int64_t M, uint8_t *dst) {
    for (int64_t g = 0; 4 * g < M; ++g) {
        uint32_t w = 0;
        for (int k = 0; k < 4 && 4 * g + k < M; ++k) {
            const int32_t v = l[4 * g + k];
            w |= ((uint32_t)(v < 0 ? 0 : v) & 63u) << (6 * k);
        }
        dst[3 * g] = (uint8_t)w;
        dst[3 * g + 1] = (uint8_t)(w >> 8);
        dst[3 * g + 2] = (uint8_t)(w >> 16);
    }
}
static int32_t *load_packed6(const uint8_t *src, int64_t M) {
    int32_t *l = (int32_t *)malloc((size_t)M * 4);
    if (!l) return 0;
    for (int64_t m = 0; m < M; ++m) {
        const uint8_t *p = src + 3 * (m >> 2);
        const uint32_t w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
        l[m] = (int32_t)((w >> (6 * (m & 3))) & 63u);
    }
    return l;
}

/* gq_hsq_levels: probabilistic_scalar_compressor.py:12-27 */
GQ_EXPORT int gq_cpu_hsq_levels(const float *u, int64_t M, int n_bit, int random_mode, const float *r, uint64_t seed,
                                const float *workspace, float *lb_ub, void *levels, int level_bytes, void *stream) {
    (void)stream;
    (void)seed;
    (void)workspace;   /* the oracle recomputes min / max from u; the pair at the head of the workspace is the same */
    if (M < 1 || !u || !lb_ub || !levels || n_bit < 1 || n_bit > 30) return GQ_ERR_INVALID_ARG;
    if (level_bytes != 1 && level_bytes != 2 && level_bytes != 4 && level_bytes != GQ_LEVELS_PACKED6) return GQ_ERR_INVALID_ARG;
    if (level_bytes == GQ_LEVELS_PACKED6 && (1 << n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0) > 63) return GQ_ERR_INVALID_ARG;
    if (random_mode != GQ_RANDOM_OFF && random_mode != GQ_RANDOM_GIVEN) return GQ_ERR_INVALID_ARG;
    if (random_mode == GQ_RANDOM_GIVEN && !r) return GQ_ERR_INVALID_ARG;
    int32_t *l = (int32_t *)malloc((size_t)M * 4);
    if (!l) return GQ_ERR_UNSUPPORTED;
    gq_oracle_minmax(u, M, lb_ub);
    gq_oracle_scalar_levels(u, M, n_bit, random_mode == GQ_RANDOM_GIVEN, r, lb_ub[0], lb_ub[1], l);
    if (level_bytes == GQ_LEVELS_PACKED6)
        store_packed6(l, M, (uint8_t *)levels);
    else
        store_codes(l, M, levels, level_bytes);
    free(l);
    return GQ_OK;
}

/* gq_hsq_decode_sum: probabilistic_scalar_compressor.py:29-33, nearest_neighbor_compressor.py:80-90,
 * ps_quantizer.py:48 */
GQ_EXPORT int gq_cpu_hsq_decode_sum(const void *codes, int code_bytes, const void *levels, int level_bytes,
                                    const float *lb_ub, const float *codebook, int R, int64_t M, int d, int K,
                                    int n_bit, float *out, void *stream) {
    (void)stream;
    (void)K;
    if (R < 1 || M < 1 || d < 1 || !codes || !levels || !codebook || !out) return GQ_ERR_INVALID_ARG;
    if (code_bytes != 1 && code_bytes != 4) return GQ_ERR_INVALID_ARG;
    if (level_bytes != 0 && level_bytes != 1 && level_bytes != 2 && level_bytes != 4 && level_bytes != GQ_LEVELS_PACKED6)
        return GQ_ERR_INVALID_ARG;
    if (level_bytes != 0 && !lb_ub) return GQ_ERR_INVALID_ARG;
    const int64_t n = M * (int64_t)d;
    float *dec = (float *)malloc((size_t)R * (size_t)n * sizeof(float));
    float *norms = (float *)malloc((size_t)M * sizeof(float));
    if (!dec || !norms) {
        free(dec);
        free(norms);
        return GQ_ERR_UNSUPPORTED;
    }
    for (int r = 0; r < R; ++r) {
        int32_t *c = load_codes((const char *)codes + (size_t)r * (size_t)M * (size_t)code_bytes, M, code_bytes);
        if (level_bytes == 0) {
            memcpy(norms, (const float *)levels + (size_t)r * (size_t)M, (size_t)M * sizeof(float));
        } else {
            int32_t *l = level_bytes == GQ_LEVELS_PACKED6
                             ? load_packed6((const uint8_t *)levels + (size_t)r * (size_t)(3 * ((M + 3) / 4)), M)
                             : load_codes((const char *)levels + (size_t)r * (size_t)M * (size_t)level_bytes, M, level_bytes);
            gq_oracle_scalar_decode(l, M, n_bit, lb_ub[2 * r], lb_ub[2 * r + 1], norms);
            free(l);
        }
        gq_oracle_hsq_decode(c, norms, codebook, M, d, dec + (size_t)r * (size_t)n);
        free(c);
    }
    if (R == 1)
        memcpy(out, dec, (size_t)n * sizeof(float));   /* the plain decompress: a -0 stays -0 */
    else
        gq_oracle_mean_users(dec, R, n, out);
    free(dec);
    free(norms);
    return GQ_OK;
}

/* gq_hsq_levels_decode: the two calls above, one after the other (R = 1, d = 16, byte codes) */
GQ_EXPORT int gq_cpu_hsq_levels_decode(const float *u, int64_t M, int n_bit, int random_mode, const float *r, uint64_t seed,
                                       const float *minmax_partials, float *lb_ub, void *levels, int level_bytes, const void *codes,
                                       const float *codebook, int K, float *out, void *stream) {
    if (level_bytes != 1 && level_bytes != GQ_LEVELS_PACKED6) return GQ_ERR_UNSUPPORTED;
    int rc = gq_cpu_hsq_levels(u, M, n_bit, random_mode, r, seed, minmax_partials, lb_ub, levels, level_bytes, stream);
    if (rc != GQ_OK) return rc;
    return gq_cpu_hsq_decode_sum(codes, 1, levels, level_bytes, lb_ub, codebook, 1, M, 16, K, n_bit, out, stream);
}

/* gq_qsgd_compress: qsgd_compressor.py:47-64 */
GQ_EXPORT int gq_cpu_qsgd_compress(const float *grad, int64_t Mb, int d, int n_bit, int random_mode, const float *r,
                                   uint64_t seed, float *norm, uint8_t *signs, void *levels, int level_bytes,
                                   void *stream) {
    (void)stream;
    (void)seed;
    if (Mb < 1 || d < 1 || !grad || !norm || !signs || !levels || n_bit < 1 || n_bit > 30) return GQ_ERR_INVALID_ARG;
    if (level_bytes != 1 && level_bytes != 4) return GQ_ERR_INVALID_ARG;
    if (random_mode != GQ_RANDOM_OFF && random_mode != GQ_RANDOM_GIVEN) return GQ_ERR_INVALID_ARG;
    if (random_mode == GQ_RANDOM_GIVEN && !r) return GQ_ERR_INVALID_ARG;
    const int64_t n = Mb * (int64_t)d;
    int32_t *l = (int32_t *)malloc((size_t)n * 4);
    if (!l) return GQ_ERR_UNSUPPORTED;
    gq_oracle_qsgd_compress(grad, Mb, d, n_bit, random_mode == GQ_RANDOM_GIVEN, r, norm, signs, l);
    if (level_bytes == 4) {
        memcpy(levels, l, (size_t)n * 4);
    } else {
        /* a zero bucket's INT_MIN is stored as 0 in the byte form (include/gq_hsq.h) */
        for (int64_t i = 0; i < n; ++i) ((uint8_t *)levels)[i] = l[i] == INT_MIN ? 0 : (uint8_t)l[i];
    }
    free(l);
    return GQ_OK;
}

/* gq_qsgd_decode_sum: qsgd_compressor.py:66-71 and ps_quantizer.py:48 */
GQ_EXPORT int gq_cpu_qsgd_decode_sum(const float *norm, const uint8_t *signs, const void *levels, int level_bytes,
                                     int R, int64_t Mb, int d, int n_bit, float *out, void *stream) {
    (void)stream;
    if (R < 1 || Mb < 1 || d < 1 || !norm || !signs || !levels || !out) return GQ_ERR_INVALID_ARG;
    if (level_bytes != 1 && level_bytes != 4) return GQ_ERR_INVALID_ARG;
    const int64_t n = Mb * (int64_t)d;
    float *dec = (float *)malloc((size_t)R * (size_t)n * sizeof(float));
    if (!dec) return GQ_ERR_UNSUPPORTED;
    for (int r = 0; r < R; ++r) {
        int32_t *l = load_codes((const char *)levels + (size_t)r * (size_t)n * (size_t)level_bytes, n, level_bytes);
        gq_oracle_qsgd_decompress(norm + (size_t)r * (size_t)Mb, signs + (size_t)r * (size_t)n, l, Mb, d, n_bit,
                                  dec + (size_t)r * (size_t)n);
        free(l);
    }
    if (R == 1)
        memcpy(out, dec, (size_t)n * sizeof(float));
    else
        gq_oracle_mean_users(dec, R, n, out);
    free(dec);
    return GQ_OK;
}

/* gq_axpy_inplace, gq_sub: ps_quantizer.py:35, :39 */
GQ_EXPORT int gq_cpu_axpy_inplace(float *grad, const float *err, float scale, int64_t n, void *stream) {
    (void)stream;
    if (n < 0 || !grad || !err) return GQ_ERR_INVALID_ARG;
    for (int64_t i = 0; i < n; ++i) {
        const float t = scale * err[i];   /* the product is rounded, then the add (no fma: -ffp-contract=off) */
        grad[i] = grad[i] + t;
    }
    return GQ_OK;
}

GQ_EXPORT int gq_cpu_sub(const float *grad, const float *decoded, float *err, int64_t n, void *stream) {
    (void)stream;
    if (n < 0 || !grad || !decoded || !err) return GQ_ERR_INVALID_ARG;
    for (int64_t i = 0; i < n; ++i) err[i] = grad[i] - decoded[i];
    return GQ_OK;
}
