// The multi-tensor entry points of include/gq_hsq.h: ONE encode / levels / decode per codec, configured by a
// descriptor, dispatching to the per-variant launchers of gq_internal.h.  What used to be a per-thread "next call"
// flag (plain decode, caller-supplied draws, armed profile slot) is an argument or a field here.
#include <math.h>

#include "gq_internal.h"

namespace gq {

static int check_batch(const gq_hsq_batch *b, const char *what) {
    if (!b) return fail(GQ_ERR_INVALID_ARG, "%s: null descriptor", what);
    if (b->struct_bytes != sizeof(gq_hsq_batch))
        return fail(GQ_ERR_INVALID_ARG, "%s: gq_hsq_batch.struct_bytes is %u, this library's layout has %zu", what,
                    b->struct_bytes, sizeof(gq_hsq_batch));
    if (b->nseg < 1 || b->ntiles < 1 || b->ntiles * 64 > 0x7FFFFFFFLL || b->d < 1 || b->K < 1 || b->K > 65536)
        return fail(GQ_ERR_INVALID_ARG, "%s: bad sizes nseg=%d ntiles=%lld d=%d K=%d", what, b->nseg, (long long)b->ntiles,
                    b->d, b->K);
    if (b->code_bytes != 1 && b->code_bytes != 4) return fail(GQ_ERR_INVALID_ARG, "%s: code_bytes must be 1 or 4", what);
    if (b->code_bytes == 1 && b->K > 256) return fail(GQ_ERR_INVALID_ARG, "%s: uint8 codes need K <= 256", what);
    if (b->level_bytes != 0 && b->level_bytes != 1 && b->level_bytes != 2 && b->level_bytes != 4 &&
        b->level_bytes != GQ_LEVELS_PACKED6)
        return fail(GQ_ERR_INVALID_ARG, "%s: level_bytes must be 0 (f32 projections), 1, 2, 4 or GQ_LEVELS_PACKED6", what);
    if (b->level_bytes == GQ_LEVELS_PACKED6 && !(b->d == 16 && b->K <= 256 && b->code_bytes == 1 && b->n_bit <= 6))
        return fail(GQ_ERR_UNSUPPORTED, "%s: the multi-tensor kernels take GQ_LEVELS_PACKED6 for d = 16, K <= 256, byte codes, n_bit <= 6", what);
    if (!b->seg_table || !b->tile_seg || !b->codebook) return fail(GQ_ERR_INVALID_ARG, "%s: null pointer in the descriptor", what);
    if (b->profile_slot >= GQ_PROFILE_SLOTS) return fail(GQ_ERR_INVALID_ARG, "%s: profile_slot %d", what, b->profile_slot);
    if (b->ndense < 0 || (b->ndense > 0 && !b->dense_table)) return fail(GQ_ERR_INVALID_ARG, "%s: ndense = %d without a dense_table", what, b->ndense);
    return GQ_OK;
}

static bool pf_dim(int d) { return d == 8 || d == 16 || d == 32; }      // shapes of the specialised level / decode kernels and of the paged encode
static bool pf_encode_dim(int d) { return pf_dim(d) || d == 12 || d == 24; }   // ... of the prefilter encode (K <= 256, a multiple of 4) (12 / 24: the reference's repaired dimensions, padded to 16 / 32)

// Which encode serves the descriptor (see gq_hsq_batched_path in the header); 0 + an error text if none.
static int batch_path(const gq_hsq_batch *b) {
    if (pf_encode_dim(b->d) && b->K >= 4 && b->K <= 256 && (b->K & 3) == 0 && b->code_bytes == 1) {
        return GQ_BATCH_PREFILTER;   // (any number of tensors: beyond 384 the segment records are read from global memory)
    } else if (pf_dim(b->d) && b->K > 256 && (b->K & 255) == 0 && b->code_bytes == 4 && b->nseg <= 384) {
        return GQ_BATCH_PAGED;
    }
    if (gqi_hsq_batched_any_supported(b->d, b->K)) return GQ_BATCH_EXACT;
    fail(GQ_ERR_UNSUPPORTED, "gq_hsq_batched_path: no multi-tensor kernel for d=%d K=%d code_bytes=%d nseg=%d", b->d, b->K,
         b->code_bytes, b->nseg);
    return 0;
}

// byte codes AND byte levels on a prefilter shape: the specialised level / decode kernels (fused error feedback)
static bool byte_wire(const gq_hsq_batch *b) {
    if (b->level_bytes == GQ_LEVELS_PACKED6) return b->d == 16 && b->K <= 256 && b->code_bytes == 1 && b->n_bit >= 1 && b->n_bit <= 6;
    return pf_dim(b->d) && b->K <= 256 && b->code_bytes == 1 && b->level_bytes == 1 && b->n_bit >= 1 && b->n_bit <= 8;
}

}  // namespace gq

GQ_API int gq_hsq_batched_path(const gq_hsq_batch *b) {
    if (gq::check_batch(b, "gq_hsq_batched_path") != GQ_OK) return 0;
    return gq::batch_path(b);
}

GQ_API int gq_hsq_encode_batched(const gq_hsq_batch *b, uint8_t *wire, float ef_scale, void *stream) {
    const int rc = gq::check_batch(b, "gq_hsq_encode_batched");
    if (rc != GQ_OK) return rc;
    if (!wire || !b->u_flat || !b->seg_minmax) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched: null pointer");
    const int ef = isnan(ef_scale) ? 0 : 1;
    const float scale = ef ? ef_scale : 0.0f;
    switch (gq::batch_path(b)) {
        case GQ_BATCH_PREFILTER:
            if (!b->workspace) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched: the prefilter path needs a workspace");
            return gqi_hsq_encode_batched_pf(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->codebook, b->d, b->K, ef, scale, wire,
                                             b->u_flat, b->seg_minmax, b->workspace, b->profile_slot, stream);
        case GQ_BATCH_PAGED:
            if (!b->workspace) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_encode_batched: the paged prefilter path needs a workspace");
            return gqi_hsq_encode_batched_paged(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->codebook, b->d, b->K, ef,
                                                scale, wire, b->u_flat, b->seg_minmax, b->workspace, stream);
        case GQ_BATCH_EXACT:
            return gqi_hsq_encode_batched_any(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->codebook, b->d, b->K,
                                              b->code_bytes, ef, scale, wire, b->u_flat, b->seg_minmax, stream);
        default:
            return GQ_ERR_UNSUPPORTED;   // text set by batch_path
    }
}

GQ_API int gq_hsq_levels_batched(const gq_hsq_batch *b, uint8_t *wire, int random_mode, uint64_t seed, const float *r_flat,
                                 int write_error, void *stream) {
    int rc = gq::check_batch(b, "gq_hsq_levels_batched");
    if (rc != GQ_OK) return rc;
    if (!wire || !b->u_flat || !b->seg_minmax) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: null pointer");
    if (random_mode != GQ_RANDOM_OFF && random_mode != GQ_RANDOM_GIVEN && random_mode != GQ_RANDOM_DEVICE &&
        random_mode != GQ_RANDOM_DEVICE_KEYED && random_mode != GQ_RANDOM_DEVICE_COUNTER)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: random_mode %d", random_mode);
    if (random_mode == GQ_RANDOM_DEVICE_COUNTER && (seed == 0 || (seed & 7) != 0))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_batched: GQ_RANDOM_DEVICE_COUNTER takes the address of two device words as `seed`");
    if (random_mode != GQ_RANDOM_GIVEN) r_flat = nullptr;
    if (gq::byte_wire(b)) {
        if (write_error && b->d != 16)
            return gqi_hsq_levels_batched_ef_d(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->u_flat, b->seg_minmax, b->n_bit,
                                               random_mode, seed, r_flat, b->codebook, b->d, b->K, wire, b->dense_table, b->ndense, stream);
        return gqi_hsq_levels_batched_d16(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->u_flat, b->seg_minmax, b->n_bit,
                                          random_mode, seed, r_flat, (write_error && b->d == 16) ? b->codebook : nullptr, b->K,
                                          b->level_bytes == GQ_LEVELS_PACKED6, wire, b->dense_table, b->ndense, stream);
    }
    if (b->level_bytes == GQ_LEVELS_PACKED6)
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_hsq_levels_batched: GQ_LEVELS_PACKED6 needs d = 16, K <= 256, n_bit <= 6");
    // error feedback with 16-bit levels on a prefilter shape (main.py's own defaults with --ef): levels and residual in one pass
    if (write_error && gq::pf_dim(b->d) && b->K <= 256 && b->code_bytes == 1 && b->level_bytes == 2 && b->n_bit >= 1 && b->n_bit <= 15)
        return gqi_hsq_levels_batched_ef16(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->u_flat, b->seg_minmax, b->n_bit,
                                           random_mode, seed, r_flat, b->codebook, b->d, b->K, wire, b->dense_table, b->ndense, stream);
    rc = gqi_hsq_levels_batched_any(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->u_flat, b->seg_minmax, b->n_bit,
                                    random_mode, seed, r_flat, b->level_bytes, wire, b->dense_table, b->ndense, stream);
    if (rc != GQ_OK || !write_error) return rc;
    return gqi_hsq_error_batched_any(b->seg_table, b->tile_seg, b->nseg, b->ntiles, wire, b->codebook, b->d, b->K,
                                     b->code_bytes, b->level_bytes, b->n_bit, stream);
}

namespace gq {
static int check_tail(const gq_step_tail *t, const char *what) {
    if (t->struct_bytes != sizeof(gq_step_tail))
        return fail(GQ_ERR_INVALID_ARG, "%s: gq_step_tail.struct_bytes is %u, this library's layout has %zu", what, t->struct_bytes,
                    sizeof(gq_step_tail));
    if (t->rows_R < 1 || t->n < 0 || (t->n > 0 && (!t->rows || !t->out)) || (t->row_stride_bytes & 3) != 0 ||
        (t->rng_state && (t->rng_pairs < 1 || t->rng_pairs > 256)) || t->reset_words < 0 ||
        (t->reset_words > 0 && (!t->reset_dst || !t->reset_src)))
        return fail(GQ_ERR_INVALID_ARG, "%s: bad gq_step_tail", what);
    return GQ_OK;
}
}  // namespace gq

namespace gq {
static StepTail tail_of(const gq_step_tail *t) {
    StepTail tail = {};
    tail.rows = static_cast<const uint8_t *>(t->rows);
    tail.row_stride_bytes = t->row_stride_bytes;
    tail.n = t->n;
    tail.out = t->out;
    tail.rng_state = t->rng_state;
    tail.reset_dst = t->reset_dst;
    tail.reset_src = t->reset_src;
    tail.R = t->rows_R;
    tail.rng_pairs = t->rng_state ? t->rng_pairs : 0;
    tail.reset_words = t->reset_words;
    return tail;
}
}  // namespace gq

GQ_API int gq_hsq_levels_decode_batched(const gq_hsq_batch *b, uint8_t *wire, int random_mode, uint64_t seed, const float *r_flat,
                                        int write_error, float *out, int plain, const gq_step_tail *t, void *stream) {
    int rc = gq::check_batch(b, "gq_hsq_levels_decode_batched");
    if (rc != GQ_OK) return rc;
    if (!wire || !out || !b->u_flat || !b->seg_minmax) return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: null pointer");
    if (t && (rc = gq::check_tail(t, "gq_hsq_levels_decode_batched")) != GQ_OK) return rc;
    if (t && (t->rows_R != 1 || ((t->rng_state || t->reset_words) && !t->ticket)))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: the tail of a one-payload step has rows_R = 1 and, with rng_state or reset words, a ticket word");
    if (t && t->n > 0 && (static_cast<const uint8_t *>(t->rows) < wire || (static_cast<const uint8_t *>(t->rows) - wire) % 4 != 0))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: tail.rows must point into `wire` (its dense region)");
    const bool served = gq::pf_dim(b->d) && b->K <= 256 && b->code_bytes == 1 && (b->level_bytes == 1 || b->level_bytes == 2) &&
                        b->n_bit >= 1 && b->n_bit <= (b->level_bytes == 1 ? 8 : 15) &&
                        (random_mode == GQ_RANDOM_OFF || random_mode == GQ_RANDOM_GIVEN || random_mode == GQ_RANDOM_DEVICE ||
                         random_mode == GQ_RANDOM_DEVICE_KEYED || random_mode == GQ_RANDOM_DEVICE_COUNTER) &&
                        !(b->level_bytes == 1 && ((int64_t)1 << b->n_bit) - (random_mode == GQ_RANDOM_OFF ? 1 : 0) > 255);
    if (!served) {   // the two launches: the same results
        rc = gq_hsq_levels_batched(b, wire, random_mode, seed, r_flat, write_error, stream);
        if (rc != GQ_OK) return rc;
        return gq_hsq_decode_sum_batched_tail(b, wire, 0, 1, out, plain, t, stream);
    }
    if (random_mode == GQ_RANDOM_DEVICE_COUNTER && (seed == 0 || (seed & 7) != 0))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_hsq_levels_decode_batched: GQ_RANDOM_DEVICE_COUNTER takes the address of two device words as `seed`");
    if (random_mode != GQ_RANDOM_GIVEN) r_flat = nullptr;
    gq::FusedTail ft = {};
    if (t) {
        ft.dense_mean = t->n > 0 ? t->out : nullptr;
        ft.dense_off = t->n > 0 ? static_cast<const uint8_t *>(t->rows) - wire : 0;
        ft.rng_state = t->rng_state;
        ft.rng_pairs = t->rng_state ? t->rng_pairs : 0;
        ft.reset_dst = t->reset_dst;
        ft.reset_src = t->reset_src;
        ft.reset_words = t->reset_words;
        ft.ticket = t->ticket;
    }
    return gqi_hsq_levels_decode_batched(b->seg_table, b->tile_seg, b->nseg, b->ntiles, b->u_flat, b->seg_minmax, b->n_bit, random_mode,
                                         seed, r_flat, b->codebook, b->d, b->K, b->level_bytes, wire, b->dense_table, b->ndense, write_error,
                                         out, plain, &ft, stream);
}

namespace gq {
// the decode-mean of `b`; *tail_taken = 1 when the launch also did the step's tail work
static int decode_sum_batched(const gq_hsq_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R, float *out,
                              int plain, const StepTail *tail, int *tail_taken, void *stream) {
    if (tail_taken) *tail_taken = 0;
    const int rc = check_batch(b, "gq_hsq_decode_sum_batched");
    if (rc != GQ_OK) return rc;
    if (!gathered || !out) return fail(GQ_ERR_INVALID_ARG, "gq_hsq_decode_sum_batched: null pointer");
    if (byte_wire(b)) {
        if (b->d == 16)
            return gqi_hsq_decode_sum_batched_d16(b->seg_table, b->tile_seg, b->nseg, b->ntiles, gathered, user_stride_bytes, R,
                                                  b->codebook, b->K, b->n_bit, b->level_bytes == GQ_LEVELS_PACKED6, out, plain, tail,
                                                  tail_taken, stream);
        return gqi_hsq_decode_sum_batched_d(b->seg_table, b->tile_seg, b->nseg, b->ntiles, gathered, user_stride_bytes, R,
                                            b->codebook, b->d, b->K, 1, b->n_bit, out, plain, tail, tail_taken, stream);
    }
    // 16-bit levels on a prefilter shape (main.py's own defaults: n_bit 8 with stochastic rounding reaches level 256)
    if (pf_dim(b->d) && b->K <= 256 && b->code_bytes == 1 && b->level_bytes == 2 && b->n_bit >= 1 && b->n_bit <= 15 && !(plain & 2))
        return gqi_hsq_decode_sum_batched_d(b->seg_table, b->tile_seg, b->nseg, b->ntiles, gathered, user_stride_bytes, R,
                                            b->codebook, b->d, b->K, 2, b->n_bit, out, plain, tail, tail_taken, stream);
    if (b->level_bytes == GQ_LEVELS_PACKED6)
        return fail(GQ_ERR_UNSUPPORTED, "gq_hsq_decode_sum_batched: GQ_LEVELS_PACKED6 needs d = 16, K <= 256, n_bit <= 6");
    return gqi_hsq_decode_sum_batched_any(b->seg_table, b->tile_seg, b->nseg, b->ntiles, gathered, user_stride_bytes, R,
                                          b->codebook, b->d, b->K, b->code_bytes, b->level_bytes, b->n_bit, out, plain, stream);
}
}  // namespace gq

GQ_API int gq_hsq_decode_sum_batched(const gq_hsq_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                     float *out, int plain, void *stream) {
    return gq::decode_sum_batched(b, gathered, user_stride_bytes, R, out, plain, nullptr, nullptr, stream);
}

GQ_API int gq_hsq_decode_sum_batched_tail(const gq_hsq_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                          float *out, int plain, const gq_step_tail *t, void *stream) {
    if (!t) return gq::decode_sum_batched(b, gathered, user_stride_bytes, R, out, plain, nullptr, nullptr, stream);
    const int trc = gq::check_tail(t, "gq_hsq_decode_sum_batched_tail");
    if (trc != GQ_OK) return trc;
    const gq::StepTail tail = gq::tail_of(t);
    int taken = 0;
    const int rc = gq::decode_sum_batched(b, gathered, user_stride_bytes, R, out, plain, &tail, &taken, stream);
    if (rc != GQ_OK || taken) return rc;
    // a decode path without the in-kernel tail (exact kernels, unaligned wires): the same work as a launch of its own
    return gq_mean_rows(t->rows, t->row_stride_bytes, t->rows_R, t->n, t->out, t->rng_state, t->rng_pairs, t->reset_dst, t->reset_src,
                        t->reset_words, stream);
}

// ---- QSGD ------------------------------------------------------------------------------------------------------
namespace gq {
static int check_qsgd(const gq_qsgd_batch *b, const char *what) {
    if (!b) return fail(GQ_ERR_INVALID_ARG, "%s: null descriptor", what);
    if (b->struct_bytes != sizeof(gq_qsgd_batch))
        return fail(GQ_ERR_INVALID_ARG, "%s: gq_qsgd_batch.struct_bytes is %u, this library's layout has %zu", what,
                    b->struct_bytes, sizeof(gq_qsgd_batch));
    if (b->nseg < 1 || b->nitems < 1 || b->n_bit < 1) return fail(GQ_ERR_INVALID_ARG, "%s: bad sizes", what);
    if (!b->seg_table || !b->item_seg) return fail(GQ_ERR_INVALID_ARG, "%s: null pointer in the descriptor", what);
    if (b->wide && !b->norm_bits) return fail(GQ_ERR_INVALID_ARG, "%s: wide buckets need norm_bits", what);
    if (b->ndense < 0 || (b->ndense > 0 && !b->dense_table)) return fail(GQ_ERR_INVALID_ARG, "%s: ndense = %d without a dense_table", what, b->ndense);
    return GQ_OK;
}
}  // namespace gq

GQ_API int gq_qsgd_compress_batched(const gq_qsgd_batch *b, uint8_t *wire, int random_mode, uint64_t seed, float ef_scale,
                                    void *stream) {
    const int rc = gq::check_qsgd(b, "gq_qsgd_compress_batched");
    if (rc != GQ_OK) return rc;
    if (!wire) return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress_batched: null pointer");
    if (b->bits != gq_qsgd_code_bits(b->n_bit, random_mode))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress_batched: bits = %d, but n_bit %d with random_mode %d packs to %d",
                        b->bits, b->n_bit, random_mode, gq_qsgd_code_bits(b->n_bit, random_mode));
    if (random_mode == GQ_RANDOM_DEVICE_COUNTER && (seed == 0 || (seed & 7) != 0))   // (the kernel dereferences it)
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_qsgd_compress_batched: GQ_RANDOM_DEVICE_COUNTER takes the address of two device words as `seed`");
    const int ef = isnan(ef_scale) ? 0 : 1;
    const float scale = ef ? ef_scale : 0.0f;
    if (b->wide)
        return gqi_qsgd_wide_compress(b->seg_table, b->item_seg, b->nseg, b->nitems, b->n_bit, random_mode, seed, ef, scale,
                                      b->norm_bits, wire, b->dense_table, b->ndense, stream);
    return gqi_qsgd_compress_batched(b->seg_table, b->item_seg, b->nseg, b->nitems, b->n_bit, random_mode, seed, ef, scale, wire,
                                     b->dense_table, b->ndense, b->bucket_hint, stream);
}

namespace gq {
static int qsgd_decode_sum_batched(const gq_qsgd_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R, float *out,
                                   int plain, const StepTail *tail, int *tail_taken, void *stream) {
    if (tail_taken) *tail_taken = 0;
    const int rc = check_qsgd(b, "gq_qsgd_decode_sum_batched");
    if (rc != GQ_OK) return rc;
    if (b->wide)
        return gqi_qsgd_wide_decode_sum(b->seg_table, b->item_seg, b->nseg, b->nitems, b->n_bit, b->bits, gathered,
                                        user_stride_bytes, R, out, plain, tail, tail_taken, stream);
    return gqi_qsgd_decode_sum_batched(b->seg_table, b->item_seg, b->nseg, b->nitems, b->n_bit, b->bits, gathered,
                                       user_stride_bytes, R, out, plain, tail, tail_taken, b->bucket_hint, stream);
}
}  // namespace gq

GQ_API int gq_qsgd_decode_sum_batched(const gq_qsgd_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                      float *out, int plain, void *stream) {
    return gq::qsgd_decode_sum_batched(b, gathered, user_stride_bytes, R, out, plain, nullptr, nullptr, stream);
}

GQ_API int gq_qsgd_decode_sum_batched_tail(const gq_qsgd_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                           float *out, int plain, const gq_step_tail *t, void *stream) {
    if (!t) return gq::qsgd_decode_sum_batched(b, gathered, user_stride_bytes, R, out, plain, nullptr, nullptr, stream);
    const int trc = gq::check_tail(t, "gq_qsgd_decode_sum_batched_tail");
    if (trc != GQ_OK) return trc;
    const gq::StepTail tail = gq::tail_of(t);
    int taken = 0;
    const int rc = gq::qsgd_decode_sum_batched(b, gathered, user_stride_bytes, R, out, plain, &tail, &taken, stream);
    if (rc != GQ_OK || taken) return rc;
    return gq_mean_rows(t->rows, t->row_stride_bytes, t->rows_R, t->n, t->out, t->rng_state, t->rng_pairs, t->reset_dst, t->reset_src,
                        t->reset_words, stream);
}
