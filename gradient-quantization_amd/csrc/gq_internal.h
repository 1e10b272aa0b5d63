// Per-variant launchers behind the exported multi-tensor entry points (gq_api.hip dispatches on the descriptor).
// C names with hidden visibility: none of these is part of the ABI.
#pragma once
#include "gq_common.hpp"

// d = 8 / 12 / 16 / 24 / 32, K <= 256 (a multiple of 4), byte codes: f16 prefilter + exact rescoring + second pass (hsq_encode_pf.hip)
GQ_INTERNAL int gqi_hsq_encode_batched_pf(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                          const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire, float *u_flat,
                                          uint32_t *seg_minmax, float *workspace, int profile_slot, void *stream);
// d in {8, 16, 32}, K = 512 ... 65536 in pages of 256, int32 codes
GQ_INTERNAL int gqi_hsq_encode_batched_paged(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                             const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire,
                                             float *u_flat, uint32_t *seg_minmax, float *workspace, void *stream);
// any d <= 104 and K: exact f32 MFMA scoring (hsq_encode.hip)
GQ_INTERNAL int gqi_hsq_batched_any_supported(int d, int K);
GQ_INTERNAL int gqi_hsq_encode_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                           const float *codebook, int d, int K, int code_bytes, int ef, float ef_scale,
                                           uint8_t *wire, float *u_flat, uint32_t *seg_minmax, void *stream);

// level quantisers (hsq_batched.hip)
GQ_INTERNAL int gqi_hsq_levels_batched_d16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                           const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                           uint64_t seed, const float *r_flat, const float *ef_codebook, int K, int packed6,
                                           uint8_t *wire, const int64_t *dense_table, int ndense, void *stream);
GQ_INTERNAL int gqi_hsq_levels_batched_ef_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                            const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                            uint64_t seed, const float *r_flat, const float *codebook, int d, int K, uint8_t *wire, const int64_t *dense_table, int ndense,
                                            void *stream);
// 16-bit levels, K <= 256, d = 8 / 16 / 32: levels + error = v - decode(wire) in one launch (hsq_levels_ef_tile_kernel)
GQ_INTERNAL int gqi_hsq_levels_batched_ef16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                            const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                            uint64_t seed, const float *r_flat, const float *codebook, int d, int K, uint8_t *wire,
                                            const int64_t *dense_table, int ndense, void *stream);
// levels (+ residual) + decode of the finished payload + the step's tail in one launch (K <= 256, d = 8 / 16 / 32, byte or 16-bit levels)
GQ_INTERNAL int gqi_hsq_levels_decode_batched(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                              const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                              uint64_t seed, const float *r_flat, const float *codebook, int d, int K, int level_bytes,
                                              uint8_t *wire, const int64_t *dense_table, int ndense, int write_error, float *out,
                                              int plain, const gq::FusedTail *ft, void *stream);
GQ_INTERNAL int gqi_hsq_levels_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                           const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                                           uint64_t seed, const float *r_flat, int level_bytes, uint8_t *wire, const int64_t *dense_table, int ndense, void *stream);
GQ_INTERNAL int gqi_hsq_error_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                          const uint8_t *wire, const float *codebook, int d, int K, int code_bytes,
                                          int level_bytes, int n_bit, void *stream);

// decode + mean (hsq_batched.hip)
GQ_INTERNAL int gqi_hsq_decode_sum_batched_d16(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                               const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                               const float *codebook, int K, int n_bit, int packed6, float *out, int plain,
                                               const gq::StepTail *tail_or_null, int *tail_taken, void *stream);
GQ_INTERNAL int gqi_hsq_decode_sum_batched_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                             const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                             const float *codebook, int d, int K, int level_bytes, int n_bit, float *out, int plain,
                                             const gq::StepTail *tail_or_null, int *tail_taken, void *stream);
GQ_INTERNAL int gqi_hsq_decode_sum_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                               const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                               const float *codebook, int d, int K, int code_bytes, int level_bytes,
                                               int n_bit, float *out, int plain, void *stream);

// QSGD on the packed wire (qsgd_batched.hip, qsgd_wide.hip)
GQ_INTERNAL int gqi_qsgd_compress_batched(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                                          int n_bit, int random_mode, uint64_t seed, int ef, float ef_scale, uint8_t *wire, const int64_t *dense_table, int ndense,
                                          int bucket_hint, void *stream);
GQ_INTERNAL int gqi_qsgd_decode_sum_batched(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                                            int n_bit, int bits, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                            float *out, int plain, const gq::StepTail *tail_or_null, int *tail_taken, int bucket_hint,
                                            void *stream);
GQ_INTERNAL int gqi_qsgd_wide_compress(const int64_t *seg_table, const int32_t *chunk_seg, int nseg, int64_t nchunks,
                                       int n_bit, int random_mode, uint64_t seed, int ef, float ef_scale,
                                       uint32_t *norm_bits, uint8_t *wire, const int64_t *dense_table, int ndense, void *stream);
GQ_INTERNAL int gqi_qsgd_wide_decode_sum(const int64_t *seg_table, const int32_t *chunk_seg, int nseg, int64_t nchunks,
                                         int n_bit, int bits, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                         float *out, int plain, const gq::StepTail *tail_or_null, int *tail_taken, void *stream);
