/*
 * gq_hsq.h -- C ABI of libgq_hsq.so, the MI355X (gfx950) implementation of the
 * gradient vector-quantisation hot path of xinyandai/gradient-quantization.
 *
 * The reference has no FFI: its boundary is the duck-typed Python protocol
 *     Compressor(size, shape, args).compress(vec) / .decompress(signature)
 *     Quantizer(Compressor, parameters, args).record(user, epoch) / .apply()
 * (compressors/nearest_neighbor_compressor.py:10,63,80; quantizers/ps_quantizer.py:7,27,46).
 * Each entry point below replaces the PyTorch op sequence inside one of those
 * methods; the citation on each says which.  The Python classes with the
 * reference's names live in gradient-quantization_amd/ and call these through
 * ctypes with raw device pointers (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name starts with `h_`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *     call only enqueues work on that stream: no allocation, no synchronisation,
 *     graph-capturable;
 *   - return value: GQ_OK (0) or a negative GQ_ERR_* code; gq_last_error() gives text;
 *   - nothing here falls back to the CPU: without a gfx950 device the calls fail.
 */
#ifndef GQ_HSQ_H
#define GQ_HSQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQ_OK 0
#define GQ_ERR_INVALID_ARG (-1)
#define GQ_ERR_UNSUPPORTED (-2)
#define GQ_ERR_HIP (-3)

/* Encode workspace (caller-allocated device memory, gq_hsq_workspace_bytes(M) bytes, 16-byte
 * aligned):
 *   [ (min,max) f32 pairs x GQ_MAX_PARTIALS | int32 x4: -, -, final flag, - | fix-up log int32[M] ]
 * The int32 x4 block (bytes [8*GQ_MAX_PARTIALS, +16)) must be ZERO before the first use; the
 * library keeps it consistent afterwards (no memset per call).  gq_hsq_encode leaves the
 * per-workgroup (min,max) of u in the pairs (unused slots hold (+inf,-inf); a workgroup that met a NaN
 * projection leaves (NaN, NaN)), final flag 0, and gq_hsq_levels folds them into (lb, ub) -- NaN if any
 * pair is NaN, as torch.min / torch.max propagate it (probabilistic_scalar_compressor.py:13-14).  (A caller
 * may instead put the final pair at slot 0 and raise the flag.)  The log marks the subvectors the prefilter
 * path recomputed exactly: entry i holds i (diagnostics; the caller fills it with -1 beforehand). */
#define GQ_MAX_PARTIALS 1024
#define GQ_FIXUP_PARTIALS 256 /* slots no grid writes: gq_hsq_encode fills them with (+inf,-inf) */
size_t gq_hsq_workspace_bytes(int64_t M);

/* random_mode of gq_hsq_levels / gq_qsgd_compress */
#define GQ_RANDOM_OFF 0    /* args.random == 0: deterministic truncation                      */
#define GQ_RANDOM_GIVEN 1  /* stochastic rounding against caller-supplied r[] (reference parity:
                              r = torch.rand(M) from the CPU generator, prob_scalar:23-25)       */
#define GQ_RANDOM_DEVICE 2 /* stochastic rounding against an on-device counter-based generator
                              seeded by `seed` (same distribution, not the same draws)          */

/* Library / device identification. */
int gq_abi_version(void);
const char *gq_last_error(void);
/*
 * Per-dispatch timing of the dominant kernel (bench.py's roofline object): gq_profile_arm(slot) makes the next
 * d16/K256 gq_hsq_encode issued by this thread attach a start / stop HIP event pair to its kernel dispatch
 * (hipExtLaunchKernelGGL), so that the pair measures the kernel alone -- an event bracket recorded around the call
 * also measures ~5-8 us of queue bubbles.  gq_profile_read(slot, &ms) waits for that dispatch and returns its
 * duration.  Nothing else changes; unarmed calls are not affected.
 */
#define GQ_PROFILE_SLOTS 64
int gq_profile_arm(int slot);
int gq_profile_read(int slot, float *kernel_ms);

/* Fills CU count and the gcnArchName (e.g. "gfx950:sramecc+:xnack-") of `device`. */
int gq_device_info(int device, int *cu_count, char *arch, size_t arch_len);

/*
 * HSQ encode -- replaces nearest_neighbor_compressor.py:65-73 (view(-1,d); mm;
 * abs; argmax; gather).  For every d-float subvector v of `grad` (M of them,
 * row-major, contiguous):
 *     p_k   = <codebook[k,:], v>   exactly as  acc=0; for j: acc=fmaf(c[j],v[j],acc)
 *     code  = first k maximising |p_k|
 *     u     = p_code              (signed)
 * Outputs: codes[M] (code_bytes = 1 -> uint8, 4 -> int32; 1 requires K <= 256),
 * u[M] f32, and per-workgroup (min,max) of u at the head of `workspace` (see above).
 * Requirements: 1 <= d <= 512, 1 <= K <= 65536, M < 2^31, grad 16-byte aligned when d == 16.
 */
int gq_hsq_encode(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes, int code_bytes,
                  float *u, float *workspace, void *stream);

/*
 * The whole compress of NearestNeighborCompressor (nearest_neighbor_compressor.py:63-78: the encode above, then
 * probabilistic_scalar_compressor.py:12-27 on u) in one call: gq_hsq_encode followed by gq_hsq_levels (declared
 * below) on the same stream, same outputs.
 */
int gq_hsq_compress(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes, int code_bytes,
                    float *u, float *workspace, int n_bit, int random_mode, const float *r, uint64_t seed,
                    float *lb_ub, void *levels, int level_bytes, void *stream);

/* Same, with the kernel chosen explicitly (diagnostics / cross-checks; results are
 * identical for every impl):  0 = auto, 1 = exact f32 MFMA, d16/K256, register-resident
 * codebook, 2 = exact f32 MFMA generic (any d, K), 3 = VALU fmaf chain with the codebook in
 * LDS, 4 = bf16x3 MFMA prefilter + exact f32 rescoring + exact fix-up for K = 256 and d in
 * {8, 16, 32} (the default for those shapes; bit-identical output), 5 = exact f32 MFMA with the codebook (chunked when it
 * does not fit) and the subvector tiles staged in LDS, any d <= 128 and any K (the default for
 * every other shape; 2 remains the fallback for d > 128). */
#define GQ_ENCODE_AUTO 0
#define GQ_ENCODE_MFMA_D16K256 1
#define GQ_ENCODE_MFMA_GENERIC 2
#define GQ_ENCODE_VALU 3
#define GQ_ENCODE_PREFILTER_D16K256 4
#define GQ_ENCODE_MFMA_LDS 5
int gq_hsq_encode_impl(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes,
                       int code_bytes, float *u, float *workspace, int impl, void *stream);

/*
 * Scalar level quantiser -- replaces probabilistic_scalar_compressor.py:12-27.
 *     lb = min(u), ub = max(u)    (finished here from the (min,max) pairs at the head of `workspace`)
 *     levels = 0                                   if lb - ub == 0
 *     x = |(u-lb)/(ub-lb)| * 2^n_bit ;  l = trunc(clamp(x, 0, 2^n_bit - 1))
 *     l += (x - l > r)                             if random_mode != GQ_RANDOM_OFF
 * Outputs lb_ub[2] (f32) and levels[M] (level_bytes 1/2/4 -> uint8/uint16/int32;
 * the value range is [0, 2^n_bit] with stochastic rounding, [0, 2^n_bit - 1] without).
 */
int gq_hsq_levels(const float *u, int64_t M, int n_bit, int random_mode, const float *r, uint64_t seed,
                  const float *workspace, float *lb_ub, void *levels, int level_bytes, void *stream);

/* Per-workgroup (min,max) of an arbitrary f32 vector into the head of a workspace
 * (gq_hsq_workspace_bytes(0) bytes suffice), for running gq_hsq_levels on a vector that did not
 * come out of gq_hsq_encode (ProbabilisticScalarCompressor used on its own; torch.min/torch.max
 * at prob_scalar:13-14). */
int gq_minmax_partials(const float *v, int64_t n, float *workspace, void *stream);

/*
 * Decode + aggregate -- replaces probabilistic_scalar_compressor.py:29-33 and
 * nearest_neighbor_compressor.py:80-90 for each of R payloads, then
 * ps_quantizer.py:48 (torch.stack(...).mean(0)) across them:
 *     n_r[m]   = float(levels_r[m]) * (ub_r - lb_r) / 2^n_bit + lb_r     (no fusion)
 *     out[m,:] = ( sum_{r=0..R-1, ascending} codebook[codes_r[m],:] * n_r[m] ) / R
 * codes: [R][M], levels: [R][M], lb_ub: [R][2].  level_bytes == 0 means `levels`
 * holds f32 norms [R][M] (the n_bit == 32 signature) and lb_ub / n_bit are ignored.
 * R == 1 is the plain decompress (a decoded -0 stays -0, as nearest_neighbor_compressor.py:85-90 returns it).  R > 1 is the
 * aggregate: torch's sum starts from +0, so the sum is (+0 + p_0 + ... + p_{R-1}) and an element whose payloads are all -0
 * comes out as +0.  The multi-tensor decodes (gq_*_decode_sum_batched*, gq_qsgd_wide_decode_sum) are always the aggregate,
 * also for R == 1 (ps_quantizer.py:48 takes the mean of one user's stack as well).
 */
int gq_hsq_decode_sum(const void *codes, int code_bytes, const void *levels, int level_bytes, const float *lb_ub,
                      const float *codebook, int R, int64_t M, int d, int K, int n_bit, float *out, void *stream);

/* Same with explicit per-payload strides (bytes): payload r starts at codes + r*code_stride_bytes,
 * levels + r*level_stride_bytes, lb_ub + r*lbub_stride_bytes.  This is what the all-gathered
 * wire buffer [R][codes | levels | lb,ub] is decoded from without a repack. */
int gq_hsq_decode_sum_strided(const void *codes, int code_bytes, int64_t code_stride_bytes, const void *levels,
                              int level_bytes, int64_t level_stride_bytes, const float *lb_ub,
                              int64_t lbub_stride_bytes, const float *codebook, int R, int64_t M, int d, int K,
                              int n_bit, float *out, void *stream);

/*
 * Multi-tensor (batched) forms for d = 16, K = 256, uint8 codes and levels: ONE launch serves every
 * tensor of a model that shares the codebook (the reference loops over parameters in Python,
 * ps_quantizer.py:33,47).  lb / ub stay per tensor.  Every tensor is padded to whole
 * 64-subvector tiles in a common index space of `ntiles` tiles:
 *   tile_seg   int32[ntiles]       tensor ("segment") of each tile, ascending
 *   seg_table  int64[nseg][8]      { grad pointer (16-byte aligned), M subvectors, first tile,
 *                                    byte offset of codes / of levels / of (lb,ub) inside ONE
 *                                    user's wire, float offset of the tensor in `out`, reserved }
 *   seg_minmax uint32[nseg][2]     order-mapped (min,max) of u; the caller resets it before each
 *                                  encode to { 0xFFFFFFFF, 0 }
 *   u_flat     float[ntiles*64]    projection spill, padded index space
 *   workspace  gq_hsq_workspace_bytes(ntiles*64) bytes (same contract as gq_hsq_encode)
 * gq_hsq_encode_batched writes codes into `wire`, u into u_flat and folds (min,max) into
 * seg_minmax; gq_hsq_levels_batched writes levels and (lb,ub) into `wire`;
 * gq_hsq_decode_sum_batched averages R users' wires (`gathered` + r*user_stride_bytes) into `out`.
 * Results are identical to the per-tensor entry points.
 *
 * Error-feedback forms (ps_quantizer.py:34-39 for all tensors at once): seg_table[seg][7] is the
 * tensor's error buffer (float*, 16-byte aligned; 0 = no feedback for this tensor).
 * gq_hsq_encode_batched_ef reads every tile as v = grad + ef_scale*error (product rounded, then the
 * add), writes v back over grad like the reference's in-place add_, and encodes v;
 * gq_hsq_levels_batched_ef additionally writes error = v - decoded (the decode of the wire it has just
 * completed) over the old error.  Same results as gq_axpy_inplace + encode + levels + decode + gq_sub.
 */
int gq_hsq_encode_batched(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                          const float *codebook, uint8_t *wire, float *u_flat, uint32_t *seg_minmax,
                          float *workspace, void *stream);
int gq_hsq_levels_batched(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                          const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                          uint64_t seed, uint8_t *wire, void *stream);
int gq_hsq_encode_batched_ef(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                             const float *codebook, float ef_scale, uint8_t *wire, float *u_flat,
                             uint32_t *seg_minmax, float *workspace, void *stream);
int gq_hsq_levels_batched_ef(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                             const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                             uint64_t seed, const float *codebook, uint8_t *wire, void *stream);
int gq_hsq_decode_sum_batched(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                              const uint8_t *gathered, int64_t user_stride_bytes, int R, const float *codebook,
                              int n_bit, float *out, void *stream);
/* The same for the other sub-dimensions that have a prefilter encode (K = 256, d = 8, 16 or 32; d = 16
 * forwards to the entry points above); the _ef forms as gq_hsq_encode_batched_ef / gq_hsq_levels_batched_ef. */
int gq_hsq_encode_batched_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                            const float *codebook, int d, uint8_t *wire, float *u_flat, uint32_t *seg_minmax,
                            float *workspace, void *stream);
int gq_hsq_encode_batched_d_ef(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                               const float *codebook, int d, float ef_scale, uint8_t *wire, float *u_flat,
                               uint32_t *seg_minmax, float *workspace, void *stream);
int gq_hsq_levels_batched_ef_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                               const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                               uint64_t seed, const float *codebook, int d, uint8_t *wire, void *stream);
int gq_hsq_decode_sum_batched_d(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                const uint8_t *gathered, int64_t user_stride_bytes, int R, const float *codebook,
                                int d, int n_bit, float *out, void *stream);

/* The same multi-tensor launches for ANY other shape the reference can be configured to
 * (nearest_neighbor_compressor.py:23-57: d from --c-dim and its repair loop, K = 2^k_bit or d, int32 codes when
 * k_bit > 8; probabilistic_scalar_compressor.py:20-25: n_bit = 8 with stochastic rounding reaches level 256):
 * code_bytes 1 | 4, level_bytes 1 | 2 | 4, sections of the wire 16-byte aligned.  Exact f32 scoring (the LDS
 * kernel of gq_hsq_encode, d <= 128).  gq_hsq_encode_batched_any: `ef` != 0 reads every tile as
 * grad + ef_scale*error (seg_table[seg][7]) and writes it back;  gq_hsq_error_batched_any then writes
 * error = grad - decode(wire) for the rows that have an error buffer (ps_quantizer.py:39).
 * gq_hsq_levels_batched_any is independent of (d, K) and also serves the prefilter encodes above. */
/* Multi-tensor prefilter encode for the larger codebooks of the prefilter dimensions (d = 8, 16 or 32; K = 512,
 * 768, ... : `--k-bit 9` and up, int32 codes): one launch per page of 256 codewords, a page's exact winner merged
 * into the (code, u) the earlier pages left in `wire` / `u_flat` (an earlier page keeps a tie: the first maximum);
 * same table, workspace and results as gq_hsq_encode_batched_any, 3-4x faster.  ef != 0: error feedback (in the
 * first page's launch).  At most 384 tensors per launch. */
int gq_hsq_encode_batched_paged(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                const float *codebook, int d, int K, int ef, float ef_scale, uint8_t *wire,
                                float *u_flat, uint32_t *seg_minmax, float *workspace, void *stream);
int gq_hsq_batched_any_supported(int d, int K);   /* 1 if gq_hsq_encode_batched_any serves (d, K) */
/* Stochastic rounding with the REFERENCE's draws in the multi-tensor level kernels
 * (probabilistic_scalar_compressor.py:23-26: r = torch.rand(M) per tensor, CPU generator): the caller draws them --
 * torch.rand is one sequential stream, so ONE torch.rand(sum of M) per record() equals the reference's per-tensor
 * calls in parameter order -- lays them out like u_flat (r_flat[tile * 64 + i] belongs to subvector i of tile
 * `tile`; padding slots are not read) and hands the device pointer over with gq_hsq_given_draws; the NEXT
 * gq_hsq_levels_batched* call on this thread with random_mode = GQ_RANDOM_GIVEN consumes it (compare is the
 * reference's strict `frac > r`). */
int gq_hsq_given_draws(const float *r_flat);

/*
 * The NEXT multi-tensor decode issued by this thread (gq_hsq_decode_sum_batched*, gq_qsgd_decode_sum_batched,
 * gq_qsgd_wide_decode_sum) is a plain decompress of its one payload instead of the parameter-server aggregate: a decoded
 * -0 stays -0.  This is the ring's hop and its final gradient (ring_quantizer.py:32,41-43,47: grad.add_(decompress(...)),
 * param.grad.data = the last decompress), where the aggregate's (+0 + sum) / R would turn those zeros positive.  Only the sign of
 * zeros differs between the two.
 */
int gq_decode_plain_next(void);
int gq_hsq_encode_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                              const float *codebook, int d, int K, int code_bytes, int ef, float ef_scale,
                              uint8_t *wire, float *u_flat, uint32_t *seg_minmax, void *stream);
int gq_hsq_levels_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                              const float *u_flat, const uint32_t *seg_minmax, int n_bit, int random_mode,
                              uint64_t seed, int level_bytes, uint8_t *wire, void *stream);
int gq_hsq_decode_sum_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                                  const uint8_t *gathered, int64_t user_stride_bytes, int R, const float *codebook,
                                  int d, int K, int code_bytes, int level_bytes, int n_bit, float *out, void *stream);
int gq_hsq_error_batched_any(const int64_t *seg_table, const int32_t *tile_seg, int nseg, int64_t ntiles,
                             const uint8_t *wire, const float *codebook, int d, int K, int code_bytes,
                             int level_bytes, int n_bit, void *stream);

/*
 * Error-feedback helpers fused around the codec (ps_quantizer.py:35,39):
 *     gq_axpy_inplace:   grad += scale * err
 *     gq_sub:            err   = grad - decoded
 */
int gq_axpy_inplace(float *grad, const float *err, float scale, int64_t n, void *stream);
int gq_sub(const float *grad, const float *decoded, float *err, int64_t n, void *stream);

/*
 * The aggregate of tensors that travel uncompressed (IdenticalCompressor, ps_quantizer.py:18,48): out[i] =
 * (+0 + rows_0[i] + ... + rows_{R-1}[i]) / R, rows ascending, a true division -- the arithmetic of torch.stack(...).mean(0)
 * on the CPU.  (torch's GPU mean multiplies by 1/R and sums in another order: last-bit differences for R = 3, 5, 6, 7.)
 * rows_r = (const float *)((const char *)rows + r * row_stride_bytes).
 */
int gq_mean_rows(const void *rows, int64_t row_stride_bytes, int R, int64_t n, float *out, void *stream);

/*
 * QSGD compress -- replaces qsgd_compressor.py:47-64.  `grad` is Mb buckets of d floats.
 *     norm[b]  = max_j |v_j| ;  x = |v/norm| * 2^n_bit ;  l = trunc(clamp(x, 0, 2^n_bit-1))
 *     l += (x - l > r)  if random_mode ;  signs = v > 0
 * Outputs norm[Mb] f32, signs[Mb*d] uint8 (0/1), levels[Mb*d] (level_bytes 1 or 4).
 * A zero bucket yields level INT_MIN (int32) / 0 (uint8) and decodes to 0, as in the reference.
 */
int gq_qsgd_compress(const float *grad, int64_t Mb, int d, int n_bit, int random_mode, const float *r, uint64_t seed,
                     float *norm, uint8_t *signs, void *levels, int level_bytes, void *stream);

/*
 * QSGD decode + aggregate -- replaces qsgd_compressor.py:66-71 for R payloads and
 * ps_quantizer.py:48:   out = ( sum_r (float(l_r) * (2*signs_r - 1)) * norm_r / 2^n_bit ) / R
 * norm: [R][Mb], signs: [R][Mb*d], levels: [R][Mb*d].
 */
int gq_qsgd_decode_sum(const float *norm, const uint8_t *signs, const void *levels, int level_bytes, int R, int64_t Mb,
                       int d, int n_bit, float *out, void *stream);

/*
 * ProbabilisticVectorCompressor encode -- the INTENDED semantics of
 * probabilistic_vector_compressor.py:42-63 (the reference's own code cannot run):
 *     p = c_dagger . v   (c_dagger = pinv(codewords^T), [K,d]) ;  l1 = sum_k |p_k|
 *     code = first k with cumsum_k(|p|/l1) >= r - 1e-5 ;  u = sign(p_code) * l1
 * r: one uniform draw per subvector (GQ_RANDOM_GIVEN: caller-supplied r[M]; GQ_RANDOM_DEVICE: in-kernel).
 * Outputs codes[M], u[M] and the (min,max) partials of u in `workspace` (gq_hsq_workspace_bytes(0)
 * bytes suffice) so that gq_hsq_levels / gq_hsq_decode_sum finish the compress / decompress
 * exactly as for the NearestNeighbor compressor.  Any d <= 104 and any K on the matrix cores (exact f32 MFMA, the
 * two sequential sums lane-local); beyond that d in {4,8,12,16,24,32,64} on the VALU kernel.
 */
int gq_pvq_encode(const float *grad, const float *c_dagger, int64_t M, int d, int K, int random_mode, const float *r,
                  uint64_t seed, void *codes, int code_bytes, float *u, float *workspace, void *stream);
/* Second stage of the ResidualCompressor (compressors/residual_compressor.py:15-24) WITHOUT a residual tensor: the
 * same encode applied to  grad - codebook1[codes1] * norm1  computed on the fly with the reference's roundings
 * (stage 1's decode, nearest_neighbor_compressor.py:85-89, then `residuals -= decompressed`).  norm1 [M] f32 is stage
 * 1's de-quantised norm per subvector.  GQ_ERR_UNSUPPORTED when d does not fit the LDS-staged kernel (d > 104). */
int gq_pvq_encode_residual(const float *grad, const void *codes1, int code1_bytes, const float *norm1,
                           const float *codebook1, const float *c_dagger, int64_t M, int d, int K, int random_mode,
                           const float *r, uint64_t seed, void *codes, int code_bytes, float *u, float *workspace,
                           void *stream);

/*
 * QSGD on a packed wire, multi-tensor form (one launch for all tensors; same arithmetic as
 * gq_qsgd_compress / gq_qsgd_decode_sum).  Per element one code = sign<<(bits-1) | level with
 * bits = gq_qsgd_code_bits(n_bit, random_mode): 4 (two codes per byte, element 2i in the low nibble)
 * when the top level is <= 7, 8 when it is <= 127, 16 (little-endian) when it is <= 32767, 0 = no packed
 * format.  Buckets are numbered across
 * tensors: bucket_seg int32[nbuckets]; seg_table int64[nseg][8] = { grad pointer (8-byte aligned),
 * d (even, <= 65536), first bucket, byte offset of the f32 norms / of the codes inside ONE user's wire,
 * float offset of the tensor in `out` (a multiple of 4), buckets, reserved }.  A zero bucket is written as level 0
 * (the reference's NaN level also decodes to 0).
 * gq_qsgd_compress_batched_ef: error feedback in the same pass (ps_quantizer.py:35-39):
 * seg_table[seg][7] = the tensor's error buffer (float*, 8-byte aligned; 0 = none); the bucket is read
 * as v = grad + ef_scale*error, v is written back over grad and error = v - decode(code) over error.
 */
int gq_qsgd_code_bits(int n_bit, int random_mode);
int gq_qsgd_compress_batched(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                             int n_bit, int random_mode, uint64_t seed, uint8_t *wire, void *stream);
int gq_qsgd_compress_batched_ef(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                                int n_bit, int random_mode, uint64_t seed, float ef_scale, uint8_t *wire,
                                void *stream);
int gq_qsgd_decode_sum_batched(const int64_t *seg_table, const int32_t *bucket_seg, int nseg, int64_t nbuckets,
                               int n_bit, int bits, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                               float *out, void *stream);

/*
 * The same packed wire for WIDE buckets -- the reference's TernGrad command (`--quantizer qsgd --c-dim 0
 * --n-bit 1`: one bucket spans the tensor, qsgd_compressor.py:15-16) or any c_dim of a few thousand and more.
 * The unit of work is a chunk of GQ_QSGD_WIDE_CHUNK consecutive elements of one bucket (the last chunk of a
 * bucket may be shorter; d must be even): chunk_seg int32[nchunks] names the tensor of each chunk, ascending;
 * seg_table int64[nseg][8] = { grad pointer (8-byte aligned), d, first chunk, byte offset of the norms / of the
 * codes inside ONE user's wire, float offset of the tensor in `out` (a multiple of 4), first word of the tensor's buckets in
 * norm_bits, error buffer (float*, 0 = none) }.  norm_bits (one uint32 per bucket; give every tensor its own
 * 128-byte line: the words are targets of atomics) must be zero before each
 * compress (max |v| is folded into it with integer atomics).  gq_qsgd_wide_compress = two launches (bucket
 * norms, then codes); ef != 0: error feedback as in gq_qsgd_compress_batched_ef.
 */
#define GQ_QSGD_WIDE_CHUNK 1024
int gq_qsgd_wide_compress(const int64_t *seg_table, const int32_t *chunk_seg, int nseg, int64_t nchunks, int n_bit,
                          int random_mode, uint64_t seed, int ef, float ef_scale, uint32_t *norm_bits, uint8_t *wire,
                          void *stream);
int gq_qsgd_wide_decode_sum(const int64_t *seg_table, const int32_t *chunk_seg, int nseg, int64_t nchunks, int n_bit,
                            int bits, const uint8_t *gathered, int64_t user_stride_bytes, int R, float *out,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GQ_HSQ_H */
