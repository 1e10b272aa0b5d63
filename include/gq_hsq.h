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
#define GQ_RANDOM_DEVICE_KEYED 3 /* gq_hsq_levels_batched / gq_qsgd_compress_batched only: as GQ_RANDOM_DEVICE, with every
                              tensor's (QSGD: bucket's) stream keyed by the bits of its (lb, ub) (QSGD: norm) besides
                              `seed`: new draws for every new gradient although `seed` stays what it was -- a
                              launch whose arguments never change (a HIP graph node)                              */
#define GQ_RANDOM_DEVICE_COUNTER 4 /* gq_hsq_levels_batched / gq_qsgd_compress_batched only: as GQ_RANDOM_DEVICE, but `seed`
                              is the ADDRESS of two device words { uint64 seed, uint64 step }: the launch's stream is keyed
                              by both, and gq_mean_rows adds one to `step` once per aggregate (apply).  The
                              launch's arguments never change (a HIP graph node), the draws do, whatever the gradients
                              are, and a run is reproducible from the two words' initial values; a `seed` that is 0 or not 8-byte
                              aligned is GQ_ERR_INVALID_ARG at both entry points
                              (probabilistic_scalar_compressor.py:22-26, qsgd_compressor.py:56-61 draw per call)          */

/* level_bytes of the level / decode entry points: 1 | 2 | 4 = one uint8 / uint16 / int32 per level, 0 = the f32
 * projections travel instead of levels (--n-bit 32), and
 * GQ_LEVELS_PACKED6 = four 6-bit levels per three bytes: group g (subvectors 4g .. 4g+3) is the 24-bit little-endian word
 *     l[4g] | l[4g+1] << 6 | l[4g+2] << 12 | l[4g+3] << 18      at byte 3g of the section (3 * ceil(M / 4) bytes),
 * for configurations whose top level is <= 63 (n_bit <= 6 without stochastic rounding, <= 5 with).  A byte per level
 * spends 8 bits on 6: this form takes 12.5 % off the (codes, levels) payload of the BASELINE configuration.  Served
 * for d = 16 with byte codes and K <= 256 (round 6: by the multi-tensor entry points too, which took K = 256 only); the decode is
 * the same arithmetic on the same integers (bit-identical).
 * The decode kernels fetch every group as ONE unaligned 32-bit word (three bytes of the group + the byte behind it), so a
 * packed section that is READ (gq_hsq_decode_sum, gq_hsq_decode_sum_strided, gq_hsq_batch_decode) must be followed by at
 * least one more readable byte: inside a wire the next section or the (lb, ub) words are; a section in a buffer of its own
 * needs 3 * ceil(M / 4) + 1 bytes of allocation.  Writers (gq_hsq_levels, gq_hsq_levels_decode) touch exactly 3 * ceil(M / 4) bytes. */
#define GQ_LEVELS_PACKED6 (-6)

/* OR-ed into the n_bit argument of gq_hsq_decode_sum / gq_hsq_decode_sum_strided and into gq_hsq_batch.n_bit for
 * gq_hsq_decode_sum_batched (opt-in; the quantizer: $GQ_AGGREGATE=fma): the mean over R >= 2 payloads accumulates
 * acc = fma(codeword element, norm, acc) instead of the reference's separately rounded product and sum (ps_quantizer.py:48
 * over nearest_neighbor_compressor.py:88): half the arithmetic per payload, and the aggregate is within 1e-6 relative L2 of the
 * bit-exact one (the north star grants 1e-5 on the decoded aggregate; codes, levels and every plain decompress are untouched).
 * Served by the d = 16 / byte-code kernels for R = 2, 4, 8, 16; every other case ignores the flag and stays bit-exact. */
#define GQ_AGGREGATE_FMA 0x100

/* Library / device identification. */
int gq_abi_version(void);            /* 5: round 6 (gq_launch_plan_create / _run / _destroy; no other prototype or struct changed: impl 6 of gq_hsq_encode_ex went -- GQ_ERR_INVALID_ARG -- and d = 12 / 24 joined the prefilter path); 4: round 5 (gq_hsq_decode_sum_batched_tail, gq_qsgd_decode_sum_batched_tail, gq_hsq_levels_decode_batched, gq_step_tail; impl 6 of gq_hsq_encode_ex; gq_qsgd_batch.reserved became bucket_hint: zero, what older callers pass, still means "unknown"); 3: round 4 (gq_mean_rows steps the GQ_RANDOM_DEVICE_COUNTER words and takes reset words; gq_hsq_batch / gq_qsgd_batch carry the dense table); 2: the round-3 descriptor form of the multi-tensor entry points */
const char *gq_last_error(void);     /* text of the calling thread's last failure (the library's only per-thread state) */
/* Fills CU count and the gcnArchName (e.g. "gfx950:sramecc+:xnack-") of `device`. */
int gq_device_info(int device, int *cu_count, char *arch, size_t arch_len);

/*
 * Per-dispatch timing of the dominant kernel (bench.py's roofline object).  An encode entry point that is handed
 * profile_slot >= 0 (gq_hsq_encode_ex; gq_hsq_batch.profile_slot) attaches a start / stop HIP event pair of that
 * slot to its d16/K256 prefilter kernel's dispatch (hipExtLaunchKernelGGL), so that the pair measures the kernel
 * alone -- an event bracket recorded around the call also measures ~5-8 us of queue bubbles.
 * gq_profile_read(slot, &ms) waits for that dispatch and returns its duration.  profile_slot = -1: a plain launch.
 */
#define GQ_PROFILE_SLOTS 64
int gq_profile_read(int slot, float *kernel_ms);

/*
 * Launch plans: the kernel launches of a captured HIP graph, issued as plain launches on a stream.  Not a piece of the reference's
 * arithmetic but of its LOOP (ps_quantizer.py:27-65 runs a step's operations one after the other on one stream): the quantizer
 * captures a step's launches once with stream capture -- whatever the configuration makes them -- and gq_launch_plan_run issues the
 * captured kernel nodes again, in order, with the arguments the graph holds.  Replaying the graph itself costs a boundary between
 * two replays that two launches on a stream do not have: 62.4 us per replay against 57.6 us for the two launches of the ResNet-50
 * step (tools/direct_vs_graph.py).  `hip_graph` is a hipGraph_t that must outlive the plan (it owns the argument storage);
 * GQ_ERR_UNSUPPORTED when the graph is not ONE chain of at most 64 kernel nodes (the caller then replays the graph as it is).
 */
int gq_launch_plan_create(void *hip_graph, void **plan, int *nodes);
int gq_launch_plan_run(void *plan, void *stream);
void gq_launch_plan_destroy(void *plan);

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

/* Same, with the kernel chosen explicitly (diagnostics / cross-checks; results are
 * identical for every impl):  0 = auto, 1 = exact f32 MFMA, d16/K256, register-resident
 * codebook, 2 = exact f32 MFMA generic (any d, K), 3 = VALU fmaf chain with the codebook in
 * LDS, 4 = f16 MFMA prefilter (one MFMA per chain and 16 dimensions) + exact f32 rescoring + second pass + exact scan
 * of what is left, for d in {8, 12, 16, 24, 32} and K <= 256 in multiples of 4 (the default for those shapes; bit-identical output.
 * Round 6: K <= 32 / <= 64 -- --k-bit 5 / 6 and K == dim, nearest_neighbor_compressor.py:40-47 -- score one / two row blocks
 * of 32 codewords instead of eight; every other K below 256 runs the eight-block kernel over zero rows.  K > 256 in
 * pages of 256: round 3's bf16 x 3 scheme), 5 = exact f32 MFMA with the codebook (chunked when it does not fit) and the
 * subvector tiles staged in LDS, any d <= 128 and any K (the default for every other shape; 2 remains the fallback for
 * d > 128).  (6, round 3's bf16 x 3 prefilter for K = 256, was removed in round 6: GQ_ERR_INVALID_ARG.)
 * profile_slot: see gq_profile_read. */
#define GQ_ENCODE_AUTO 0
#define GQ_ENCODE_MFMA_D16K256 1
#define GQ_ENCODE_MFMA_GENERIC 2
#define GQ_ENCODE_VALU 3
#define GQ_ENCODE_PREFILTER_D16K256 4
#define GQ_ENCODE_MFMA_LDS 5
int gq_hsq_encode_ex(const float *grad, const float *codebook, int64_t M, int d, int K, void *codes, int code_bytes,
                     float *u, float *workspace, int impl, int profile_slot, void *stream);

/*
 * Scalar level quantiser -- replaces probabilistic_scalar_compressor.py:12-27.
 *     lb = min(u), ub = max(u)    (finished here from the (min,max) pairs at the head of `workspace`)
 *     levels = 0                                   if lb - ub == 0
 *     x = |(u-lb)/(ub-lb)| * 2^n_bit ;  l = trunc(clamp(x, 0, 2^n_bit - 1))
 *     l += (x - l > r)                             if random_mode != GQ_RANDOM_OFF
 * Outputs lb_ub[2] (f32) and levels[M] (level_bytes 1/2/4 -> uint8/uint16/int32, GQ_LEVELS_PACKED6 -> 3 * ceil(M/4)
 * bytes; the value range is [0, 2^n_bit] with stochastic rounding, [0, 2^n_bit - 1] without).
 * The whole compress of NearestNeighborCompressor (nearest_neighbor_compressor.py:63-78) is gq_hsq_encode followed
 * by gq_hsq_levels on the same stream.
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
 * comes out as +0.  The multi-tensor decodes take the choice as an argument (`plain`).
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

/* Levels + decode of ONE payload in a single launch: decompress(compress(g)) from the encode's outputs --
 * nearest_neighbor_compressor.py:74-78 (u -> ProbabilisticScalarCompressor.compress, probabilistic_scalar_compressor.py:12-27)
 * followed by :80-90 (decompress), i.e. what PSQuantizer.record computes for a user (ps_quantizer.py:37) and what a
 * single-rank step runs after its encode.  Same results as gq_hsq_levels followed by gq_hsq_decode_sum with R = 1 (lb_ub,
 * the level section and out, bit for bit); one launch less.  d = 16 with byte codes; level_bytes 1 or GQ_LEVELS_PACKED6;
 * K <= 256; u / codes / levels 4-byte aligned, out / codebook 16-byte aligned -- GQ_ERR_UNSUPPORTED otherwise (use the two calls). */
int gq_hsq_levels_decode(const float *u, int64_t M, int n_bit, int random_mode, const float *r, uint64_t seed,
                         const float *minmax_partials, float *lb_ub, void *levels, int level_bytes, const void *codes,
                         const float *codebook, int K, float *out, void *stream);

/*
 * Multi-tensor forms: ONE launch per step and operation serves every tensor of a model that shares a codebook (the
 * reference loops over parameters in Python, ps_quantizer.py:33,47).  lb / ub stay per tensor.  The tensors and the
 * configuration are described by ONE struct that the caller fills once (round 2 had six spellings of the encode and
 * two per-thread "next call" flags; all of that is a field or an argument now):
 *
 *   Every tensor is padded to whole 64-subvector tiles in a common index space of `ntiles` tiles:
 *   tile_seg   int32[ntiles]       tensor ("segment") of each tile, ascending
 *   seg_table  int64[nseg][8]      { grad pointer (16-byte aligned when d % 4 == 0, else 4), M subvectors, first tile,
 *                                    byte offset of codes / of levels / of (lb,ub) inside ONE user's wire (sections
 *                                    16-byte aligned), float offset of the tensor in `out`,
 *                                    error buffer (float *, aligned like grad; 0 = no error feedback for this tensor) }
 *   seg_minmax uint32[nseg][2]     order-mapped (min,max) of u; the caller resets it before each encode to
 *                                  { 0xFFFFFFFF, 0 }
 *   u_flat     float[ntiles*64]    projection spill, padded index space
 *   workspace  gq_hsq_workspace_bytes(ntiles*64) bytes (same contract as gq_hsq_encode); may be NULL when
 *              gq_hsq_batched_path() says GQ_BATCH_EXACT
 *
 * gq_hsq_batched_path(b): which kernels serve (d, K, code_bytes, level_bytes, nseg) --
 *   GQ_BATCH_PREFILTER  K <= 256 (a multiple of 4; round 5: K = 256 only), d in {8, 16, 32}, byte codes: f16 prefilter + exact rescoring +
 *                       second pass (any number of tensors); K <= 32 / <= 64 score one / two row blocks of 32 codewords;
 *                       round 6: d = 12 / 24 too (the reference's repaired dimensions, nearest_neighbor_compressor.py:23-29) as
 *                       rows of 12 / 24 floats through the d = 16 / 32 kernels; their level / decode launches are the generic ones
 *   GQ_BATCH_PAGED      d in {8, 16, 32}, K = 512, 768, ... 65536, int32 codes, <= 384 tensors: the prefilter with the
 *                       pages of 256 codewords resident in LDS (an earlier page keeps a tie: the first maximum)
 *   GQ_BATCH_EXACT      any other d <= 104 and K: exact f32 MFMA scoring, codebook chunked in LDS when large
 *   0                   not served (the caller takes the per-tensor entry points); gq_last_error() says why
 * Results are identical to the per-tensor entry points on every path.
 */
#define GQ_BATCH_PREFILTER 1
#define GQ_BATCH_PAGED 2
#define GQ_BATCH_EXACT 3
typedef struct gq_hsq_batch {
    uint32_t struct_bytes;     /* sizeof(gq_hsq_batch): a caller built against another layout is refused */
    int32_t d, K;              /* nearest_neighbor_compressor.py:23-43 */
    int32_t code_bytes;        /* 1 (K <= 256) | 4 */
    int32_t level_bytes;       /* 1 | 2 | 4, GQ_LEVELS_PACKED6, or 0: the f32 projections travel (--n-bit 32, nearest_neighbor_compressor.py:14,75-76) */
    int32_t n_bit;             /* probabilistic_scalar_compressor.py:7 */
    int32_t nseg;
    int32_t profile_slot;      /* -1, or the gq_profile_read slot the d16/K256 encode's dispatch is timed into */
    int64_t ntiles;
    const int64_t *seg_table;
    const int32_t *tile_seg;
    const float *codebook;     /* [K, d] f32, row-normalised by the caller (utils/vec_np.py:4-10) */
    float *u_flat;
    uint32_t *seg_minmax;
    float *workspace;
    /* The tensors that travel uncompressed (IdenticalCompressor: ps_quantizer.py:18-19, <= 1000 elements each) ride in the
     * level launch: dense_table int64[ndense][3] = { source (float *, device), byte offset in ONE user's wire (a multiple
     * of 4), elements }; gq_hsq_levels_batched copies every source to wire + offset.  ndense == 0: nothing to copy. */
    const int64_t *dense_table;
    int32_t ndense;
    int32_t reserved;
} gq_hsq_batch;
int gq_hsq_batched_path(const gq_hsq_batch *b);

/* Encode of every tensor into ONE user's `wire` (codes), u into u_flat, (min,max) into seg_minmax --
 * nearest_neighbor_compressor.py:65-73 per tensor.  ef_scale: NaN = no error feedback; otherwise every tile is read as
 * v = grad + ef_scale*error (ps_quantizer.py:35: product rounded, then the add), v is written back over grad like the
 * reference's in-place add_, and v is encoded (rows whose seg_table[seg][7] is 0 are encoded as they are). */
int gq_hsq_encode_batched(const gq_hsq_batch *b, uint8_t *wire, float ef_scale, void *stream);

/* Levels and (lb, ub) of every tensor into `wire` -- probabilistic_scalar_compressor.py:12-27 per tensor.
 * r_flat: the caller's draws for random_mode = GQ_RANDOM_GIVEN, laid out like u_flat (r_flat[tile * 64 + i] belongs to
 * subvector i of tile `tile`; padding slots are not read), NULL otherwise.  Reference parity: r = torch.rand(M) per
 * tensor from the CPU generator (probabilistic_scalar_compressor.py:23-26) is one sequential stream, so ONE
 * torch.rand(sum of M) per record() equals the reference's per-tensor calls in parameter order PROVIDED every tensor
 * draws -- the reference returns before torch.rand when lb == ub (:15-16), see DESIGN.md.
 * write_error != 0 (after an encode with error feedback): additionally error = v - decode(wire) over the old error,
 * ps_quantizer.py:37-39, for the rows that have an error buffer. */
int gq_hsq_levels_batched(const gq_hsq_batch *b, uint8_t *wire, int random_mode, uint64_t seed, const float *r_flat,
                          int write_error, void *stream);

/* Decode + mean of R users' wires (`gathered` + r*user_stride_bytes) into `out` -- ps_quantizer.py:47-48.
 * plain != 0: the plain decompress of ONE payload as the reference returns it (a -0 stays -0) -- the ring's hop and
 * final gradient (ring_quantizer.py:32,41-47), the two-phase / error-feedback round trips (ps_quantizer.py:37,52-61);
 * plain == 0: the aggregate (+0 + sum) / R, also for R == 1.  Only the sign of zeros differs.
 * `plain` is a flags word: bit 0 as above; bit 1 (value 2, with bit 0 clear): GQ_AGGREGATE_FMA for this launch.
 * Only d, K, the widths, n_bit, nseg, ntiles, seg_table, tile_seg and codebook of `b` are read: a decode of a PART of
 * the tensors (split exchange) passes a copy of the struct with another table. */
int gq_hsq_decode_sum_batched(const gq_hsq_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                              float *out, int plain, void *stream);

/* The same launch may take the aggregate's small per-step work along -- everything gq_mean_rows (below) does: the mean of the
 * uncompressed tensors' rows over rows_R payloads (ps_quantizer.py:18,48), one step of the GQ_RANDOM_DEVICE_COUNTER words, the
 * reset of the accumulators the next step's kernels fold into.  All of it depends on EARLIER launches only.  A step of the
 * ResNet-50 list is then three kernels (encode, levels, decode-mean) instead of four; a launch of its own cost ~4 us of
 * kernel and a boundary in a ~70 us step.  Decode paths without the in-kernel form (exact kernels, unaligned wires) run
 * gq_mean_rows behind the decode: the results are the same either way.  t == NULL: gq_hsq_decode_sum_batched. */
#define GQ_TICKET_WORDS 544   /* (1 + 16) counters of the last-workgroup hand-over, each on a 128-byte line of its own */
typedef struct gq_step_tail {
    uint32_t struct_bytes;       /* sizeof(gq_step_tail) */
    int32_t rows_R;              /* rows of the mean (>= 1) */
    const void *rows;            /* row r = (const float *)((const char *)rows + r * row_stride_bytes); n == 0: no mean */
    int64_t row_stride_bytes;
    int64_t n;
    float *out;                  /* float[n] */
    uint64_t *rng_state;         /* NULL or rng_pairs (1 .. 256) consecutive { uint64 seed, uint64 step } pairs: step += 1 */
    uint64_t *reset_dst;         /* reset_words 64-bit words reset_src -> reset_dst (0: none) */
    const uint64_t *reset_src;
    int32_t rng_pairs;
    int32_t reset_words;
    uint32_t *ticket;            /* gq_hsq_levels_decode_batched only: GQ_TICKET_WORDS device words, zero before the first launch (the
                                    launch leaves them zero); else NULL */
} gq_step_tail;
int gq_hsq_decode_sum_batched_tail(const gq_hsq_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                   float *out, int plain, const gq_step_tail *t, void *stream);

/* One rank, one user per step: decompress(compress(g)) for every tensor is the aggregate (ps_quantizer.py:37,48 with one
 * decoded tensor per parameter).  This is gq_hsq_levels_batched followed by gq_hsq_decode_sum_batched_tail over the ONE
 * payload `wire` as ONE launch -- the multi-tensor twin of gq_hsq_levels_decode: levels and (lb, ub) into the wire (+ the
 * residual when write_error), out = (+0 + codebook[code] * norm) / 1 (plain: the decompress as it is), the uncompressed
 * tensors copied into the wire AND averaged (t->rows points at the wire's dense region, rows_R = 1), and -- by the last
 * workgroup to finish, told by t->ticket -- the step of the draws' words and the accumulators' reset.  A step of the
 * ResNet-50 list is then two kernels.  Served in one launch for K <= 256, d in {8, 16, 32}, byte codes, byte or 16-bit levels;
 * every other descriptor runs the two calls (same results). */
int gq_hsq_levels_decode_batched(const gq_hsq_batch *b, uint8_t *wire, int random_mode, uint64_t seed, const float *r_flat,
                                 int write_error, float *out, int plain, const gq_step_tail *t, void *stream);

/*
 * Error-feedback helpers around the per-tensor codec (ps_quantizer.py:35,39):
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
/* rng_state != NULL (GQ_RANDOM_DEVICE_COUNTER): the same launch adds one to the step words of rng_pairs (1 .. 256)
 * consecutive { uint64 seed, uint64 step } pairs -- a caller keeps one pair per (tensor group, user slot) and steps them
 * once per aggregate; with n == 0 that is all the call does. */
/* reset_words > 0: the same launch also copies reset_words 64-bit words reset_src -> reset_dst (device memory): the
 * accumulators the next step's multi-tensor kernels fold into (gq_hsq_batch.seg_minmax, wide QSGD norm words) go back to
 * their empty state in the step's LAST launch instead of a copy in front of its first -- what a replayed HIP graph of a
 * whole step wants (every extra node costs ~4 us). */
int gq_mean_rows(const void *rows, int64_t row_stride_bytes, int R, int64_t n, float *out, uint64_t *rng_state, int rng_pairs,
                 uint64_t *reset_dst, const uint64_t *reset_src, int reset_words, void *stream);

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
 * QSGD on a packed wire, multi-tensor form (one launch for all tensors; same arithmetic as
 * gq_qsgd_compress / gq_qsgd_decode_sum).  Per element one code = sign<<(bits-1) | level with
 * bits = gq_qsgd_code_bits(n_bit, random_mode): 4 (two codes per byte, element 2i in the low nibble)
 * when the top level is <= 7, 8 when it is <= 127, 16 (little-endian) when it is <= 32767, 0 = no packed format.
 * A zero bucket is written as level 0 (the reference's NaN level also decodes to 0).
 *
 * wide == 0: the unit of work is a bucket; buckets are numbered across tensors: item_seg int32[nitems] names the
 *   tensor of each bucket; seg_table int64[nseg][8] = { grad pointer (8-byte aligned), d (even, <= 65536), first
 *   bucket, byte offset of the f32 norms / of the codes inside ONE user's wire, float offset of the tensor in `out`
 *   (a multiple of 4), buckets, error buffer (float *, 0 = none) }.
 * wide != 0: WIDE buckets -- the reference's TernGrad command (`--quantizer qsgd --c-dim 0 --n-bit 1`: one bucket
 *   spans the tensor, qsgd_compressor.py:15-16) or any bucket of about a thousand elements and more (the caller's choice:
 *   both forms take any d; gq_amd/codecs.py sends buckets of >= 1,024 elements here).  The unit of work is a chunk
 *   of GQ_QSGD_WIDE_CHUNK consecutive elements of one bucket (the last chunk of a bucket may be shorter; d must be
 *   even): item_seg names the tensor of each chunk, ascending; seg_table int64[nseg][8] = { grad pointer, d, first
 *   chunk, byte offset of the norms / of the codes, float offset in `out`, first word of the tensor's buckets in
 *   norm_bits, error buffer }.  norm_bits (one uint32 per bucket; give every tensor its own 128-byte line: the words
 *   are targets of atomics) must be zero before each compress (max |v| is folded into it with integer atomics); the
 *   compress is two launches (bucket norms, then codes).
 * ef_scale: NaN = none; otherwise error feedback in the same pass (ps_quantizer.py:35-39): the bucket is read as
 *   v = grad + ef_scale*error, v is written back over grad and error = v - decode(code) over error.
 * plain: as gq_hsq_decode_sum_batched.
 */
#define GQ_QSGD_WIDE_CHUNK 1024
typedef struct gq_qsgd_batch {
    uint32_t struct_bytes;     /* sizeof(gq_qsgd_batch) */
    int32_t n_bit;             /* qsgd_compressor.py:10 */
    int32_t bits;              /* gq_qsgd_code_bits(n_bit, random_mode) */
    int32_t wide;
    int32_t nseg;
    int32_t bucket_hint;       /* the bucket width most elements have, 0 = unknown (was `reserved`): picks how many lanes the
                                  bucketed kernels give a bucket (d / 8, between 2 and 16); every value gives the same results */
    int64_t nitems;
    const int64_t *seg_table;
    const int32_t *item_seg;
    uint32_t *norm_bits;       /* wide only */
    const int64_t *dense_table; /* as gq_hsq_batch.dense_table: copied by gq_qsgd_compress_batched's launch */
    int32_t ndense;
    int32_t reserved2;
} gq_qsgd_batch;
int gq_qsgd_code_bits(int n_bit, int random_mode);
int gq_qsgd_compress_batched(const gq_qsgd_batch *b, uint8_t *wire, int random_mode, uint64_t seed, float ef_scale,
                             void *stream);
int gq_qsgd_decode_sum_batched(const gq_qsgd_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                               float *out, int plain, void *stream);
/* ... with the aggregate's small per-step work riding in the same launch (gq_step_tail: see gq_hsq_decode_sum_batched_tail). */
int gq_qsgd_decode_sum_batched_tail(const gq_qsgd_batch *b, const uint8_t *gathered, int64_t user_stride_bytes, int R,
                                    float *out, int plain, const gq_step_tail *t, void *stream);

/*
 * ProbabilisticVectorCompressor encode -- probabilistic_vector_compressor.py:42-63 (pinned by the reference's own
 * output):
 *     p = c_dagger . v   (c_dagger = pinv(codewords^T), [K,d]) ;  l1 = sum_k |p_k|
 *     code = first k with cumsum_k(|p|/l1) >= r - 1e-5 ;  u = sign(p_code) * l1
 * r: one uniform draw per subvector (GQ_RANDOM_GIVEN: caller-supplied r[M]; GQ_RANDOM_DEVICE: in-kernel).
 * Outputs codes[M], u[M] and the (min,max) partials of u in `workspace` (gq_hsq_workspace_bytes(0)
 * bytes suffice) so that gq_hsq_levels / gq_hsq_decode_sum finish the compress / decompress
 * exactly as for the NearestNeighbor compressor.  Any d <= 104 and any K on the matrix cores (exact f32 MFMA, the
 * two sequential sums lane-local); beyond that d in {4,8,12,16,24,32,64} on the VALU kernel.
 * stage1 != NULL: the second stage of the ResidualCompressor (compressors/residual_compressor.py:15-24) WITHOUT a
 * residual tensor -- the same encode applied to  grad - codebook1[codes1] * norm1  computed on the fly with the
 * reference's roundings (stage 1's decode, nearest_neighbor_compressor.py:85-89, then `residuals -= decompressed`);
 * norm1 [M] f32 is stage 1's de-quantised norm per subvector.  GQ_ERR_UNSUPPORTED when d does not fit the
 * LDS-staged kernel (d > 104).
 */
typedef struct gq_pvq_stage1 {
    const void *codes1;
    int32_t code1_bytes;
    int32_t reserved;
    const float *norm1;
    const float *codebook1;
} gq_pvq_stage1;
int gq_pvq_encode(const float *grad, const float *c_dagger, int64_t M, int d, int K, int random_mode, const float *r,
                  uint64_t seed, void *codes, int code_bytes, float *u, float *workspace, const gq_pvq_stage1 *stage1,
                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GQ_HSQ_H */
