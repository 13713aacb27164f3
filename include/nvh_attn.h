/*
 * nvh_attn.h — C ABI of the MI355X (gfx950) paged-attention backend for nano-vllm.
 *
 * This is the drop-in boundary for `--attn-backend hip`: the entry points below are exactly
 * what a reference-side binding (ctypes) calls from
 * nanovllm/layers/attention.py's `Attention.forward`.  Plain C, raw device pointers and sizes,
 * no torch / pybind types.  One process per GPU, single caller thread per process.
 *
 * Contract:
 *   - The caller owns ALL memory, including the decode workspace.  The library allocates nothing,
 *     frees nothing and keeps no state besides a thread-local error string.
 *   - Every call only enqueues kernels on `stream` (a hipStream_t; NULL = default stream).  No host
 *     synchronisation, no host reads of device data, grid sizes are functions of the static
 *     arguments only -> every call is legal inside HIP-graph capture
 *     (nanovllm/engine/model_runner.py:316-370 captures the decode forward).
 *   - Return 0 on success; a negative NVH_E_* code for rejected arguments; a positive hipError_t
 *     for a failed launch.  nvh_last_error() gives the message.  Nothing throws across the ABI.
 *   - K/V cache layout is the reference's: [num_blocks, block_size, num_kv_heads, head_dim],
 *     contiguous (model_runner.py:144-145, asserts attention.py:51-54).  All cache offsets are
 *     64-bit inside the kernels.
 *   - Element type of q/k/v/caches: bf16 (NVH_BF16), the reference's runtime dtype
 *     (model_runner.py:35).  Outputs are bf16, or fp32 (NVH_F32) for pre-rounding parity checks.
 *   - softmax scale is passed explicitly and must be head_dim**-0.5 to match the reference's
 *     flash / sdpa backends (attention.py:96,101).
 */
#ifndef NVH_ATTN_H
#define NVH_ATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVH_VERSION 205          /* major*100 + minor; 201: nvh_allreduce_status; 202: nvh_linear_desc.prefetch; 203: NVH_DECODE_CHUNKED_P256; 204: nvh_qkv_rope_attend; 205: nvh_prefill_varlen_pv16 */

/* dtype codes */
#define NVH_BF16 0
#define NVH_F32  1

/* argument errors (negative so they never collide with hipError_t) */
#define NVH_E_DTYPE      (-1)    /* unsupported element type */
#define NVH_E_SHAPE      (-2)    /* unsupported head_dim / group size / block_size */
#define NVH_E_STRIDE     (-3)    /* stride not a multiple of 8 elements (16-byte vector access) */
#define NVH_E_WORKSPACE  (-4)    /* workspace too small */
#define NVH_E_NULL       (-5)    /* required pointer is NULL */
#define NVH_E_ALIGN      (-6)    /* pointer not 16-byte aligned */

int nvh_version(void);
const char* nvh_last_error(void);

/*
 * Scatter new K/V rows into the paged cache.
 * Replaces: store_kvcache + store_kvcache_kernel, nanovllm/layers/attention.py:19-55
 *           (slot < 0 rows are skipped, as nanovllm/layers/attention_triton.py:29-31 does for -1).
 *   k, v           [n_tokens, kvh, hd]; inner two dims contiguous, row strides in ELEMENTS
 *                  (v is a strided view of the fused qkv projection, models/qwen3.py:104-106)
 *   k_cache/v_cache [num_blocks, block_size, kvh, hd] contiguous; row `slot` = slot*kvh*hd elements
 *   slot_mapping   int32 [n_tokens], slot = block_id*block_size + offset (model_runner.py:212-221,254-258)
 * n_tokens == 0 is a no-op.
 */
int nvh_store_kvcache(const void* k, const void* v, void* k_cache, void* v_cache,
                      const int32_t* slot_mapping, int n_tokens, int kvh, int hd,
                      int64_t k_row_stride, int64_t v_row_stride, int dtype, void* stream);

/*
 * Bytes of caller-owned scratch nvh_paged_decode needs; a pure function of the static shapes so it can be
 * allocated once before graph capture.  Layout: a fixed 64 KiB header of arrival tickets (one uint32 per
 * (sequence, kv head), each on a 128-byte line of its own: 512 pairs can be split over several workgroups, far more than the
 * device has CUs for), 8 KiB of counters for nvh_qkv_rope_attend's hand-off (ready / done per kv head, one per line; a status word),
 * then the partial records of the context chunks (each padded to a 256-byte boundary).  The caller ZERO-FILLS the buffer once
 * (hipMemset / torch.zeros); every launch returns its tickets to zero, so the buffer is reusable across calls,
 * shapes and graph replays without further clearing.  ONE launch at a time per workspace: calls that may run concurrently
 * (different streams) need a workspace each; a launch that was aborted mid-flight leaves the tickets undefined (zero-fill again).
 */
size_t nvh_paged_decode_workspace(int batch, int h, int hd, int max_blocks, int block_size);

/*
 * Decode: one query token per sequence attends its whole paged context.
 * Replaces: flash_attn_with_kvcache(q.unsqueeze(1), k_cache, v_cache, cache_seqlens=context_lens,
 *           block_table=block_tables, softmax_scale, causal=True), nanovllm/layers/attention.py:99-101
 *           (oracle body nanovllm/layers/attention_sdpa.py:122-182).
 *   out            [batch, h, hd]  (out_dtype NVH_BF16 or NVH_F32), contiguous
 *   q              [batch, h, hd], contiguous per row; q_row_stride in elements
 *   block_tables   int32 [batch, max_blocks], row stride bt_row_stride (elements); entries at or past
 *                  ceil(ctx/block_size) are never dereferenced (they are -1 in eager mode and 0 under
 *                  graph replay, model_runner.py:160-169,299)
 *   context_lens   int32 [batch]; 0 -> the row's output is zeros (graph padding rows)
 *   h % kvh == 0, h/kvh <= 16, hd in {64, 128},
 *   block_size a multiple of 64.
 */
int nvh_paged_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                     const int32_t* block_tables, const int32_t* context_lens,
                     int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                     int64_t q_row_stride, int64_t bt_row_stride, float scale,
                     int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same call through a chosen formulation of the decode kernel (tests and A/B measurements; nvh_paged_decode always runs
 * NVH_DECODE_CHUNKED with waves = chunks = 0).  All variants compute the same function and are held to the same parity bar:
 *   NVH_DECODE_CHUNKED     one launch: MFMA tiles over the GQA group, split-KV passes dealt to `chunks` workgroups per
 *                          (sequence, kv head), last-arriver combine.  waves per workgroup: 4 or 8; 0 = the default (8 at hd 64, 4 at hd 128).
 *                          chunks: 0 = one wave of workgroups over the device's CUs, > 0 = that many (clamped to the passes).
 *                          hd 64: passes of 256 tokens, or of 128 where a pair is split over 3-5 workgroups (few passes per workgroup:
 *                          the finer passes balance the workgroups better; measured -2.5 % on average there, +1..12 % with more chunks)
 *   NVH_DECODE_CHUNKED_P128 / _P256 / _P64  the chunked kernel with the pass size forced: hd 64 takes 128 or 256 (16- / 32-token wave tiles),
 *                          hd 128 takes 128 or 64 (four waves x 32- / 16-token tiles; the default picks 64 where a pair is split over at most
 *                          8 workgroups: -5 % at Qwen2-7B's per-rank shape 7/1/128, bs 32); a size the head_dim does not have is ignored
 *   NVH_DECODE_SPLIT_MFMA  the single-pass MFMA split kernel + a combine launch (flash-decoding in two launches).
 *   NVH_DECODE_SPLIT_VALU  north_star's literal form: VALU dot products with wavefront-level (DPP / permlane) max and sum
 *                          reductions, no MFMA; groups of at most 8 query heads per kv head; + the combine launch.
 */
enum { NVH_DECODE_CHUNKED = 0, NVH_DECODE_SPLIT_MFMA = 1, NVH_DECODE_SPLIT_VALU = 2, NVH_DECODE_CHUNKED_P128 = 3, NVH_DECODE_CHUNKED_P256 = 4, NVH_DECODE_CHUNKED_P64 = 5 };
int nvh_paged_decode_variant(int variant, int waves, int chunks, void* out, const void* q, const void* k_cache, const void* v_cache,
                             const int32_t* block_tables, const int32_t* context_lens,
                             int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                             int64_t q_row_stride, int64_t bt_row_stride, float scale,
                             int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * nvh_paged_decode that ALSO writes the bf16 result in MFMA-fragment order (out_packed: [ceil(batch/16)][h*hd/32][64][8],
 * nvh_pack_index(row = sequence, col = head*hd + dim, cols = h*hd)) for a following nvh_linear_small_m_ex with x_packed = 1
 * (the output projection, models/qwen3.py:118).  Same arguments otherwise; `out` is still written.
 */
int nvh_paged_decode_packed(void* out, void* out_packed, const void* q, const void* k_cache, const void* v_cache,
                            const int32_t* block_tables, const int32_t* context_lens,
                            int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                            int64_t q_row_stride, int64_t bt_row_stride, float scale,
                            int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Fused decode step for one layer: store this step's K/V row of every sequence, then attend.
 * Replaces the pair of calls at nanovllm/layers/attention.py:84-86 and :99-101 with one launch
 * sequence; result identical to nvh_store_kvcache followed by nvh_paged_decode.
 *   k_new, v_new   [batch, kvh, hd] with row strides in elements; slot_mapping int32 [batch]
 */
int nvh_decode_step(void* out, const void* q, const void* k_new, const void* v_new,
                    void* k_cache, void* v_cache, const int32_t* slot_mapping,
                    const int32_t* block_tables, const int32_t* context_lens,
                    int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                    int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                    int64_t bt_row_stride, float scale, int dtype, int out_dtype,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * Prefill: packed variable-length causal attention.
 * Replaces: flash_attn_varlen_func(q, k, v, cu_seqlens_q/k, max_seqlen_q/k, softmax_scale,
 *           causal=True, block_table=...), nanovllm/layers/attention.py:93-96
 *           (oracle body nanovllm/layers/attention_sdpa.py:65-119).
 *   block_tables == NULL : k, v are the new tokens [total_k, kvh, hd] with row strides (elements).
 *   block_tables != NULL : k, v are the paged caches (prefix-cache prefill, attention.py:90-91);
 *                          k/v_row_stride are ignored; sequence i's keys are found through
 *                          block_tables[i, :]; block_size/max_blocks/bt_row_stride describe it.
 *   cu_seqlens_q/k int32 [batch+1] on device; max_seqlen_q bounds the launch grid (host int, as in
 *                  model_runner.py:204-207).  The causal mask is bottom-right aligned: query row r
 *                  of a sequence sees keys 0 .. r + (Sk - Sq)  (flash-attn semantics; equals the
 *                  oracle's top-left mask whenever Sq == Sk).
 *                  CONTRACT: max_seqlen_q / max_seqlen_k must bound every sequence in cu_seqlens_q/k (as the reference's
 *                  runner guarantees).  The library cannot check device data without a host sync; rows / keys beyond an
 *                  understated maximum are not processed (never an out-of-bounds access).
 *   out            [total_q, h, hd] contiguous, out_dtype NVH_BF16 or NVH_F32.
 */
int nvh_prefill_varlen(void* out, const void* q, const void* k, const void* v,
                       const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                       const int32_t* block_tables, int batch, int max_seqlen_q, int max_seqlen_k,
                       int h, int kvh, int hd, int block_size, int max_blocks,
                       int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                       int64_t bt_row_stride, float scale, int dtype, int out_dtype, void* stream);

/*
 * The same call with the kernel chosen by the caller (tests and A/B measurements; nvh_prefill_varlen passes 0, 0):
 *   kernel       NVH_PREFILL_AUTO, NVH_PREFILL_TILED (64-row query tiles, K/V tiles double-buffered through LDS) or
 *                NVH_PREFILL_SHORT (one workgroup per (sequence, kv head), K/V resident in LDS: needs max_seqlen_q <=
 *                max_seqlen_k <= 128 and block_tables == NULL, otherwise an error)
 *                NVH_PREFILL_TILED_F16V (a MEASUREMENT variant, not a drop-in: block_tables must be NULL): `v` holds the values as IEEE fp16 rows,
 *                converted by the caller (same layout and strides); P is rounded to fp16 and P V runs as one fp16 MFMA per operand pair instead
 *                of the bf16 hi + lo pair.  Error 4.5e-4 on the reference goldens (hi + lo: 6e-6), |v| <= 65504 required; DESIGN.md section 12.2
 *   short_waves  waves per workgroup of the short-sequence kernel: 0 = auto, 8 or 16
 */
enum { NVH_PREFILL_AUTO = 0, NVH_PREFILL_TILED = 1, NVH_PREFILL_SHORT = 2, NVH_PREFILL_TILED_F16V = 3 };
int nvh_prefill_varlen_variant(int kernel, int short_waves, void* out, const void* q, const void* k, const void* v,
                               const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                               const int32_t* block_tables, int batch, int max_seqlen_q, int max_seqlen_k,
                               int h, int kvh, int hd, int block_size, int max_blocks,
                               int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                               int64_t bt_row_stride, float scale, int dtype, int out_dtype, void* stream);

/* Helper of the NVH_PREFILL_TILED_F16V measurement form: bf16 rows [n_rows, row_elems] (row strides in elements) -> IEEE fp16 rows, exact for
 * 2^-24 <= |x| <= 65504 (larger finite magnitudes are clamped: the caller must know its V stays in range — nvh_prefill_varlen_pv16 below checks it).
 * 16-byte aligned rows, row_elems % 8 == 0. */
int nvh_bf16_rows_to_f16(void* out, const void* in, int n_rows, int row_elems, int64_t in_row_stride, int64_t out_row_stride, void* stream);

/*
 * nvh_prefill_varlen (block_tables == NULL) with P V on the fp16 matrix pipe — same arguments, same result within the parity bar, 1.2-1.4x faster from 1024
 * keys per sequence on (the tile loop is bound by vector-instruction issue and the bf16 hi + lo split of P is a third of it).
 * Replaces: flash_attn_varlen_func, nanovllm/layers/attention.py:93-96 (whose own P is a single bf16: 8 significant bits against the 11 used here).
 * Two launches, no host read, nothing to clear, capture-safe: V is converted to fp16 rows in `scratch` (exact for 2^-24 <= |v| <= 65504; smaller magnitudes
 * become 0) and every 64-row group gets a range flag (set when a finite value does not fit fp16); the attention kernel ORs the flags that cover a
 * sequence's rows and runs fp16 P x fp16 V for it, or — a flag set, or the sequence reaching past total_k — the exact bf16 hi + lo form on the caller's own
 * V rows, i.e. exactly nvh_prefill_varlen for that sequence.  Numerics of the fp16 form: P in [0, 1] rounded to 11 significant bits, products accumulated
 * in fp32: |error| <= 2^-12 * max|v| per output element, 4.5e-4 on the reference's golden vectors (N(0,1) values; hi + lo: 6e-6; bf16 OUTPUT rounding
 * alone: 2^-9 * |o|).  Shapes the short-sequence kernel takes (max_seqlen_q <= max_seqlen_k <= 128, >= 128 (sequence, kv head) pairs:
 * nvh_prefill_pv16_uses_scratch() == 0, scratch may be NULL) run ONE launch: at head_dim 64 and more than 64 keys each workgroup converts its resident
 * V images to fp16 in LDS, range check folded into the barrier that publishes them (a (sequence, kv head) pair out of range keeps hi + lo); otherwise
 * that kernel runs unchanged.
 *   total_k        rows of k / v (= cu_seqlens_k[batch] on the host)
 *   scratch        device memory of at least nvh_prefill_pv16_scratch_bytes(total_k, kvh, hd) bytes (flags: 4 * ceil(total_k / 64) rounded up to 256, then
 *                  total_k*kvh*hd*2), 16-byte aligned, owned by the caller, not shared with a call in flight on another stream
 */
size_t nvh_prefill_pv16_scratch_bytes(int total_k, int kvh, int hd);
int nvh_prefill_pv16_uses_scratch(int batch, int max_seqlen_q, int max_seqlen_k, int kvh, int hd);
int nvh_prefill_varlen_pv16(void* out, const void* q, const void* k, const void* v,
                            const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int batch, int max_seqlen_q, int max_seqlen_k, int total_k,
                            int h, int kvh, int hd, int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                            float scale, int dtype, int out_dtype, void* scratch, size_t scratch_bytes, void* stream);

/*
 * "Next" row: the step immediately before attention, fused into one launch.
 * (optional per-head RMSNorm ->) neox RoPE on q and k IN PLACE inside the fused qkv projection output, then
 * k (rotated) and v rows are stored to the paged cache.
 * Replaces: q_norm/k_norm (nanovllm/models/qwen3.py:108-114, layers/layernorm.py:17-27), rotary_emb
 *           (layers/rotary_embedding.py:39-55) and store_kvcache (layers/attention.py:84-86).
 *   qkv            [n_tokens, (h + 2*kvh) * hd] bf16, row stride qkv_row_stride (elements): q heads | k heads | v heads
 *   positions      int64 [n_tokens]; cos_sin fp32 [max_position, hd] = cos(0..hd/2) | sin(0..hd/2)
 *                  (the reference's cos_sin_cache, rotary_embedding.py:29-36)
 *   q_norm_w/k_norm_w  bf16 [hd] or NULL (both or neither); eps as in RMSNorm
 *   k_cache/v_cache/slot_mapping  as nvh_store_kvcache; NULL caches or slot < 0 -> no store for that row
 */
int nvh_rope_store(void* qkv, const int64_t* positions, const float* cos_sin,
                   const void* q_norm_w, const void* k_norm_w, float eps,
                   void* k_cache, void* v_cache, const int32_t* slot_mapping,
                   int n_tokens, int h, int kvh, int hd, int64_t qkv_row_stride, int dtype, void* stream);

/*
 * Engine widening (not part of the attention parity bar): the row-wise elementwise ops either side of the attention
 * block, one launch each instead of 5-6 eager launches per layer.
 *   nvh_add_rmsnorm  RMSNorm.forward(x[, residual]), nanovllm/layers/layernorm.py:17-50: if `residual` is non-NULL it is
 *                    updated in place to bf16(x + residual) and the norm is taken of the fp32 sum; out = norm * weight.
 *                    x/out/residual [n_rows, hidden] bf16 with row strides in elements; hidden % 8 == 0, hidden <= 8192.
 *   nvh_silu_mul     SiluAndMul.forward, nanovllm/layers/activation.py:11-14: out[:, i] = silu(gu[:, i]) * gu[:, inter + i].
 */
int nvh_add_rmsnorm(void* out, const void* x, void* residual, const void* weight, float eps, int n_rows, int hidden,
                    int64_t x_row_stride, int64_t out_row_stride, int64_t residual_row_stride, int dtype, void* stream);
int nvh_silu_mul(void* out, const void* gate_up, int n_rows, int inter, int64_t gate_up_row_stride, int64_t out_row_stride,
                 int dtype, void* stream);
/*   nvh_argmax_rows  greedy sampling (temperature 0; nanovllm/layers/sampler.py, bench_my.py:31): out[i] = argmax_j x[i, j],
 *                    int64, ties -> lowest index; x [n_rows, n] bf16, 16-byte aligned rows (row stride % 8 == 0). */
int nvh_argmax_rows(int64_t* out, const void* x, int n_rows, int n, int64_t x_row_stride, int dtype, void* stream);
/*   nvh_greedy_advance  the same arg-max fused with the host work between two decode steps, done on the device so a step is
 *                    pure graph replay: append the token (engine/scheduler.py:99-110) and rebuild next step's decode metadata
 *                    (engine/model_runner.py:244-269): tokens_log[row_steps[r], r] = tok; row_steps[r]++; input_ids[r] = tok;
 *                    positions[r]++; context_lens[r]++; slot_mapping[r] = block_tables[r, (ctx-1)/bs]*bs + (ctx-1)%bs with the
 *                    new ctx.  Rows whose context_lens is 0 (graph padding) are left untouched. */
int nvh_greedy_advance(const void* logits, int n_rows, int n, int64_t logits_row_stride,
                       int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                       const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                       int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps, int dtype, void* stream);

/*
 * Engine widening: weight-streaming linear layer for decode-sized batches, out = x . W^T (+ bias), m <= 64 rows.
 * Replaces F.linear in nanovllm/layers/linear.py:55-190 at decode (the hipBLASLt path stays for larger m).
 *   x [m, k] bf16 (row stride in elements); w [n, k] bf16 contiguous (nn.Linear layout); bias [n] bf16 or NULL
 *   silu_inter == 0 : out [m, n] = x w^T + bias
 *   silu_inter  > 0 : n == 2*silu_inter, rows [0, inter) of w are gate_proj and [inter, 2*inter) up_proj
 *                     (MergedColumnParallelLinear, linear.py:90-111); out [m, inter] = SiLU(x gate^T) * (x up^T), i.e. the
 *                     projection and layers/activation.py:11-14 in one launch; bias must be NULL.
 *   k % 64 == 0, n % 16 == 0 (and inter % 16 == 0).
 */
int nvh_linear_small_m(void* out, const void* x, const void* w, const void* bias, int m, int n, int k, int silu_inter,
                       int64_t x_row_stride, int64_t out_row_stride, int dtype, void* stream);

/*
 * The same weight-streaming product with the neighbouring row-wise ops of a decoder layer fused in, so that a decode
 * layer is 6 launches (qkv, attention split, attention combine, o_proj, gate_up, down) instead of 10:
 *   norm_weight != NULL : prologue x := RMSNorm(x) * norm_weight   (RMSNorm.rms_forward, nanovllm/layers/layernorm.py:17-27)
 *   norm_folded != 0    : the same norm with its weight already multiplied into w by the caller (w := w * diag(norm_weight));
 *                         sum(x^2) is taken inside the K loop and rows are scaled by rsqrt(mean(x^2) + norm_eps) in the epilogue
 *                         (same algebra, without re-rounding the normalised activations to bf16); norm_weight must be NULL
 *   epilogue NVH_EPI_NONE          out [m, n] = product + bias
 *            NVH_EPI_SILU_MUL      as nvh_linear_small_m with silu_inter
 *            NVH_EPI_RESIDUAL_ADD  out is the residual stream [m, n]: out += product, in place (the add of
 *                                  add_rms_forward, layernorm.py:35-36, done by the producing GEMM instead of the consuming norm)
 *            NVH_EPI_ROPE_STORE    n == (h + 2*kvh) * hd (fused qkv projection, models/qwen3.py:104-106): + bias, neox RoPE on
 *                                  the q and k heads, q -> out [m, h*hd]; k and v rows -> k_cache / v_cache at slot_mapping[row]
 *                                  (slot < 0: skipped).  Same arithmetic and rounding points as nvh_rope_store; no q/k norm.
 */
#define NVH_EPI_NONE          0
#define NVH_EPI_SILU_MUL      1
#define NVH_EPI_RESIDUAL_ADD  2
#define NVH_EPI_ROPE_STORE    3
typedef struct nvh_linear_desc {
    void* out;
    const void* x;
    const void* w;
    const void* bias;               /* NULL or [n] */
    int32_t m, n, k, silu_inter;
    int64_t x_row_stride, out_row_stride;
    const void* norm_weight;        /* NULL or [k] */
    float norm_eps;
    int32_t epilogue;
    const int64_t* positions;       /* ROPE_STORE only, below */
    const float* cos_sin;
    void* k_cache;
    void* v_cache;
    const int32_t* slot_mapping;
    int32_t h, kvh, hd;
    int32_t norm_folded;
    /* streaming form (csrc/linear_stream.hip), all optional: */
    int32_t x_packed;               /* x is in MFMA-fragment order [ceil(m/16)][k/32][64][8] (nvh_pack_index); x_row_stride unused */
    void* out_packed;               /* NULL or: the epilogue also writes its bf16 result in fragment order
                                       ([ceil(m/16)][cols/32][64][8], cols = n, or silu_inter for SILU_MUL); `out` may then be
                                       NULL for NONE / SILU_MUL */
    void* workspace;                /* NULL or nvh_linear_small_m_workspace() bytes, ZERO-FILLED once by the caller (the
                                       kernel leaves its tickets zero); needed when k > 1024: K is then split over
                                       workgroups and the last-arriving one sums the partials in a fixed order */
    size_t workspace_bytes;
    /* greedy candidates (NONE epilogue, k <= 1024): every workgroup also writes, per row, the largest bf16 output among its
       columns and that column's index to candidate_val / candidate_idx [workgroup * candidate_stride + row]
       (nvh_linear_small_m_candidate_groups() workgroups; ties -> lowest column).  With candidates `out` may be NULL: the LM
       head + arg-max of a greedy decode step then never materialises the logits (layers/embed_head.py:66 +
       layers/sampler.py at temperature 0); finish with nvh_greedy_advance_candidates. */
    float* candidate_val;
    int32_t* candidate_idx;
    int64_t candidate_stride;
    /* prefetch hint (streaming form, optional): device memory the NEXT launches on this stream will stream once — typically the
       weights of the following projection.  A decode-sized projection occupies 36-224 of the device's CUs; the workgroups this
       launch adds for the idle ones read the range (default cache policy) and drop it, so that the later launch finds its operand
       in the caches.  Pure hint: results do not depend on it, only whole lines inside [prefetch, prefetch + prefetch_bytes) are
       read, NULL / 0 = none; ranges above 4 MiB are ignored (large prefetches cost the launch more than they save). */
    const void* prefetch;
    size_t prefetch_bytes;
} nvh_linear_desc;
int nvh_linear_small_m_ex(const nvh_linear_desc* desc, int dtype, void* stream);
/* residual[n_rows, hidden] += y (bf16, one rounding — the add of add_rms_forward, layers/layernorm.py:35-36) and, if `packed` is
 * not NULL, the updated rows again in fragment order for the next nvh_linear_small_m_ex with x_packed: the step between a
 * tensor-parallel all-reduce of a row-parallel projection (layers/linear.py:185-190) and the next projection. */
int nvh_residual_add_pack(void* residual, const void* y, void* packed, int n_rows, int hidden, int64_t residual_row_stride,
                          int64_t y_row_stride, int dtype, void* stream);
size_t nvh_linear_small_m_workspace(int m, int n, int k, int epilogue);
int nvh_linear_small_m_candidate_groups(int n, int k);      /* candidate records per row of a NONE launch; 0 = unsupported shape */

/*
 * "Next" row (taken to its end): the front of a decode layer as ONE call — the fused qkv projection
 * (+ folded RMSNorm, bias, RoPE, K/V store) and the decode attention on its output.  Replaces the sequence nanovllm/models/qwen3.py:104-117
 * (qkv_proj -> rotary_emb -> self.attn) with nanovllm/layers/attention.py:84-86 (store_kvcache) and :99-101 (flash_attn_with_kvcache)
 * inside it.  Result: that of nvh_linear_small_m_ex(qkv) followed by nvh_paged_decode_packed on the q rows it wrote.
 *   qkv            a NVH_EPI_ROPE_STORE descriptor (see nvh_linear_small_m_ex): out = the q rows [m, h*hd], k_cache / v_cache /
 *                  slot_mapping = where this step's K / V rows go; prefetch = the output projection's weights (optional hint)
 *   attn_out       [m, h, hd] bf16; attn_out_packed: NULL or the same rows in fragment order (as nvh_paged_decode_packed)
 *   block_tables / context_lens / block_size / max_blocks / bt_row_stride / scale   as nvh_paged_decode (context_lens COUNT the token
 *                  this step stores, model_runner.py:252-258; rows with context 0 give zeros)
 *   workspace      nvh_paged_decode_workspace(m, h, hd, max_blocks, block_size) bytes, zero-filled once
 * Three forms were built and measured on MI355X (profiles/r03_qkv_attend_*); nvh_qkv_rope_attend runs the winner,
 * nvh_qkv_rope_attend_variant reaches all of them (tests and A/B):
 *   mode 1  TWO launches (the projection, then the attention call).  The winner at every shape tried; mode 0 = this.
 *   mode 2  ONE launch (csrc/qkv_attend.hip): producer workgroups compute the projection's tiles and publish q and the new K/V rows
 *           write-through; the attention workgroups start their K/V stream with the launch and wait (bounded) only for those.  Needs
 *           hd == 64, k <= 1024 (no split-K), x_packed, norm_folded, h / kvh <= 16, kvh <= 15 and a grid of at most two workgroups
 *           per CU (all resident: consumers wait for producers) on a device this stream has to itself; NVH_E_SHAPE otherwise.
 *           q and the cache rows are the same bits as mode 1's, the attention output the same bits as NVH_DECODE_CHUNKED_P128's.
 *           2.2 us per layer SLOWER than mode 1 at Qwen2-0.5B bs = 32: the stream's 16+ MB in flight queue in front of every load of the
 *           producer -> consumer chain.  A consumer whose wait runs out (~1 s) writes NaN rows and sets a status word instead of
 *           hanging (nvh_qkv_rope_attend_status, host-synchronous); spin_limit > 0 shortens the wait; missing_producers > 0 makes
 *           every consumer wait for producers that do not exist (the time-out path).
 *   mode 3  mode 1 with the projection launch's idle CUs touching the first `spin_limit` (default 1) passes of every attention workgroup's
 *           K/V into the caches (hd 64).  +1.3 % on the decode step with one pass, +5 % with two: a loss as well.
 * *one_launch (nullable) reports whether the one-launch kernel ran.  Capture-safe in every mode.
 */
int nvh_qkv_rope_attend(const nvh_linear_desc* qkv, void* attn_out, void* attn_out_packed,
                        const int32_t* block_tables, const int32_t* context_lens, int block_size, int max_blocks,
                        int64_t bt_row_stride, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int nvh_qkv_rope_attend_variant(int mode, uint32_t spin_limit, int missing_producers, int* one_launch,
                                const nvh_linear_desc* qkv, void* attn_out, void* attn_out_packed,
                                const int32_t* block_tables, const int32_t* context_lens, int block_size, int max_blocks,
                                int64_t bt_row_stride, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int nvh_qkv_rope_attend_status(const void* workspace, uint32_t* timed_out);
/* nvh_greedy_advance on candidate records instead of logits: token = column of the best candidate of each row */
int nvh_greedy_advance_candidates(const float* candidate_val, const int32_t* candidate_idx, int groups, int64_t candidate_stride,
                                  int n_rows, int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                                  const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                                  int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps, void* stream);
/* The same, and the NEXT step's token embedding in the same launch (VocabParallelEmbedding.forward at tp = 1,
 * nanovllm/layers/embed_head.py:34-45): hidden_out[r, :] = embed_weight[input_ids[r], :] with the token just chosen (rows with
 * context_lens 0 keep their token), and, if hidden_packed is not NULL, the same rows in fragment order for a following
 * nvh_linear_small_m_ex with x_packed.  embed_weight [vocab, hidden] bf16 contiguous, hidden % 32 == 0, 16-byte aligned buffers.
 * The chosen token is always a valid row of embed_weight: arg-max follows torch.argmax (NaN counts as the maximum, ties -> lowest
 * index, an all -inf row -> 0) and the index is clamped to [0, vocab). */
int nvh_greedy_advance_candidates_embed(const float* candidate_val, const int32_t* candidate_idx, int groups, int64_t candidate_stride,
                                        int n_rows, int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                                        const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                                        int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps,
                                        const void* embed_weight, int vocab, int hidden, void* hidden_out, int64_t hidden_row_stride, void* hidden_packed,
                                        int dtype, void* stream);
/* offset (in elements) of activation element (row, col) of an [m, cols] matrix in fragment order */
int64_t nvh_pack_index(int row, int col, int cols);

/*
 * "Next" row: the step immediately AFTER attention under tensor parallelism — the all-reduce of the
 * row-parallel projections' partial sums (RowParallelLinear.forward, nanovllm/layers/linear.py:185-190: dist.all_reduce over
 * NCCL) with the residual add that follows it (add_rms_forward, nanovllm/layers/layernorm.py:35-36) fused in.  One-shot over
 * xGMI instead of RCCL's ring: every rank maps every peer's staging buffer and flag table through hipIpc handles, signals with
 * one remote store per peer, reads the p-1 peers' partials over p-1 distinct links at once and sums in rank order in fp32 (all
 * ranks produce the same bits).  Decode-sized messages ([<= 64, hidden] bf16); capture-safe; no trailing barrier (staging is
 * double-buffered by a device-resident epoch).  A dead peer ends in NaN rows and a failure mark, not a hang.
 *
 * Set-up (host-synchronous, once, NOT capturable), per rank:
 *   nvh_comm_alloc(&stage, nvh_allreduce_stage_bytes(max_rows, hidden));   nvh_comm_alloc(&flags, nvh_allreduce_flag_bytes(world));
 *   nvh_comm_alloc(&state, 64);                                            (zero-filled fine-grained device memory)
 *   nvh_comm_ipc_export(stage / flags, handle[NVH_COMM_IPC_HANDLE_BYTES]) -> exchange the handles by any host channel ->
 *   nvh_comm_ipc_open(peer handle, &peer_ptr) for every peer; upload the two [world] pointer tables (own pointers at [rank])
 *   to device memory.  `state` stays local.
 * Call (every rank, same order of calls, same rows / hidden):
 *   x              [rows, hidden] bf16 partial sums of this rank (row stride x_row_stride elements)
 *   epilogue       NVH_AR_EPI_NONE: out[rows, hidden] = bf16(sum over ranks);  out may alias x
 *                  NVH_AR_EPI_RESIDUAL_ADD: out is the residual stream: out = bf16(out + bf16(sum)); packed (nullable): the
 *                  updated rows again in fragment order (nvh_pack_index) for the next nvh_linear_small_m_ex with x_packed
 *   stage_bytes    size of EACH rank's staging buffer (>= nvh_allreduce_stage_bytes(rows, hidden))
 *   state          uint32[16], local: [0] calls completed, [2] != 0 after a peer timed out OR was found two calls ahead (the epoch that
 *                  failed; that call's rows are NaN), [3] if non-zero replaces the poll limit (tests force the time-out with it)
 * nvh_allreduce_status (host-synchronous: waits for the device, NOT capturable) reads those two words back.
 */
#define NVH_COMM_IPC_HANDLE_BYTES 64
enum { NVH_AR_EPI_NONE = 0, NVH_AR_EPI_RESIDUAL_ADD = 1 };
int nvh_comm_alloc(void** ptr, size_t bytes);
int nvh_comm_free(void* ptr);
int nvh_comm_ipc_export(void* ptr, void* handle_out);
int nvh_comm_ipc_open(const void* handle, void** ptr);
int nvh_comm_ipc_close(void* ptr);
size_t nvh_allreduce_stage_bytes(int max_rows, int hidden);
size_t nvh_allreduce_flag_bytes(int world);
int nvh_allreduce_status(const void* state, uint32_t* calls_completed, uint32_t* failed_epoch);
int nvh_allreduce_oneshot(void* out, const void* x, void* packed, void* const* stage_ptrs, void* const* flag_ptrs, void* state,
                          int world, int rank, int rows, int hidden, int64_t x_row_stride, int64_t out_row_stride,
                          size_t stage_bytes, int epilogue, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NVH_ATTN_H */
