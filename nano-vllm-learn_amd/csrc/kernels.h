// Internal launcher interface between api.hip (argument validation, C ABI) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nvh {

int launch_store_kvcache(const void* k, const void* v, void* k_cache, void* v_cache,
                         const int32_t* slot_mapping, int n_tokens, int kvh, int hd,
                         int64_t k_row_stride, int64_t v_row_stride, hipStream_t stream);

constexpr int kPv16GroupRows = 64;   // rows per range flag of the bf16 -> fp16 conversion (one conversion workgroup per group)
int launch_bf16_rows_to_f16(void* out, const void* in, int n_rows, int row_elems, int64_t in_row_stride, int64_t out_row_stride, int32_t* group_flags, hipStream_t stream);

struct DecodeArgs {
    void* out;                   // [B, H, D] bf16 or f32
    const uint16_t* q;           // [B, H, D] bf16, row stride q_row_stride
    const uint16_t* k_cache;     // [NB, bs, KVH, D]
    const uint16_t* v_cache;
    const int32_t* block_tables; // [B, max_blocks], row stride bt_row_stride
    const int32_t* context_lens; // [B]
    float* ws_acc;               // [B, H, num_splits, D] un-normalised partial outputs
    float* ws_ml;                // [B, H, num_splits, 2]  (running max in log2 domain, sum)
    int batch, h, kvh, hd, block_size, max_blocks, num_splits;
    int64_t q_row_stride, bt_row_stride;
    float scale_log2;            // softmax scale * log2(e)
    int out_f32;
    unsigned long long* stamps;  // diagnostic builds only (NVH_STAMPS); null otherwise
    // chunked kernel (default): partial records [B*KVH][chunks][G rows of (D + 4) floats: O, max, sum, 0, 0; padded to 256 B] fp32 alias ws_acc; arrival tickets per (b, kv head)
    unsigned* counters;          // [B*KVH], zero before the launch, left zero by it
    int chunks;                  // workgroups per (sequence, kv head); passes are dealt to them round-robin
    uint16_t* out_packed;        // nullable: bf16 output also in MFMA-fragment order [ceil(B/16)][H*D/32][64][8] (pack_index)
    int impl;                    // 0 chunked MFMA (default), 1 split MFMA + combine, 2 split VALU + combine (nvh_paged_decode_variant)
    int waves;                   // chunked kernel: 0 / 8 = eight waves per workgroup, 4 = four
    int pass_tokens;             // chunked kernel: 0 = chosen from the chunk count; forced: 128 / 256 at D = 64, 64 / 128 at D = 128
};

// tokens one workgroup of the split kernel covers (static function of head_dim)
int decode_split_tokens(int hd);
int launch_paged_decode(const DecodeArgs& a, hipStream_t stream);
int decode_chunks(int batch, int kvh, int num_splits, int forced);

struct RopeStoreArgs {
    uint16_t* qkv;               // [N, (H+2KVH)*D] fused projection output, rotated in place
    const int64_t* positions;    // [N]
    const float* cos_sin;        // [max_position, D]: cos(0..D/2) | sin(0..D/2)
    const uint16_t* q_norm_w;    // [D] or null (no per-head RMSNorm)
    const uint16_t* k_norm_w;
    float eps;
    uint16_t* k_cache;           // nullable: no store
    uint16_t* v_cache;
    const int32_t* slot_mapping; // nullable
    int n_tokens, h, kvh, hd;
    int64_t qkv_row_stride;
};
int launch_rope_store(const RopeStoreArgs& a, hipStream_t stream);

int launch_add_rmsnorm(void* out, const void* x, void* residual, const void* w, float eps, int n_rows, int hidden,
                       int64_t x_stride, int64_t out_stride, int64_t res_stride, hipStream_t stream);
int launch_silu_mul(void* out, const void* gate_up, int n_rows, int inter, int64_t gu_stride, int64_t out_stride, hipStream_t stream);
int max_rmsnorm_hidden(void);
int launch_residual_add_pack(void* residual, const void* y, void* packed, int n_rows, int hidden, int64_t res_stride, int64_t y_stride,
                             hipStream_t stream);
struct AdvanceArgs {             // all null: plain argmax
    int64_t* input_ids;          // [rows] next step's input token
    int64_t* positions;          // [rows] += 1
    int32_t* context_lens;       // [rows] += 1 (rows with 0 are padding: untouched)
    int32_t* slot_mapping;       // [rows] slot of the token the next step stores
    const int32_t* block_tables; // [rows, width]
    int64_t bt_stride;
    int block_size;
    int64_t* tokens_log;         // [steps, log_stride] generated tokens
    int64_t log_stride;
    int64_t* row_steps;          // [rows] tokens generated so far per row (log row index)
    // next step's embedding lookup (candidates form only; all null: none): embed[input_ids[row]] -> hidden_out row and,
    // if hidden_packed, the same row in fragment order (pack_index) for the first projection of the next step
    const uint16_t* embed;       // [vocab, hidden] bf16
    int hidden;
    uint16_t* hidden_out;        // [rows, hidden], row stride hidden_stride
    int64_t hidden_stride;
    uint16_t* hidden_packed;     // nullable
};
int launch_argmax_rows(int64_t* out, const void* x, int n_rows, int n, int64_t stride, const AdvanceArgs& adv, hipStream_t stream);
int launch_argmax_candidates(const float* cand_val, const int32_t* cand_idx, int groups, int64_t cand_stride, int n_rows, int vocab,
                             const AdvanceArgs& adv, hipStream_t stream);   // vocab: chosen indices are clamped to [0, vocab)
int linear_stream_candidate_groups(int n, int k);   // workgroups (= candidate records per row) of a NONE launch, 0 if unsupported

enum { EPI_NONE = 0, EPI_SILU = 1, EPI_RESADD = 2, EPI_ROPE = 3 };   // == NVH_EPI_* in nvh_attn.h

struct LinearArgs {
    void* out;                   // NONE: [M, N]; SILU: [M, inter]; RESADD: the residual stream [M, N], updated in place; ROPE: q [M, H*D]
    const uint16_t* x;           // [M, K] bf16, row stride x_stride
    const uint16_t* w;           // [N, K] bf16 contiguous (nn.Linear layout)
    const uint16_t* bias;        // [N] bf16 or null
    int M, N, K;
    int inter;                   // SILU: N == 2*inter, gate rows [0, inter), up rows [inter, 2*inter)
    int64_t x_stride, out_stride;
    const uint16_t* norm_w;      // norm_mode 1: RMSNorm weight [K]
    float norm_eps;
    int norm_mode;               // 0 none; 1 exact RMSNorm prologue; 2 folded RMSNorm (w already multiplied by the norm weight)
    int epi;
    const int64_t* positions;    // ROPE: [M]
    const float* cos_sin;        // ROPE: [max_position, D]
    uint16_t* k_cache;           // ROPE: paged caches and slots of this step's rows
    uint16_t* v_cache;
    const int32_t* slots;
    int h, kvh, hd;              // ROPE: N == (h + 2*kvh) * hd
    unsigned long long* stamps;  // diagnostic builds only (NVH_STAMPS); null otherwise
    // ---- streaming kernel only (linear_stream.hip)
    int x_packed;                // x is in MFMA-fragment order: [ceil(M/16)][K/32][64 lanes][8] bf16 (see pack_index)
    uint16_t* out_packed;        // nullable: the epilogue also writes its bf16 result in fragment order, for the next GEMM
    void* ws_raw;                // caller's workspace (counters + partials), ws_bytes long; split into the two fields below
    size_t ws_bytes;
    float* ws;                   // split-K partials [tile][split][NB*MT*256 + MT*16] fp32 (null: no split-K)
    unsigned* counters;          // [tiles] arrival tickets, zero before the launch, left zero by it
    int ksplit;                  // workgroups sharing one tile's K range (>= 1)
    int tiles;                   // weight row tiles (16 rows, or 16 + 16 partner rows for SILU / ROPE)
    float* cand_val;             // NONE only, nullable: greedy candidates instead of (or besides) the outputs: every workgroup
    int32_t* cand_idx;           // writes, per row, the largest bf16 output of its columns and that column's index:
    int64_t cand_stride;         // cand_*[workgroup * cand_stride + row]; ties -> lowest column (lm_head + argmax in one pass)
    const void* pf_ptr;          // nullable: bytes a LATER launch will stream (its weights): the CUs this launch leaves idle read them
    int64_t pf_bytes;            // once (default cache policy), so that the later launch finds them in the memory-side cache
    // the same for the FIRST K/V images of the decode attention launch that follows a ROPE projection (nvh_qkv_rope_attend mode 3): the
    // prefetching workgroups walk block_tables / context_lens and touch the rows the attention workgroups will ask for first — never the
    // row of token ctx - 1, which this very launch writes
    struct KvPrefetch {
        const uint16_t* k_cache;     // null: none
        const uint16_t* v_cache;
        const int32_t* block_tables;
        const int32_t* context_lens;
        int64_t bt_stride;
        int batch, kvh, hd, block_size, chunks, pass_tokens, passes;   // the attention launch's geometry; `passes` first passes per workgroup
    } pf_kv;
};
int launch_linear_small_m(const LinearArgs& a, hipStream_t stream);
int launch_linear_stream(const LinearArgs& a, hipStream_t stream);   // linear_stream.hip; returns -100 when the shape is not its own
size_t linear_stream_workspace_bytes(int m, int n, int k, int epi);

// element (row, col) of an [M, C] activation in MFMA-fragment order (A operand of v_mfma_f32_16x16x32_bf16):
// [row / 16][col / 32][lane = 16 * ((col / 8) % 4) + row % 16][col % 8]
__host__ __device__ inline int64_t pack_index(int row, int col, int cols) {
    return ((((int64_t)(row >> 4) * (cols >> 5) + (col >> 5)) * 64 + (((col >> 3) & 3) << 4) + (row & 15)) << 3) + (col & 7);
}

// one-shot all-reduce over IPC-mapped peer buffers (allreduce_oneshot.hip)
enum { AR_EPI_NONE = 0, AR_EPI_RESIDUAL_ADD = 1 };       // == NVH_AR_EPI_* in nvh_attn.h
constexpr int AR_MAX_BLOCKS = 32;                        // workgroups per launch at most: the stride of a flag table row
struct AllReduceArgs {
    const uint16_t* x;           // [rows, hidden] bf16: this rank's partial sums, row stride x_stride
    uint16_t* out;               // NONE: the reduced rows; RESIDUAL_ADD: the residual stream, updated in place; row stride out_stride
    uint16_t* packed;            // RESIDUAL_ADD, nullable: the updated rows again in MFMA-fragment order (pack_index)
    void* const* stage;          // device array [world]: staging buffers (2 slots of slot_bytes each), own at [rank], peers IPC-mapped
    uint32_t* const* flags;      // device array [world]: flag tables [world][AR_MAX_BLOCKS] uint32, own at [rank]
    uint32_t* state;             // local: [0] epoch of the last finished call, [1] finished workgroups of the running one, [2] failed epoch
    size_t slot_bytes;
    int world, rank, rows, hidden, epi;
    int64_t x_stride, out_stride;
};
int allreduce_blocks(int rows, int hidden);
int launch_allreduce_oneshot(const AllReduceArgs& a, int blocks, hipStream_t stream);

// qkv projection (RoPE / store epilogue) + decode attention in ONE launch (qkv_attend.hip)
size_t qkv_attend_sync_bytes(void);
bool qkv_attend_supported(const LinearArgs& l, const DecodeArgs& d);
int launch_qkv_attend(const LinearArgs& l, const DecodeArgs& d, void* sync, unsigned spin_limit, int missing_producers, const void* pf_ptr,
                      int64_t pf_bytes, hipStream_t stream);

struct PrefillArgs {
    void* out;                   // [Tq, H, D]
    const uint16_t* q;           // [Tq, H, D], row stride q_row_stride
    const uint16_t* k;           // [Tk, KVH, D] rows (k_row_stride) or paged cache when block_tables != null
    const uint16_t* v;
    const int32_t* cu_q;         // [B+1]
    const int32_t* cu_k;         // [B+1]
    const int32_t* block_tables; // nullable
    int batch, max_seqlen_q, max_seqlen_k, h, kvh, hd, block_size, max_blocks;
    int64_t q_row_stride, k_row_stride, v_row_stride, bt_row_stride;
    float scale_log2;
    int out_f32;
    unsigned long long* stamps;  // diagnostic builds only (NVH_STAMPS); null otherwise
    int kernel;                  // 0 auto, 1 tiled kernel only, 2 short-sequence kernel (error if the shape does not allow it)
    int short_waves;             // short-sequence kernel: 0 auto, 8 or 16 waves per workgroup
    int short_pv16 = 0;                   // nvh_prefill_varlen_pv16 on the short-sequence kernel's shapes: convert V to fp16 inside the kernel (D = 64)
    const uint16_t* v16 = nullptr;        // nvh_prefill_varlen_pv16: fp16 copy of the V rows [Tk, KVH, D] (row stride v16_row_stride) ...
    int64_t v16_row_stride = 0;
    int pv16_rows = 0;                    // rows converted (total_k of the call): a sequence reaching past them takes the exact form
    const int32_t* pv16_flags = nullptr;  // ... and one word per kPv16GroupRows rows, non-zero where a value did not fit fp16 (that sequence uses `v`, hi + lo bf16 P)
};
int launch_prefill_varlen(const PrefillArgs& a, hipStream_t stream);

}  // namespace nvh
